#!/bin/bash
# One profiling session on the GPU box (run through gpurun): kernel-trace statistics of bench.py, HBM traffic and
# MFMA-busy counters of the per-kernel micro-benchmark at the BASELINE batch and at a streaming batch.  PMC passes are
# separate runs with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).  Output: gpurun_out/$1/...
set -e
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
P="rocprofv3 --output-format csv"
$P --kernel-trace --stats -d $OUT/bench -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench trace done"
for B in 128 16384; do
  IT=20; [ $B = 16384 ] && IT=5
  $P --kernel-trace --stats -d $OUT/kb_${B}_trace -- python3 tools/kernel_bench.py --batch $B --iters $IT > $OUT/kb_${B}.txt 2>> $OUT/kb.err
  $P --kernel-trace --pmc FETCH_SIZE -d $OUT/kb_${B}_fetch -- python3 tools/kernel_bench.py --batch $B --iters $IT >> $OUT/kb.err 2>&1
  $P --kernel-trace --pmc WRITE_SIZE -d $OUT/kb_${B}_write -- python3 tools/kernel_bench.py --batch $B --iters $IT >> $OUT/kb.err 2>&1
  $P --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES -d $OUT/kb_${B}_mfma -- python3 tools/kernel_bench.py --batch $B --iters $IT >> $OUT/kb.err 2>&1
  echo "kernel_bench B=$B done"
done
find $OUT -name "*.csv" | sort
