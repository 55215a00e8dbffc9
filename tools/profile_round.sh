#!/bin/bash
# One profiling session on the GPU box (run through gpurun): kernel-trace statistics of bench.py (fp32 headline leg and
# the bf16 storage leg), HBM traffic and MFMA-busy counters of the per-kernel micro-benchmark at the BASELINE batch (both
# storage types) and at a streaming batch.  PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md,
# rocprofv3 section).  Output: gpurun_out/$1/...
set -e
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
P="rocprofv3 --output-format csv"
BF="--steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1"
$P --kernel-trace --stats -d $OUT/bench -- python3 bench.py $BF > $OUT/bench.json 2> $OUT/bench.err
$P --kernel-trace --stats -d $OUT/bench_bf16 -- python3 bench.py $BF --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
echo "bench traces done"
for CFG in "128 f32" "128 bf16" "16384 f32"; do
  set -- $CFG; B=$1; DT=$2
  IT=20; [ $B = 16384 ] && IT=5
  T=${B}_${DT}
  $P --kernel-trace --stats -d $OUT/kb_${T}_trace -- python3 tools/kernel_bench.py --batch $B --iters $IT --dtype $DT > $OUT/kb_${T}.txt 2>> $OUT/kb.err
  $P --kernel-trace --pmc FETCH_SIZE -d $OUT/kb_${T}_fetch -- python3 tools/kernel_bench.py --batch $B --iters $IT --dtype $DT >> $OUT/kb.err 2>&1
  $P --kernel-trace --pmc WRITE_SIZE -d $OUT/kb_${T}_write -- python3 tools/kernel_bench.py --batch $B --iters $IT --dtype $DT >> $OUT/kb.err 2>&1
  $P --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES -d $OUT/kb_${T}_mfma -- python3 tools/kernel_bench.py --batch $B --iters $IT --dtype $DT >> $OUT/kb.err 2>&1
  echo "kernel_bench B=$B $DT done"
done
find $OUT -name "*.csv" | sort
