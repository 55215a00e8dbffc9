#!/bin/bash
# Kernel statistics and PMC counters of ONE bench.py configuration on the GPU box (through gpurun):
#   tools/profile_config.sh <tag> <bench.py flags ...>     e.g.  tools/profile_config.sh molhiv_bf16 --shape molhiv --batch 1024 --n-pad 64 --dtype bf16
# kernel-trace statistics of the captured step (hipGraph replays), then three --pmc passes of the same step launched
# eagerly (--no-graph; separate runs with --kernel-trace only: MI355X_MICROARCH.md, rocprofv3 section).
# Output: gpurun_out/prof_<tag>/{stats,fetch,write,sq}/...; tools/collect_config.sh turns it into profiles/<prefix>_<tag>_*.
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P="rocprofv3 --output-format csv"
BF="--steps 30 --warmup 5 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1"
$P --kernel-trace --stats -d $OUT/stats -- python3 bench.py $BF "$@" > $OUT/bench.json 2> $OUT/bench.err
BP="--steps 6 --warmup 2 --no-graph --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1"
$P --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python3 bench.py $BP "$@" >> $OUT/pmc.err 2>&1
$P --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python3 bench.py $BP "$@" >> $OUT/pmc.err 2>&1
$P --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES -d $OUT/sq -- python3 bench.py $BP "$@" >> $OUT/pmc.err 2>&1
echo "profile $TAG done"
