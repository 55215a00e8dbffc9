set -e
OUT=gpurun_out/r3p
mkdir -p $OUT
export TMPDIR=/tmp
BF="--steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/pattern -- python3 bench.py $BF --shape pattern --batch 64 --n-pad 128 --k-eig 32 > $OUT/pattern.json 2> $OUT/pattern.err
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --shape pattern --batch 64 --n-pad 128 --k-eig 32 > $OUT/pattern_plain.json 2> $OUT/pattern_plain.err
find $OUT -name "*kernel_stats.csv"
