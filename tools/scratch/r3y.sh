set -e
OUT=gpurun_out/r3y
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
bash tools/profile_round.sh r03g > gpurun_out/r03g.log 2>&1
tail -2 gpurun_out/r03g.log
for k in fwd bwd; do
  for dt in f32 bf16; do
    python tools/block_timing.py --kernel $k --dtype $dt > $OUT/timing_${dt}_$k.txt 2>&1
    [ $k = bwd ] && python tools/block_timing.py --kernel bwd --split --dtype $dt > $OUT/timing_${dt}_bwdsplit.txt 2>&1
  done
done
python tools/train_bench.py > $OUT/train_bench.txt 2>&1 || true
