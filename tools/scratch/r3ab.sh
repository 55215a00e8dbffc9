set -e
OUT=gpurun_out/r3ab
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
for ec in 0 1; do
for cfg in "" "--dtype bf16" "--batch 512" "--shape molhiv --batch 1024 --n-pad 64 --dtype bf16"; do
  FETA_EARLY_COLSUM=$ec python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > $OUT/b.json 2> $OUT/b.err
  python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('EARLY=$ec $cfg', d['value'], d['ms_per_step'])
P
done
done
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/bench -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 > $OUT/p.json 2> $OUT/p.err
