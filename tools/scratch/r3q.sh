set -e
OUT=gpurun_out/r3q
mkdir -p $OUT
python -m pytest tests/test_modules_gpu.py -m gpu -x -q -k "attn_out or layernorm_stack" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for ao in 0 1; do
for cfg in "--shape pattern --batch 64 --n-pad 128 --k-eig 32" "--shape pattern --batch 64 --n-pad 188 --k-eig 32" "--shape pattern --batch 64 --n-pad 120 --k-eig 32 --layer-norm"; do
  FETA_ATTN_OUT=$ao python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > $OUT/b.json 2> $OUT/b.err
  python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('ATTN_OUT=$ao $cfg', d['value'], d['ms_per_step'])
P
done
done
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/pattern -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32 > $OUT/pattern.json 2> $OUT/pattern.err
