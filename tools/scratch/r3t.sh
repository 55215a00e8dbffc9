set -e
OUT=gpurun_out/r3t
mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "test_attn" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for cfg in "--shape pattern --batch 64 --n-pad 128 --k-eig 32" "--shape pattern --batch 64 --n-pad 188 --k-eig 32" "--shape pattern --batch 64 --n-pad 120 --k-eig 32 --layer-norm"; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > $OUT/b.json 2> $OUT/b.err
  python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$cfg', d['value'], d['ms_per_step'], [(r['kernel'],r['launch_us'])]+[(o['kernel'],o['launch_us']) for o in r['other_kernels']])
P
done
