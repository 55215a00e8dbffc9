set -e
OUT=gpurun_out/r3z
mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -x -q -k "test_attn or attn_out or layernorm_stack" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for hk in 0 1; do
for cfg in "--shape pattern --batch 64 --n-pad 188 --k-eig 32" "--shape pattern --batch 64 --n-pad 160 --k-eig 32"; do
  FETA_ATTN_BWD_HEAD=$hk python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > $OUT/b.json 2> $OUT/b.err
  python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('HEAD=$hk $cfg', d['value'], d['ms_per_step'], [(r['kernel'],r['launch_us'])]+[(o['kernel'],o['launch_us']) for o in r['other_kernels']][:3])
P
done
done
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p128 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32 > $OUT/p128.json 2> $OUT/p128.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p188 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 188 --k-eig 32 > $OUT/p188.json 2> $OUT/p188.err
