set -e
OUT=gpurun_out/r3r
mkdir -p $OUT
python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "attn or coeff" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for cfg in "--shape pattern --batch 64 --n-pad 128 --k-eig 32" "--shape pattern --batch 64 --n-pad 188 --k-eig 32" "--shape pattern --batch 64 --n-pad 120 --k-eig 32 --layer-norm" "" "--dtype bf16" "--shape molhiv --batch 1024 --n-pad 64"; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > $OUT/b.json 2> $OUT/b.err
  python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('$cfg', d['value'], d['ms_per_step'])
P
done
export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p188 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 188 --k-eig 32 > $OUT/p188.json 2> $OUT/p188.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p128 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32 > $OUT/p128.json 2> $OUT/p128.err
