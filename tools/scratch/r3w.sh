set -e
OUT=gpurun_out/r3w
mkdir -p $OUT
python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "attn_block or fused_kernels_edge or layernorm_stack or bf16" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for lp in 0 1; do
for cfg in "--shape molhiv --batch 1024 --n-pad 64" "--batch 512" "--batch 2048"; do
  FETA_BLOCK_BWD_LOOP=$lp python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > $OUT/b.json 2> $OUT/b.err
  python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('LOOP=$lp $cfg', d['value'], d['ms_per_step'], [(r['kernel'],r['launch_us'])]+[(o['kernel'],o['launch_us']) for o in r['other_kernels']][:4])
P
done
done
for cfg in "--shape molhiv --batch 1024 --n-pad 64 --dtype bf16" "--batch 512 --dtype bf16"; do
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > $OUT/b.json 2> $OUT/b.err
  python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$cfg', d['value'], d['ms_per_step'], [(r['kernel'],r['launch_us'])]+[(o['kernel'],o['launch_us']) for o in r['other_kernels']][:4])
P
done
