set -e
mkdir -p gpurun_out/r3o
python -m pytest tests -m gpu -x -q > gpurun_out/r3o/tests.log 2>&1 || { tail -30 gpurun_out/r3o/tests.log; exit 1; }
tail -2 gpurun_out/r3o/tests.log
for form in "FETA_BLOCK_FWD_WAVES=4" "FETA_BLOCK_FWD_WAVES=8"; do
  tag=$(echo $form | tr '=' '_')
  for cfg in "--batch 512" "--shape molhiv --batch 1024 --n-pad 64" "--shape molhiv --batch 1024 --n-pad 64 --dtype bf16" "--batch 512 --dtype bf16" "--shape mutag --batch 32 --n-pad 28 --k-eig 8 --layer-norm"; do
    env $form python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 $cfg > gpurun_out/r3o/b.json 2> gpurun_out/r3o/b.err
    python - <<P
import json
d=json.loads(open('gpurun_out/r3o/b.json').read().strip().splitlines()[-1])
r=d['roofline']
allk=[(r['kernel'],r['launch_us'])]+[(o['kernel'],o['launch_us']) for o in r['other_kernels']]
print('$form $cfg', d['value'], d['ms_per_step'], [k for k in allk if 'attn_block_fwd' in k[0]])
P
  done
done
