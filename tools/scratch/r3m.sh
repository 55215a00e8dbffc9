set -e
mkdir -p gpurun_out/r3m
python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "attn_block or fused_kernels_edge" > gpurun_out/r3m/tests.log 2>&1 || { tail -30 gpurun_out/r3m/tests.log; exit 1; }
tail -3 gpurun_out/r3m/tests.log
for form in "FETA_BLOCK_FWD_WAVES=4" "FETA_BLOCK_FWD_WGS=1" "FETA_BLOCK_FWD_WGS=2"; do
  tag=$(echo $form | tr '=' '_')
  for dt in f32 bf16; do
    env $form python tools/block_timing.py --kernel fwd --dtype $dt > gpurun_out/r3m/timing_${tag}_$dt.txt 2>&1
    env $form python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-literal --stream-batch 0 --dtype $dt > gpurun_out/r3m/bench_${tag}_$dt.json 2> gpurun_out/r3m/bench_${tag}_$dt.err
    python - <<P
import json
d=json.loads(open('gpurun_out/r3m/bench_${tag}_$dt.json').read().strip().splitlines()[-1])
r=d['roofline']
allk=[(r['kernel'],r['launch_us'])]+[(o['kernel'],o['launch_us']) for o in r['other_kernels']]
print('$form $dt', d['value'], d['ms_per_step'], [k for k in allk if 'attn_block_fwd' in k[0]])
P
  done
done
