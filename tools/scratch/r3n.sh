set -e
mkdir -p gpurun_out/r3n
for form in "FETA_BLOCK_FWD_WAVES=4" "FETA_BLOCK_FWD_WGS=2"; do
  tag=$(echo $form | tr '=' '_')
  for dt in f32 bf16; do
    env $form python tools/block_timing.py --kernel fwd --dtype $dt > gpurun_out/r3n/timing_${tag}_$dt.txt 2>&1
    for rep in 1 2; do
    env $form python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-literal --stream-batch 0 --dtype $dt > gpurun_out/r3n/bench_${tag}_$dt.json 2> gpurun_out/r3n/bench_${tag}_$dt.err
    python - <<P
import json
d=json.loads(open('gpurun_out/r3n/bench_${tag}_$dt.json').read().strip().splitlines()[-1])
r=d['roofline']
allk=[(r['kernel'],r['launch_us'])]+[(o['kernel'],o['launch_us']) for o in r['other_kernels']]
print('$form $dt', d['value'], d['ms_per_step'], allk)
P
    done
  done
done
python tools/ffn_timing.py > gpurun_out/r3n/ffn_timing.txt 2>&1 || true
