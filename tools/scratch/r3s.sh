set -e
OUT=gpurun_out/r3s
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<P
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['launch_us'], d['roofline']['frac'])
print('bf16', d['bf16_leg']['value'], 'two-phase', d['two_phase_n1']['value'])
for e in d['extra_configs']:
    print(e['config'], e['value'], e['ms_per_step'], e['roofline']['kernel'], e['roofline']['launch_us'], e['roofline']['frac'])
print(d['cpu_baseline'])
P
