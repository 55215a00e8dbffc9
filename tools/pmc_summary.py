"""Joins rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/kernel_bench.py (dispatch order is
deterministic: 3 warm-up + ITERS launches per bench row) into per-kernel-config HBM traffic.
usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <iters> <kb.log>"""
import csv, json, re, sys

def seq(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return [(r['Kernel_Name'], float(r['Counter_Value'])) for r in rows if 'feta::' in r['Kernel_Name']]

f, w = seq(sys.argv[1]), seq(sys.argv[2])
iters = int(sys.argv[3]) + 3
names = [l.split('  ')[0].strip() for l in open(sys.argv[4]) if re.search(r'\d+\.\d+%\s*$', l)]
# launches per bench row (kernels each C-ABI call enqueues)
per_call = {'attn_bwd': 2, 'coeff_bwd': 2, 'rowlin_bwd': 2, 'bn_bwd': 2}
out, i = [], 0
for nm in names:
    k = next((v for p, v in per_call.items() if nm.startswith(p)), 1)
    nf = f[i:i + iters * k]; nw = w[i:i + iters * k]
    i += iters * k
    fk = sum(v for _, v in nf) / iters; wk = sum(v for _, v in nw) / iters
    out.append({'bench_row': nm, 'kernels': sorted(set(n.split('(')[0] for n, _ in nf)),
                'FETCH_SIZE_KB_per_call': round(fk, 1), 'WRITE_SIZE_KB_per_call': round(wk, 1)})
print(json.dumps(out, indent=1))
