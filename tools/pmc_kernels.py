"""Mean of every collected PMC counter per kernel symbol, from one or more rocprofv3
counter_collection.csv files (one --pmc pass each).

usage: python tools/pmc_kernels.py <counter_collection.csv> [...] [--match substr] [--json out.json]
"""
import collections
import csv
import json
import re
import sys

args = sys.argv[1:]
match, out_json = '', None
files = []
i = 0
while i < len(args):
    if args[i] == '--match':
        match = args[i + 1]
        i += 2
    elif args[i] == '--json':
        out_json = args[i + 1]
        i += 2
    else:
        files.append(args[i])
        i += 1

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if match not in name:
            continue
        short = re.sub(r'\(.*', '', name).replace('void ', '')
        acc[short][r['Counter_Name']].append(float(r['Counter_Value']))

res = {}
for k in sorted(acc):
    res[k] = {c: sum(v) / len(v) for c, v in acc[k].items()}
    res[k]['dispatches'] = max(len(v) for v in acc[k].values())
    print(k)
    for c in sorted(acc[k]):
        print('    %-32s %16.1f' % (c, res[k][c]))
if out_json:
    json.dump(res, open(out_json, 'w'), indent=1)
