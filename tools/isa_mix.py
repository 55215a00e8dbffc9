"""Instruction mix of the gfx950 code of one source file, per kernel (a quick look at what a wave
spends its issue slots on: MFMA vs VALU address arithmetic vs memory instructions).
usage: python tools/isa_mix.py feta_tmlr_amd/csrc/filter.hip [substring-of-kernel-name]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
out = '/tmp/isa_mix.s'
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                '-I' + os.path.join(ROOT, 'feta_tmlr_amd', 'csrc'), '--cuda-device-only', '-S', src, '-o', out],
               check=True, stderr=subprocess.DEVNULL)
name, counts = None, None
res = []
for line in open(out):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        name, counts = m.group(1), collections.Counter()
        res.append((name, counts))
        continue
    if counts is None or not line.startswith('\t'):
        continue
    tok = line.strip().split()
    if not tok or tok[0].startswith('.') or tok[0].startswith(';'):
        continue
    op = tok[0]
    if op == 's_endpgm':
        counts['total'] += 1
        counts = None
        continue
    counts['total'] += 1
    if op.startswith('v_mfma'):
        counts['mfma'] += 1
    elif op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
        counts['load'] += 1
    elif op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store')):
        counts['store'] += 1
    elif op.startswith('ds_'):
        counts['lds'] += 1
    elif op.startswith('s_waitcnt'):
        counts['waitcnt'] += 1
    elif op.startswith('s_load'):
        counts['sload'] += 1
    elif op.startswith('v_'):
        counts['valu'] += 1
    elif op.startswith('s_'):
        counts['salu'] += 1
    else:
        counts['other'] += 1
for name, c in res:
    if pat in name and c['total'] > 20:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        print('%-70s %s' % (dem[:70], dict(c)))
