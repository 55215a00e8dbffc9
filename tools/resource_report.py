"""Per-kernel resources of libfeta_hip.so from hipcc's -Rpass-analysis=kernel-resource-usage remarks:

    python tools/resource_report.py [file.hip ...] [--scratch]      # default: every csrc/*.hip

name (demangled), VGPRs, AGPRs, scratch bytes per lane, waves per SIMD, LDS bytes.  --scratch: only kernels that spill."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'feta_tmlr_amd', 'csrc')


def report(src):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
           '-I' + CSRC, '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: (?:[^:]+: )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                      r'LDS Size \[bytes/block\]|SGPRs): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(' ')[0]] = v
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r['name'] = re.sub(r'\(.*', '', n.replace('feta::', '').replace('void ', ''))
    return rows


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    files = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    for f in files:
        for r in report(f):
            if '--scratch' in sys.argv and r.get('ScratchSize', '0') == '0':
                continue
            print('%-72s vgpr %4s agpr %3s scratch %4s occ %2s' % (r['name'][:72], r.get('VGPRs'), r.get('AGPRs'),
                                                                    r.get('ScratchSize'), r.get('Occupancy')))
