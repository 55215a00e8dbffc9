"""Builds feta_tmlr_amd/libfeta_hip.so (the C ABI of include/feta_hip.h) with hipcc for gfx950.

    python -m feta_tmlr_amd.build [--force] [--report]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
OUT = os.path.join(PKG, 'libfeta_hip.so')
OBJ = os.path.join(PKG, 'csrc', 'build')
ARCH = 'gfx950'


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return hdr + [os.path.join(ROOT, 'include', 'feta_hip.h'), os.path.abspath(__file__)]


def build(force=False, report=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    flags = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC',
             '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
    if report:
        flags.append('-Rpass-analysis=kernel-resource-usage')
    dep_t = max(os.path.getmtime(p) for p in _deps())
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_t):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([_hipcc()] + flags + ['-c', src, '-o', obj], capture_output=True, text=True)
        return job, r

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for (src, obj), r in ex.map(cc, jobs):
            if r.returncode != 0:
                raise RuntimeError('hipcc failed on %s:\n%s' % (src, r.stderr[-4000:]))
            if report:
                sys.stderr.write(r.stderr)
            if verbose:
                print('hipcc -c', os.path.relpath(src, ROOT))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + '.o') for s in sources()]
    if jobs or not os.path.exists(OUT):
        r = subprocess.run([_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', OUT] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if verbose:
            print('linked', os.path.relpath(OUT, ROOT))
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv, report='--report' in sys.argv)
