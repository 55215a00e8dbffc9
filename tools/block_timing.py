"""Phase timing inside attn_block_fwd_kernel (s_memtime stamps of workgroup 0, wave 0): builds a
diagnostic copy of the library with -DFETA_TIMING into tools/_timing/ (git-ignored) and prints the
cycles between stamps.  Run on the GPU box:  python tools/block_timing.py [--batch 128]"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feta_tmlr_amd import _abi   # noqa: E402

OUT = os.path.join(ROOT, 'tools', '_timing', 'libfeta_timing.so')


def build():
    srcs = sorted(glob.glob(os.path.join(ROOT, 'feta_tmlr_amd', 'csrc', '*.hip')))
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DFETA_TIMING',
           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'feta_tmlr_amd', 'csrc')] + srcs + ['-o', OUT]
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--n-pad', type=int, default=37)
    ap.add_argument('--build-only', action='store_true')
    a = ap.parse_args()
    if not os.path.exists(OUT) or a.build_only:
        build()
    if a.build_only:
        return
    lib = ctypes.CDLL(OUT)
    abi = _abi.bind(lib)
    dev = torch.device('cuda:0')
    b, n, d, h = a.batch, a.n_pad, 64, 4
    m = b * n
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    nr = torch.randint(9, n + 1, (b,), generator=g, dtype=torch.int32).to(dev)
    x, w_in, b_in, w_o, b_o = rnd(m, d), rnd(3 * d, d) / 8, rnd(3 * d), rnd(d, d) / 8, rnd(d)
    pe = torch.rand(b, n, n, generator=g).to(dev)
    qkv, out = torch.empty(m, 3 * d, device=dev), torch.empty(m, d, device=dev)
    y, yst = torch.empty(m, d, device=dev), torch.empty(b, 2, d, device=dev)
    ast = torch.empty(b, h, n, 2, device=dev)
    deg = torch.rand(m, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        abi.attn_block_fwd(b, n, 0.25, st, x=x, w_in=w_in, b_in=b_in, w_out=w_o, b_out=b_o, pe=pe, n_real=nr,
                           rowscale=deg, qkv=qkv, out=out, attn_stats=ast, attn=None, y=y, y_stats=yst)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.feta_debug_block_stamps(buf)
    t = list(buf)[:6]
    names = ['loads + staging + finalize', 'in_proj', 'attention core', 'barrier wait', 'concat store + out_proj']
    print('n_real[0] =', int(nr[0]))
    for i, nm in enumerate(names):
        print('%-30s %8d cycles' % (nm, t[i + 1] - t[i]))
    print('%-30s %8d cycles' % ('total', t[5] - t[0]))


if __name__ == '__main__':
    main()
