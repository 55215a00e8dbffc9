"""Phase timing inside ffn_fwd_kernel (s_memtime stamps of workgroup 0, thread 0; diagnostic build of
tools/block_timing.py).   python tools/ffn_timing.py [--batch 128]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from feta_tmlr_amd import _abi   # noqa: E402
import block_timing              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--n-pad', type=int, default=37)
a = ap.parse_args()
if not os.path.exists(block_timing.OUT):
    block_timing.build()
lib = ctypes.CDLL(block_timing.OUT)
abi = _abi.bind(lib)
dev = torch.device('cuda:0')
m, d, ff = a.batch * a.n_pad, 64, 128
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
x, w1, b1, w2, b2 = rnd(m, d), rnd(ff, d) / 8, rnd(ff), rnd(d, ff) / 11, rnd(d)
h, y, st2 = torch.empty(m, ff, device=dev), torch.empty(m, d, device=dev), torch.empty(abi.ffn_blocks(m), 2, d, device=dev)
g1 = min(a.batch, 256)
stats1, prm = rnd(g1, 2, d).abs(), torch.rand(2, d, generator=g).to(dev) + 0.5
desc = abi.ffn_desc(m, ff, Gx=g1, x=x, w1=w1, b1=b1, w2=w2, b2=b2, h=h, y=y, y_stats=st2, x_stats=stats1,
                    x_gamma=prm[0], x_beta=prm[1], x_bn_out=torch.empty(4, d, device=dev))
stream = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    abi.ffn_launch(desc, stream)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.feta_debug_ffn_stamps(buf)
t = list(buf)
for i, nm in enumerate(['x / bias / weight requests + LDS staging', 'BatchNorm finalize', 'BN apply + first product (+ h store)',
                        'second product', 'exchange + epilogue + statistics']):
    print('%-44s %8d cycles' % (nm, t[i + 1] - t[i]))
print('%-44s %8d cycles' % ('total', t[5] - t[0]))
