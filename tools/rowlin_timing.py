"""Phase timing inside rowlin_bwd_kernel (s_memtime stamps: dX role = workgroup 0, dW role = the first
dW workgroup), diagnostic build of tools/block_timing.py.  Shapes of the fused stack at the BASELINE
batch.   python tools/rowlin_timing.py [--batch 128]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--n-pad', type=int, default=37)
    a = ap.parse_args()
    if not os.path.exists(block_timing.OUT):
        block_timing.build()
    lib = ctypes.CDLL(block_timing.OUT)
    abi = _abi.bind(lib)
    dev = torch.device('cuda:0')
    m = a.batch * a.n_pad
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    G = abi.rowlin_blocks(m)
    RC = abi.rowlin_chunks(m)
    for name, ki, no, extras in (('in_proj bwd (B5: add + sums)', 64, 192, 'as'),
                                 ('out_proj bwd (B3: BN-backward gradient)', 64, 64, 'g'),
                                 ('linear1 bwd (B2: relu, add, sums)', 64, 128, 'ras'),
                                 ('linear2 bwd (B1: BN-backward gradient)', 128, 64, 'g')):
        x, w, dy, dx = rnd(m, ki), rnd(no, ki) / ki ** 0.5, rnd(m, no), torch.empty(m, ki, device=dev)
        total = no * ki + no
        part = torch.empty(RC, total, device=dev)
        kw = dict(x=x, w=w, dy=dy, dx=dx, partial_ptr=part.data_ptr(), partial_ld=total)
        prm = torch.rand(4, max(ki, no), generator=g).to(dev)
        if 'g' in extras:
            kw.update(g_y=rnd(m, no), g_bn=prm[:, :no].contiguous(), g_sum=rnd(G, 2, no), Gs=G,
                      g_fin_out=torch.empty(2, no, device=dev), dgamma=torch.empty(no, device=dev),
                      dbeta=torch.empty(no, device=dev))
        if 'r' in extras:
            kw.update(relu_y=rnd(m, no))
        if 'a' in extras:
            kw.update(add_dout=rnd(m, ki), add_y=rnd(m, ki), add_bn=prm[:, :ki].contiguous(), add_fin=rnd(2, ki))
        if 's' in extras:
            kw.update(sum_y=rnd(m, ki), sum_bn=prm[:, :ki].contiguous(), sum_out=torch.empty(G, 2, ki, device=dev))
        dsc = abi.rowlin_ex(m, ki, no, **kw)
        for _ in range(5):
            abi.rowlin_bwd_ex(dsc, None, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            abi.rowlin_bwd_ex(dsc, None, st)
        e1.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        lib.feta_debug_rowlin_stamps(buf)
        t = list(buf)
        print('%s  KI=%d NO=%d: %.2f us/launch' % (name, ki, no, e0.elapsed_time(e1) * 1e3 / 50))
        xs = ['finalize + params', 'load batch + W staging', 'MFMA + epilogue', 'sums reduce']
        print('   dX role: ' + ', '.join('%s %d' % (nm, t[i + 1] - t[i]) for i, nm in enumerate(xs)) + ' cycles')
        ws = ['finalize + params', 'staging loads', 'barrier', 'MFMA', 'partial store']
        print('   dW role: ' + ', '.join('%s %d' % (nm, t[17 + i] - t[16 + i]) for i, nm in enumerate(ws)) + ' cycles')
        print('   dW role starts %d cycles after the dX role of workgroup 0' % (t[16] - t[0]))


if __name__ == '__main__':
    main()
