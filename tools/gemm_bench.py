"""The C x C ``self.linear`` of the coefficient generator (transformer/models.py:145,284) at the BASELINE shape
R = H*B = 512, K = N = C = 1024: feta_lin_fwd / feta_lin_bwd (csrc/lin.hip, LDS-tiled; fp32 and bf16 compute) against
the library GEMMs PyTorch picks, inside a hipGraph (what a captured step pays).  Run on the GPU box."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feta_tmlr_amd import _lib      # noqa: E402


def graph_time(fn, reps=10, inner=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(s.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn(s.cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--r', type=int, default=512)
    ap.add_argument('--k', type=int, default=1024)
    ap.add_argument('--n', type=int, default=1024)
    a = ap.parse_args()
    abi = _lib.abi()
    dev = torch.device('cuda:0')
    x, w, b = torch.randn(a.r, a.k, device=dev), torch.randn(a.n, a.k, device=dev) / a.k ** 0.5, torch.randn(a.n, device=dev)
    dy = torch.randn(a.r, a.n, device=dev)
    y, dx, dw, db = torch.empty(a.r, a.n, device=dev), torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    gf = 2.0 * a.r * a.k * a.n / 1e9
    for bf16 in (False, True):
        t = graph_time(lambda st: abi.lin_fwd(x, w, b, y, st, bf16=bf16))
        print('lin_fwd %s: %.2f us  (%.1f TFLOP/s)' % ('bf16' if bf16 else 'fp32', t, gf / t * 1e-3 * 1e3))
        t = graph_time(lambda st: abi.lin_bwd(x, w, dy, dx, dw, db, st, bf16=bf16))
        print('lin_bwd %s: %.2f us  (%.1f TFLOP/s over both products)' % ('bf16' if bf16 else 'fp32', t, 2 * gf / t))
    t = graph_time(lambda st: torch.addmm(b, x, w.t(), out=y))
    print('library addmm fp32: %.2f us' % t)
    t = graph_time(lambda st: (torch.mm(dy, w, out=dx), torch.mm(dy.t(), x, out=dw)))
    print('library dX + dW fp32: %.2f us' % t)
    xb, wb, dyb = x.bfloat16(), w.bfloat16(), dy.bfloat16()
    yb = torch.empty(a.r, a.n, device=dev, dtype=torch.bfloat16)
    t = graph_time(lambda st: torch.mm(xb, wb.t(), out=yb))
    print('library mm bf16 (operands already bf16): %.2f us' % t)


if __name__ == '__main__':
    main()
