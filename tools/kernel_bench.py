"""Per-kernel timing of the hand-written kernels at a given batch (HIP events on the launch
stream): launch time, algorithmic HBM bytes and the fraction of the 8 TB/s
HBM roofline.  Not the headline benchmark (bench.py) - a tool to see which kernel is where.

    python tools/kernel_bench.py --batch 128
    python tools/kernel_bench.py --batch 16384 --iters 20
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from feta_tmlr_amd import _lib   # noqa: E402

PEAK = 8000.0


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--n-pad', type=int, default=37)
    ap.add_argument('--heads', type=int, default=4)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--order', type=int, default=4)
    ap.add_argument('--k-eig', type=int, default=16)
    ap.add_argument('--iters', type=int, default=100)
    ap.add_argument('--json', action='store_true')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help='storage type of the fused-stack rows (the general-kernel rows stay fp32)')
    ap.add_argument('--batch-first', action='store_true',
                    help='token tensors stored [B,N,...] (a graph is contiguous) instead of the reference seq-first [N,B,...]')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    abi, st = _lib.abi(), _lib.stream_handle()
    b, n, h, d, p, k = a.batch, a.n_pad, a.heads, a.dim, a.order, a.k_eig
    dh = d // h
    c = p * dh * dh
    m = n * b
    g = torch.Generator(device='cpu').manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    nr = torch.randint(9, n + 1, (b,), generator=g, dtype=torch.int32).to(dev)
    mean_n = float(nr.float().mean())

    bf = a.batch_first
    lead = (b, n) if bf else (n, b)
    as_bn = (lambda t: t) if bf else (lambda t: t.transpose(0, 1))   # -> [B, N, ...] view
    qkv = rnd(*lead, 3 * d)
    v5 = qkv.view(*lead, 3, h, dh)
    q, kk, v = (as_bn(v5[:, :, i]) for i in range(3))
    tok = lambda: as_bn(torch.empty(*lead, h, dh, device=dev))
    out, dout = tok(), as_bn(rnd(*lead, h, dh))
    attn = torch.empty(b, h, n, n, device=dev)
    stats = torch.empty(b, h, n, 2, device=dev)
    pe = torch.rand(b, n, n, generator=g).to(dev)
    delta = torch.empty(b, h, n, device=dev)
    dqkv = torch.empty_like(qkv)
    g5 = dqkv.view(*lead, 3, h, dh)
    dq, dk, dv = (as_bn(g5[:, :, i]) for i in range(3))
    sc = dh ** -0.5
    rows = []

    def add(name, fn, nbytes):
        t = timeit(fn, a.iters)
        rows.append((name, t * 1e6, nbytes, nbytes / t / 1e9))

    f4 = 4
    add('attn_fwd (+attn write)', lambda: abi.attn_fwd(q, kk, v, pe, nr, out, attn, stats, sc, st),
        f4 * b * (3 * n * d + n * n + n * d + 2 * h * n + h * n * n))
    add('attn_fwd (no attn write)', lambda: abi.attn_fwd(q, kk, v, pe, nr, out, None, stats, sc, st),
        f4 * b * (3 * n * d + n * n + n * d + 2 * h * n))
    add('attn_bwd (dq + dkdv)', lambda: abi.attn_bwd(q, kk, v, pe, nr, out, dout, stats, delta, dq, dk, dv, sc, st),
        f4 * b * (3 * n * d + 2 * n * d + n * n + 2 * h * n + 3 * n * d + h * n))

    s = rnd(c)
    gb = rnd(c) * 0.1
    cj = torch.empty(h * b, n, device=dev)
    pooled = torch.empty(h * b, c, device=dev)
    attn_in = torch.rand(b, h, n, n, generator=g).to(dev)
    add('coeff_fwd', lambda: abi.coeff_fwd(attn_in, nr, s, gb, cj, pooled, st), f4 * b * (h * n * n + h * c + h * n))
    groups = abi.coeff_bwd_groups(b, h)
    partial = torch.empty(2, groups, c, device=dev)
    ds, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    dpool = rnd(h * b, c)
    add('coeff_bwd (+2 colsum)', lambda: abi.coeff_bwd(cj, nr, s, gb, dpool, partial, ds, db, b, n, h, st),
        f4 * b * (h * c + h * n))

    x = as_bn(rnd(*lead, h, dh))
    y, dy, dx = tok(), as_bn(rnd(*lead, h, dh)), tok()
    coeff = rnd(h * b, c)
    bias = rnd(dh)
    u = rnd(b, n, k)
    lam = torch.rand(b, k, generator=g).to(dev) * 2 - 1
    lhat = rnd(b, n, n) * 0.1
    dcoeff = torch.empty_like(coeff)
    dbp = torch.empty(b * h, dh, device=dev)
    add('spec_filter_fwd (K=%d)' % k, lambda: abi.spec_filter_fwd(x, u, lam, coeff, bias, nr, y, p, 1, st),
        f4 * b * (n * d + n * k + k + h * c + n * d))
    add('spec_filter_bwd (K=%d)' % k,
        lambda: abi.spec_filter_bwd(x, u, lam, coeff, nr, dy, dx, dcoeff, dbp, p, 1, st),
        f4 * b * (2 * n * d + n * k + k + h * c + n * d + h * c))
    add('cheb_filter_fwd', lambda: abi.cheb_filter_fwd(x, lhat, coeff, bias, nr, y, p, 1, st),
        f4 * b * (n * d + n * n + h * c + n * d))
    add('cheb_filter_bwd', lambda: abi.cheb_filter_bwd(x, lhat, coeff, nr, dy, dx, dcoeff, dbp, p, 1, st),
        f4 * b * (2 * n * d + n * n + h * c + n * d + h * c))

    x2 = rnd(m, d)
    for (ki, no, nm) in ((d, 3 * d, 'in_proj'), (d, d, 'out_proj'), (d, 2 * d, 'linear1'), (2 * d, d, 'linear2')):
        xi = rnd(m, ki)
        w = rnd(no, ki) / ki ** 0.5
        bb = rnd(no)
        yo = torch.empty(m, no, device=dev)
        sto = torch.empty(abi.rowlin_blocks(m) + 1, 2, no, device=dev)
        add('rowlin_fwd %s %dx%d' % (nm, ki, no), lambda: abi.rowlin_fwd(xi, w, bb, None, None, yo, sto, False, st),
            f4 * (m * ki + no * ki + m * no))
        dyo = rnd(m, no)
        dxi = torch.empty(m, ki, device=dev)
        part = torch.empty(abi.rowlin_chunks(m), no * ki + no, device=dev)
        dwdb = torch.empty(no * ki + no, device=dev)
        add('rowlin_bwd %s (+colsum)' % nm, lambda: abi.rowlin_bwd(xi, w, dyo, None, None, dxi, part, dwdb, st),
            f4 * (2 * m * ki + m * no + 2 * no * ki))
    yb = rnd(m, d)
    stb = torch.empty(abi.rowlin_blocks(m) + 1, 2, d, device=dev)
    abi.bn_stats(yb, stb, st)
    ob = torch.empty(m, d, device=dev)
    mr = torch.empty(2, d, device=dev)
    gm, bt = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    add('bn_apply_fwd', lambda: abi.bn_apply_fwd(yb, stb, gm, bt, ob, mr, None, None, 0.1, 1e-5, st), f4 * 2 * m * d)
    pb = torch.empty(abi.rowlin_blocks(m), 2, d, device=dev)
    dyb = torch.empty(m, d, device=dev)
    dg, dbt = torch.empty(d, device=dev), torch.empty(d, device=dev)
    add('bn_bwd (reduce + apply)', lambda: abi.bn_bwd(yb, ob, mr, gm, pb, dyb, dg, dbt, st), f4 * 5 * m * d)

    if not bf:   # the kernels of a fused-stack layer in the variants the stack issues
        from feta_tmlr_amd.benchcases import stack_layer_cases
        lowp = a.dtype == 'bf16'
        for name, _, fn, nbytes, _ in stack_layer_cases(abi, st, dev, b, n, d, h, 2 * d, pe, nr,
                                                        dtype=torch.bfloat16 if lowp else torch.float32):
            add(name, fn, nbytes)
        # the C x C linear of the coefficient generator (csrc/lin.hip) with the column sums it carries in the step:
        # filter-bias partials and linear_cat's split-K weight-gradient partials
        r_ = h * b
        if abi.lin_supported(r_, c, c):
            lw, lb, lx, ldy = rnd(c, c) / c ** 0.5, rnd(c), rnd(r_, c), rnd(r_, c)
            ly, ldx, ldw, ldb = (torch.empty(r_, c, device=dev), torch.empty(r_, c, device=dev),
                                 torch.empty(c, c, device=dev), torch.empty(c, device=dev))
            add('lin_fwd %dx%dx%d' % (r_, c, c), lambda: abi.lin_fwd(lx, lw, lb, ly, st, bf16=lowp), f4 * (2 * r_ * c + c * c))
            cat_part = rnd(abi.rowlin_chunks(m), d * 2 * d + d)
            pairs = [(rnd(r_, dh), torch.empty(dh, device=dev)), (cat_part, torch.empty(cat_part.shape[1], device=dev))]
            add('lin_bwd (dx, dw, db + 2 colsum)', lambda: abi.lin_bwd(lx, lw, ldy, ldx, ldw, ldb, st, pairs=pairs, bf16=lowp),
                f4 * (3 * r_ * c + 2 * c * c + cat_part.numel()))

    if a.json:
        print(json.dumps({'batch': b, 'mean_nodes': mean_n,
                          'kernels': [{'name': r[0], 'us': round(r[1], 2), 'bytes': r[2],
                                       'GBps': round(r[3], 1), 'frac': round(r[3] / PEAK, 4)} for r in rows]}))
        return
    print('batch %d graphs, N_pad %d (mean n %.1f), d %d, H %d, P %d, K %d' % (b, n, mean_n, d, h, p, k))
    print('%-34s %10s %12s %10s %8s' % ('kernel', 'us/launch', 'alg. MB', 'GB/s', 'of 8TB/s'))
    for name, us, nb, gbs in rows:
        print('%-34s %10.2f %12.3f %10.1f %7.1f%%' % (name, us, nb / 1e6, gbs, 100 * gbs / PEAK))


if __name__ == '__main__':
    main()
