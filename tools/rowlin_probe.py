"""Times feta_rowlin_bwd_ex with individual options switched on (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feta_tmlr_amd import _lib
abi, st = _lib.abi(), _lib.stream_handle()
dev = torch.device('cuda:0')
m = 37 * 128
def timeit(fn, iters=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for ki, no in ((128, 64), (64, 64), (64, 128), (64, 192)):
    G, RC = abi.rowlin_blocks(m), abi.rowlin_chunks(m)
    r = lambda *s: torch.randn(*s, device=dev)
    x, w, dy, dx = r(m, ki), r(no, ki), r(m, no), r(m, ki)
    part, dwdb = r(RC, no * ki + no), r(no * ki + no)
    gy, gbn, gsum, gfin = r(m, no), r(4, no), r(G, 2, no), r(2, no)
    dg, db, gfo = r(no), r(no), r(2, no)
    ad, ay, abn, afin = r(m, ki), r(m, ki), r(4, ki), r(2, ki)
    sy, sbn, so = r(m, ki), r(4, ki), r(G, 2, ki)
    rs = torch.rand(m, device=dev)
    base = dict(x=x, w=w, dy=dy, dx=dx, partial=part)
    variants = {
        'plain': {},
        'g_y+g_fin': dict(g_y=gy, g_bn=gbn, g_fin=gfin),
        'g_y+g_sum': dict(g_y=gy, g_bn=gbn, g_sum=gsum, g_fin_out=gfo, dgamma=dg, dbeta=db),
        'add': dict(add_dout=ad, add_y=ay, add_bn=abn, add_fin=afin),
        'sums': dict(sum_y=sy, sum_bn=sbn, sum_out=so),
        'add+sums': dict(add_dout=ad, add_y=ay, add_bn=abn, add_fin=afin, sum_y=sy, sum_bn=sbn, sum_out=so),
        'x_bn': dict(x_bn=abn),
        'rowscale': dict(rowscale=rs),
    }
    for name, extra in variants.items():
        kw = dict(base); kw.update(extra)
        d = abi.rowlin_ex(m, ki, no, Gs=G, **kw)
        t = timeit(lambda: abi.rowlin_bwd_ex(d, dwdb, st))
        print('KI=%3d NO=%3d %-12s %7.2f us (incl. colsum)' % (ki, no, name, t))
