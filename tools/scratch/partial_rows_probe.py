"""How much of each fused kernel's launch is the re-reduction of its producer's partial rows: the same launch with
G = 1 / 128 / 148 / 256 partial rows (HIP events, 200 launches)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from feta_tmlr_amd import _lib
abi, st = _lib.abi(), _lib.stream_handle()
dev = torch.device('cuda:0')
b, n, d, h, ff = 128, 37, 64, 4, 128
m = b * n
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
new = lambda *s: torch.empty(*s, device=dev)
nr = torch.randint(9, n + 1, (b,), generator=g, dtype=torch.int32).to(dev)
pe = torch.rand(b, n, n, generator=g).to(dev)
prm = (torch.rand(4, d, generator=g) + 0.5).to(dev)

def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for G in (1, 128, 148, 256):
    stats = rnd(G + 1, 2, d).abs()
    # ffn_fwd
    x, w1, b1, w2, b2 = rnd(m, d), rnd(ff, d) / 8, rnd(ff), rnd(d, ff) / 11, rnd(d)
    hb, y2, st2 = new(m, ff), new(m, d), new(abi.ffn_blocks(m) + 1, 2, d)
    fd = abi.ffn_desc(m, ff, Gx=G, x=x, w1=w1, b1=b1, w2=w2, b2=b2, h=hb, y=y2, y_stats=st2, x_stats=stats,
                      x_gamma=prm[0], x_beta=prm[1], x_bn_out=new(4, d))
    t_ffn = timeit(lambda: abi.ffn_launch(fd, st))
    # attn_block_fwd
    w_in, b_in, w_o, b_o, deg = rnd(3 * d, d) / 8, rnd(3 * d), rnd(d, d) / 8, rnd(d), torch.rand(m, generator=g).to(dev)
    qkv, out, y1, st1 = new(m, 3 * d), new(m, d), new(m, d), new(abi.attn_block_stat_rows(b, n) + 1, 2, d)
    ast = new(b, h, n, 2)
    d0 = abi.attn_block_desc(b, n, 0.25, attn=None, x=x, w_in=w_in, b_in=b_in, w_out=w_o, b_out=b_o, pe=pe, n_real=nr,
                             rowscale=deg, qkv=qkv, out=out, attn_stats=ast, y=y1, y_stats=st1, x_stats=stats, Gx=G,
                             x_gamma=prm[0], x_beta=prm[1], x_bn_out=new(4, d))
    t_blk = timeit(lambda: abi.attn_block_launch(d0, st))
    print('G = %3d   ffn_fwd %6.2f us   attn_block_fwd %6.2f us' % (G, t_ffn, t_blk))
