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
    ap.add_argument('--lib', default=OUT, help='a diagnostic build to load instead (A/B timing of an older source)')
    ap.add_argument('--no-stats', dest='stats', action='store_false')
    a = ap.parse_args()
    if not os.path.exists(OUT) or a.build_only:
        build()
    if a.build_only:
        return
    lib = ctypes.CDLL(a.lib)
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
    # as inside the layer stack: the input is seen through the previous BatchNorm, whose statistics this launch
    # finalizes from the per-block partial sums
    G = abi.ffn_blocks(m) if a.stats else 0
    xst = torch.rand(max(G, 1), 2, d, generator=g).to(dev) + 1.0
    xst[:, 1] += 2.0 * m
    gam, bet, prm = rnd(d), rnd(d), torch.empty(4, d, device=dev)
    kw = dict(x_stats=xst, x_gamma=gam, x_beta=bet, x_bn_out=prm) if a.stats else {}
    print('partial rows of the input statistics:', G)
    call = lambda: abi.attn_block_fwd(b, n, 0.25, st, Gx=G, x=x, w_in=w_in, b_in=b_in, w_out=w_o, b_out=b_o, pe=pe,
                                      n_real=nr, rowscale=deg, qkv=qkv, out=out, attn_stats=ast, attn=None, y=y,
                                      y_stats=yst, **kw)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    # eight launches back to back inside one hipGraph (what a captured step issues)
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        st = s2.cuda_stream
        call()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s2):
            for _ in range(8):
                call()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print('launch-to-launch inside the graph: %.2f us' % (e0.elapsed_time(e1) / 160 * 1e3))
    buf = (ctypes.c_ulonglong * 256)()
    lib.feta_debug_block_stamps(buf)
    t = [[[buf[(l * 8 + w) * 8 + i] for i in range(8)] for w in range(8)] for l in range(4)]
    nwg = (b + 31) // 32
    names = ['loads + staging + finalize', 'in_proj', 'attention core', 'barrier wait', 'concat store + out_proj']
    order = sorted(range(4), key=lambda l: t[l][0][0])      # the ring of the last four launches, oldest first
    tick = 0.01   # us per s_memrealtime tick (100 MHz)
    print('n_real[0] =', int(nr[0]))
    for li, l in enumerate(order):
        t0 = min(t[l][w][0] for w in range(nwg))
        t5 = max(t[l][w][5] for w in range(nwg))
        line = 'launch %d: first stamp -> last stamp %.2f us; workgroup starts (us after the first): %s' % (
            li, (t5 - t0) * tick, ' '.join('%.2f' % ((t[l][w][0] - t0) * tick) for w in range(nwg)))
        print(line)
        if li > 0:
            prev = order[li - 1]
            print('   gap to the previous launch (its last stamp -> this first stamp): %.2f us' %
                  ((t0 - max(t[prev][w][5] for w in range(nwg))) * tick))
    l = order[-1]
    for w in range(nwg):
        print('workgroup %3d: weights in LDS %.2f, statistics final %.2f, rows staged %.2f |' % (
            32 * w, (t[l][w][6] - t[l][w][0]) * tick, (t[l][w][7] - t[l][w][6]) * tick, (t[l][w][1] - t[l][w][7]) * tick), end=' ')
        print('workgroup %3d: %s   total %.2f us' % (32 * w, '  '.join(
            '%s %.2f' % (nm.split()[0], (t[l][w][i + 1] - t[l][w][i]) * tick) for i, nm in enumerate(names)),
            (t[l][w][5] - t[l][w][0]) * tick))


if __name__ == '__main__':
    main()
