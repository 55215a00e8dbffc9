"""Phase timing inside the fused kernels of the layer stack: builds a diagnostic copy of the library with
-DFETA_TIMING into tools/_timing/ (git-ignored) whose kernels record s_memrealtime (100 MHz, one clock for the
whole chip) of wave 0 of every 32nd workgroup at their phase boundaries, over four consecutive launches
(FETA_RT_STAMP, csrc/feta_rowops.h).  Prints the launch-to-launch time of the kernel inside a hipGraph (what a
captured step pays), the start skew of the workgroups, the gap between consecutive launches and the time of every
phase.  Run on the GPU box:

    python tools/block_timing.py --kernel fwd            # attn_block_fwd as the stack issues it
    python tools/block_timing.py --kernel bwd [--split]   # attn_block_bwd, one / two workgroups per graph
"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feta_tmlr_amd import _abi                              # noqa: E402
from feta_tmlr_amd.benchcases import stack_layer_cases      # noqa: E402

OUT = os.path.join(ROOT, 'tools', '_timing', 'libfeta_timing.so')

KERNELS = {
    'fwd': ('attn_block_fwd (no attn write)', 'feta_debug_block_stamps',
            ['prologue loads, weights in LDS', 'statistics final', 'rows + pe staged', 'in_proj', 'attention core',
             'barrier', 'concat store + out_proj'], [0, 6, 7, 1, 2, 3, 4, 5]),
    'bwd': ('attn_block_bwd', 'feta_debug_bbwd_stamps',
            ['BatchNorm-1 sums final', 'graph tiles staged', 'dconcat', 'attention backward', 'barrier + dq/dk/dv tiles',
             'dx + sums', 'weight gradients'], [0, 1, 2, 3, 4, 5, 6, 7]),
}


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, 'feta_tmlr_amd', 'csrc', '*.hip')))
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DFETA_TIMING',
           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'feta_tmlr_amd', 'csrc')] + srcs + ['-o', OUT]
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--n-pad', type=int, default=37)
    ap.add_argument('--kernel', choices=sorted(KERNELS), default='fwd')
    ap.add_argument('--split', action='store_true', help='bwd: the two-workgroups-per-graph form')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='storage type of the token tensors')
    ap.add_argument('--build-only', action='store_true')
    ap.add_argument('--lib', default=OUT, help='a diagnostic build to load instead (A/B timing of an older source)')
    a = ap.parse_args()
    if not os.path.exists(OUT) or a.build_only:
        build()
    if a.build_only:
        return
    lib = ctypes.CDLL(a.lib)
    abi = _abi.bind(lib)
    dev = torch.device('cuda:0')
    b, n, d, h = a.batch, a.n_pad, 64, 4
    g = torch.Generator().manual_seed(0)
    nr = torch.randint(9, n + 1, (b,), generator=g, dtype=torch.int32).to(dev)
    pe = torch.rand(b, n, n, generator=g).to(dev)
    case, getter, names, order = KERNELS[a.kernel]
    split = a.kernel == 'bwd' and a.split
    if split:
        case = 'attn_block_bwd (two workgroups per graph)'
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        st = s2.cuda_stream
        fn = {nm: f for nm, _, f, _, _ in stack_layer_cases(abi, st, dev, b, n, d, h, 2 * d, pe, nr,
                                                           dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32)}[case]
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        # eight launches back to back inside one hipGraph (what a captured step issues)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s2):
            for _ in range(8):
                fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print('%s: launch-to-launch inside the graph %.2f us' % (case, e0.elapsed_time(e1) / 160 * 1e3))
    buf = (ctypes.c_ulonglong * 256)()
    getattr(lib, getter)(buf)
    t = [[[buf[(l * 8 + w) * 8 + i] for i in range(8)] for w in range(8)] for l in range(4)]
    grid = 2 * b if split else b
    nwg = min(8, (min(grid, 256) + 31) // 32)
    first, last = order[0], order[-1]
    ring = sorted(range(4), key=lambda l: t[l][0][first])      # the last four launches, oldest first
    tick = 0.01   # us per s_memrealtime tick (100 MHz)
    for li, l in enumerate(ring):
        t0 = min(t[l][w][first] for w in range(nwg))
        t5 = max(t[l][w][last] for w in range(nwg))
        print('launch %d: first stamp -> last stamp %.2f us; workgroup starts (us after the first): %s' % (
            li, (t5 - t0) * tick, ' '.join('%.2f' % ((t[l][w][first] - t0) * tick) for w in range(nwg))))
        if li > 0:
            prev = ring[li - 1]
            print('   gap to the previous launch (its last stamp -> this first stamp): %.2f us' %
                  ((t0 - max(t[prev][w][last] for w in range(nwg))) * tick))
    l = ring[-1]
    for w in range(nwg):
        print('workgroup %3d (n = %2d):' % (32 * w, int(nr[(32 * w) // (2 if split else 1)])),
              ' | '.join('%s %.2f' % (nm, (t[l][w][order[i + 1]] - t[l][w][order[i]]) * tick) for i, nm in enumerate(names)),
              ' | total %.2f us' % ((t[l][w][last] - t[l][w][first]) * tick))


if __name__ == '__main__':
    main()
