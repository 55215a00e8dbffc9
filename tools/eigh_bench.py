"""Spectrum producer on the MI355X against the host path it replaces: edge list -> Lhat -> eigh ->
diffusion kernel for one collated batch (feta_lhat_from_edges + feta_eigh_sym + feta_spectral_kernel,
three launches) vs numpy.linalg.eigh + U exp(-lam) U^T graph by graph (what transformer/data.py does
per sample; the reference uses np.linalg.eig and scipy expm per graph, position_encoding.py:65-72,137)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feta_tmlr_amd import functional as FF           # noqa: E402
from feta_tmlr_amd.transformer import data as D      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cases', default='zinc:128,zinc:1024,pattern:64,molhiv64:1024')
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda:0')
for case in args.cases.split(','):
    shape, bsz = case.split(':')
    bsz = int(bsz)
    kw = dict(n_max=64) if shape == 'molhiv64' else {}
    ds = D.SyntheticGraphDataset(shape.replace('64', ''), bsz, in_dim=2, seed=0, pos_enc=False, with_eig=False, **kw)
    t0 = time.perf_counter()
    for g in ds.samples:
        lam, u = np.linalg.eigh(D.lhat_numpy(g.edge_index, g.num_nodes))
        pe = (u * np.exp(-(lam + 1.0))) @ u.T
    host = time.perf_counter() - t0
    b9, cache = D.collate(ds.samples, device=dev)
    ei, bat = b9[6], b9[7]
    n = cache.n_pad

    def run():
        lhat = FF.lhat_from_edges(ei, bat, cache.node_off, bsz, n)
        u, lam, sw = FF.eigh_sym(lhat, cache.n_real, 2.0, return_sweeps=True)
        pe = FF.spectral_kernel(u, lam, cache.n_real, 'diffusion', lam_offset=1.0)
        return lhat, u, lam, sw, pe
    for _ in range(3):
        out = run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tot = [0.0, 0.0, 0.0]
    for _ in range(args.iters):
        ev[0].record()
        lhat = FF.lhat_from_edges(ei, bat, cache.node_off, bsz, n)
        ev[1].record()
        u, lam = FF.eigh_sym(lhat, cache.n_real, 2.0)
        ev[2].record()
        pe = FF.spectral_kernel(u, lam, cache.n_real, 'diffusion', lam_offset=1.0)
        ev[3].record()
        torch.cuda.synchronize()
        for i in range(3):
            tot[i] += ev[i].elapsed_time(ev[i + 1])
    ms = [t / args.iters for t in tot]
    print('%-10s B=%5d N_pad=%3d  host numpy %8.1f ms (%6.1f us/graph) | device lhat %.3f + eigh %.3f + kernel %.3f = %.3f ms '
          '(%.2f us/graph, %.0fx)  sweeps max %d'
          % (shape, bsz, n, host * 1e3, host * 1e6 / bsz, ms[0], ms[1], ms[2], sum(ms), sum(ms) * 1e3 / bsz,
             host * 1e3 / sum(ms), int(out[3].max())), flush=True)
