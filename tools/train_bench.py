"""Whole optimisation steps per second (row H1 of SURVEY 8a): forward + loss + backward + Adam update of the
ZINC task shell (DiffGraphTransformerGenGCN: embedding, encoder stack + filter stage, pooling, classifier; L1
loss, Adam, experiments/run_transformer_gengcn.py:115-164,310-317) on a BASELINE-shaped synthetic batch, as ONE
captured hipGraph (train.GraphedTrainStep) and eagerly (train.train_step).  bench.py times the metric
BASELINE.json names (fwd+bwd of the encoder path); this is the number a training loop sees."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feta_tmlr_amd import train as T                                # noqa: E402
from feta_tmlr_amd.transformer import data as D                     # noqa: E402
from feta_tmlr_amd.transformer.models import DiffGraphTransformerGenGCN   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--layer-norm', action='store_true')
ap.add_argument('--dropout', type=float, default=0.0,
                help='--dropout of the reference scripts (experiments/run_transformer_gengcn.py:47): attention-probability and '
                     'activation dropout; the layers then run op by op (the fused stack has no dropout), the attention masks '
                     'are keyed on the device so that the step stays ONE hipGraph')
ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
ds = D.SyntheticGraphDataset('zinc', args.batch, in_dim=28, seed=0)
batch9, cache = D.collate(ds.samples, k_eig=16, n_pad=37, device=dev)
model = DiffGraphTransformerGenGCN(28, 1, 64, 4, dim_feedforward=128, dropout=args.dropout, nb_layers=3,
                                   batch_norm=not args.layer_norm, filter_order=4, heads_share_graph=True,
                                   filter_mode='spectral').to(dev)
if args.dtype == 'bf16':
    from feta_tmlr_amd.transformer.layers import set_storage_dtype
    set_storage_dtype(model, torch.bfloat16)
model.train()
crit = T.make_criterion('zinc')


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps


opt = T.make_optimizer('zinc', model.parameters(), lr=1e-3)
gc = T.prepare_cache(model, batch9, cache)
eager = timed(lambda: T.train_step('zinc', model, crit, opt, batch9, gc, lr=1e-3))
opt_g = T.make_optimizer('zinc', model.parameters(), lr=1e-3, capturable=True)
graphed = T.GraphedTrainStep('zinc', model, crit, opt_g, batch9, cache)
graphed.set_lr(1e-3)
cap = timed(lambda: graphed(batch9, cache))
# ---- N3: what it costs to put the NEXT batch on the device ------------------------------------------
big = D.SyntheticGraphDataset('zinc', 8 * args.batch, in_dim=28, seed=1, pos_enc=True, with_eig=True)
rng = __import__('numpy').random.default_rng(0)
id_sets = [rng.choice(len(big), size=args.batch, replace=False) for _ in range(20)]


def host_collate():
    for ids in id_sets:      # the reference's shape: per-graph loop, dense pe / U built on the host, pageable copies
        D.collate([big[i] for i in ids], k_eig=16, n_pad=37, device=dev)
    torch.cuda.synchronize()


packed = D.PackedGraphs(big.samples)
stager = D.BatchStager(packed, args.batch, 37, dev, pos_enc='diffusion', k_eig=16)


def staged():
    for ids in id_sets:      # vectorised fill of pinned buffers, async copies, pe / U / lambda produced on the device
        stager.stage(ids)
    torch.cuda.synchronize()


def timed_host(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps / len(id_sets)


t_coll, t_stage = timed_host(host_collate), timed_host(staged)
print('collate of one ZINC batch (B=%d, pe + U/lambda included): per-graph host collate %.1f us/graph | pinned stager + '
      'device spectrum %.2f us/graph (%.0fx); step time per graph %.2f us'
      % (args.batch, t_coll / args.batch * 1e6, t_stage / args.batch * 1e6, t_coll / t_stage, cap / args.batch * 1e6))
print('ZINC task, B=%d, %s, dropout %.2f, %s: eager %.3f ms/step (%.0f graphs/s) | one hipGraph per step %.3f ms/step (%.0f graphs/s)'
      % (args.batch, 'LayerNorm' if args.layer_norm else 'BatchNorm', args.dropout, args.dtype, eager * 1e3, args.batch / eager,
         cap * 1e3, args.batch / cap))
