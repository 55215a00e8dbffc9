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
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
ds = D.SyntheticGraphDataset('zinc', args.batch, in_dim=28, seed=0)
batch9, cache = D.collate(ds.samples, k_eig=16, n_pad=37, device=dev)
model = DiffGraphTransformerGenGCN(28, 1, 64, 4, dim_feedforward=128, dropout=0.0, nb_layers=3,
                                   batch_norm=not args.layer_norm, filter_order=4, heads_share_graph=True,
                                   filter_mode='spectral').to(dev)
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
print('ZINC task, B=%d, %s: eager %.3f ms/step (%.0f graphs/s) | one hipGraph per step %.3f ms/step (%.0f graphs/s)'
      % (args.batch, 'LayerNorm' if args.layer_norm else 'BatchNorm', eager * 1e3, args.batch / eager,
         cap * 1e3, args.batch / cap))
