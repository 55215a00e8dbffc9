"""Generates the golden fixtures under tests/golden/ from the CPU oracle (fp64 master, stored
as fp32).  The reference has no fixtures of its own for this path and cannot be executed
(SURVEY F2/F9), so these vectors pin the ORACLE against drift and give the GPU tests inputs and
expected outputs that do not depend on /root/reference or on re-running the oracle.

    python tests/golden/make_golden.py          # rewrites the .npz files (deterministic)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from feta_tmlr_amd.transformer import data as D                     # noqa: E402
from feta_tmlr_amd.transformer.models import DiffGraphTransformerGenGCN   # noqa: E402
from oracle import feta_oracle as O                                 # noqa: E402

F64 = torch.float64


def f32(t):
    return t.detach().to(torch.float32).numpy()


def model_step(name, shape, bsz, d, heads, layers, order, batch_norm, share, seed):
    """One full forward+backward of DiffGraphTransformerGenGCN: inputs, parameters, output,
    coefficients and every gradient."""
    in_dim = 12
    torch.manual_seed(seed)
    model = DiffGraphTransformerGenGCN(in_dim, 1, d, heads, dim_feedforward=2 * d, dropout=0.0,
                                       nb_layers=layers, batch_norm=batch_norm, filter_order=order,
                                       heads_share_graph=bool(share))
    with torch.no_grad():
        model.encoder.spectral_gnns.bias.normal_(0, 0.1)
        model.encoder.gcn.bias.normal_(0, 0.1)
    ds = D.SyntheticGraphDataset(shape, bsz, in_dim=in_dim, seed=seed)
    batch9, cache = D.collate(ds.samples)
    x, mask, pe, _, degree, labels, edge_index, batch, fi = batch9
    p64 = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()
           if v.dtype.is_floating_point}
    x64 = x.double().requires_grad_(True)
    out, coeff = O.graph_transformer_gengcn(x64, edge_index, batch, fi, mask, pe.double(), degree.double(),
                                            p64, num_layers=layers, num_heads=heads, order=order,
                                            batch_norm=batch_norm, heads_share_graph=bool(share))
    w = torch.linspace(0.5, 1.5, out.numel(), dtype=F64).view_as(out)
    ((out * w).sum() + 0.01 * coeff.pow(2).sum()).backward()
    arrays = {'x': f32(x), 'mask': mask.numpy(), 'pe': f32(pe), 'degree': f32(degree),
              'edge_index': edge_index.numpy(), 'batch': batch.numpy(), 'feature_indices': fi.numpy(),
              'n_real': cache.n_real.numpy(), 'out': f32(out), 'coeff': f32(coeff), 'loss_w': f32(w),
              'dx': f32(x64.grad),
              'cfg': np.array([d, heads, layers, order, int(batch_norm), int(share), in_dim])}
    for k, v in p64.items():
        arrays['param/' + k] = f32(v)
        if v.grad is not None:
            arrays['grad/' + k] = f32(v.grad)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)


def filter_vectors(name, shape, bsz, heads, dh, order, seed, n_min=None, n_max=None):
    """Filter stage only: x, coeff, bias, graph -> y and gradients, both head modes."""
    ds = D.SyntheticGraphDataset(shape, bsz, in_dim=4, seed=seed, n_min=n_min, n_max=n_max)
    batch9, cache = D.collate(ds.samples)
    mask, edge_index, batch, fi = batch9[1], batch9[6], batch9[7], batch9[8]
    n = mask.shape[1]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(bsz, n, heads, dh, generator=g, dtype=F64) * (~mask)[:, :, None, None]
    coeff = torch.randn(heads, bsz, order * dh * dh, generator=g, dtype=F64) / dh ** 0.5
    bias = 0.1 * torch.randn(dh, generator=g, dtype=F64)
    dy = torch.randn(bsz, n, heads, dh, generator=g, dtype=F64)
    arrays = {'x': f32(x), 'coeff': f32(coeff), 'bias': f32(bias), 'dy': f32(dy), 'mask': mask.numpy(),
              'edge_index': edge_index.numpy(), 'batch': batch.numpy(), 'feature_indices': fi.numpy(),
              'n_real': cache.n_real.numpy(), 'node_off': cache.node_off.numpy(),
              'cfg': np.array([heads, dh, order])}
    for share in (0, 1):
        xr, cr, br = (t.clone().requires_grad_(True) for t in (x, coeff, bias))
        y = O.filter_stage_faithful(xr, cr, edge_index, fi, batch, br, order, (n, bsz, heads * dh), bool(share))
        y = y.view(n, bsz, heads, dh).permute(1, 0, 2, 3)
        (y * dy).sum().backward()
        arrays.update({'y%d' % share: f32(y), 'dx%d' % share: f32(xr.grad),
                       'dcoeff%d' % share: f32(cr.grad), 'dbias%d' % share: f32(br.grad)})
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)


def train_step(name, task, seed, batch_norm, mode='cheb', lr=1e-3):
    """One full optimisation step of a task shell: batch tuple + parameters ->
    loss, model output, the L2 norm of every parameter gradient and of every parameter after one
    Adam/AdamW update (fp64 oracle, torch CPU optimiser)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import train_checks as TC
    from feta_tmlr_amd import train as T
    model, batch9, cache = TC.build_case(task, torch.device('cpu'), seed=seed, bsz=6, d=16, heads=2,
                                         layers=2, order=2, batch_norm=batch_norm, mode=mode)
    p64 = TC.params64(model)
    out, loss, coeff, _ = TC.oracle_forward(task, model, batch9, p64, batch_norm)
    loss.backward()
    arrays = {'loss': f32(loss), 'out': f32(out), 'coeff': f32(coeff),
              'cfg': np.array([16, 2, 2, 2, int(batch_norm), 1]), 'lr': np.array(lr)}
    for k, v in zip(('x', 'mask', 'pe', 'lap_pe', 'degree', 'labels', 'edge_index', 'batch',
                     'feature_indices'), batch9):
        if v is not None:
            arrays['batch/' + k] = v.numpy()
    if cache.u is not None:
        arrays['cache/u'], arrays['cache/lam'] = cache.u.numpy(), cache.lam.numpy()
    for k, v in p64.items():
        arrays['param/' + k] = f32(v)
        if v.grad is not None:
            arrays['gnorm/' + k] = np.array(float(v.grad.norm()))
    opt = T.make_optimizer(task, [v for v in p64.values() if v.grad is not None], lr=lr)
    opt.step()
    for k, v in p64.items():
        if v.grad is not None:
            arrays['pnorm_after/' + k] = np.array(float(v.detach().norm()))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)


def main():
    train_step('step_zinc_bn', 'zinc', 21, True)
    train_step('step_tu', 'tu', 22, False)
    train_step('step_molhiv_spectral', 'molhiv', 23, False, mode='spectral')
    train_step('step_sbm', 'sbm', 24, False)
    model_step('model_mutag_b4', 'mutag', 4, 16, 2, 2, 4, False, 0, seed=11)
    model_step('model_zinc_b8_bn', 'zinc', 8, 32, 4, 2, 4, True, 0, seed=12)
    filter_vectors('filter_zinc_b8', 'zinc', 8, 4, 16, 4, seed=13)
    filter_vectors('filter_pattern_n120', 'pattern', 1, 2, 16, 4, seed=14, n_min=120, n_max=120)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
