"""Shared checks of the task shells (MolHiv / SBM / graph-level) and of one optimisation step against
the oracle; used by the emulation suite (CPU) and the MI355X suite."""
import torch

import kernel_checks as KC
from feta_tmlr_amd import train as T
from feta_tmlr_amd.transformer import data as D
from feta_tmlr_amd.transformer import models as M
from oracle import feta_oracle as O


def build_case(task, dev, seed=0, bsz=4, d=32, heads=2, layers=2, order=3, batch_norm=False,
               share=1, mode='cheb', nb_class=3, lap_dim=0):
    """-> model (on dev), batch9, cache (on dev), oracle forward closure.
    lap_dim > 0: Laplacian eigenvector node features (``--lappe --lap-dim``, BASELINE config 5) through the
    shells' ``embedding_lap_pos_enc`` branch (transformer/models.py:523-526)."""
    torch.manual_seed(seed)
    kw = dict(dim_feedforward=2 * d, dropout=0.0, nb_layers=layers, batch_norm=batch_norm,
              filter_order=order, heads_share_graph=bool(share), filter_mode=mode,
              lap_pos_enc=lap_dim > 0, lap_pos_enc_dim=lap_dim)
    if task == 'molhiv':
        model = M.DiffGraphTransformerGenGCNMolHiv(9, 1, d, heads, **kw)
        ds = D.SyntheticGraphDataset('mutag', bsz, seed=seed, n_min=4, n_max=17, features='atom',
                                     labels='binary', nan_label_frac=0.3)
        ds.samples[0].y = float('nan')      # at least one unlabeled graph
        ds.samples[1].y = 1.0
        if len(ds.samples) > 2:
            ds.samples[2].y = 0.0           # both classes present
    elif task == 'sbm':
        model = M.DiffGraphTransformerGenGCNSBM(5, nb_class, d, heads, **kw)
        ds = D.SyntheticGraphDataset('pattern', bsz, in_dim=5, seed=seed, n_min=6, n_max=21,
                                     labels='node')
        for g in ds.samples:
            g.y = g.y % nb_class
    elif task == 'tu':
        model = M.DiffGraphTransformerGenGCN(7, nb_class, d, heads, **kw)
        ds = D.SyntheticGraphDataset('mutag', bsz, in_dim=7, seed=seed, n_min=4, n_max=17,
                                     labels='class', nb_class=nb_class)
    else:
        model = M.DiffGraphTransformerGenGCN(7, 1, d, heads, **kw)
        ds = D.SyntheticGraphDataset('zinc', bsz, in_dim=7, seed=seed, n_min=4, n_max=17)
    with torch.no_grad():
        model.encoder.spectral_gnns.bias.normal_(0, 0.1)
        model.encoder.gcn.bias.normal_(0, 0.1)
    if lap_dim > 0:
        from feta_tmlr_amd.transformer.position_encoding import LapEncoding
        LapEncoding(lap_dim, normalization='sym').apply_to(ds)
        with torch.no_grad():
            model.embedding_lap_pos_enc.bias.normal_(0, 0.1)
    n_pad = max(g.num_nodes for g in ds.samples)
    batch9, cache = D.collate(ds.samples, k_eig=n_pad if mode == 'spectral' else None, device=dev)
    model = model.to(dev)
    return model, batch9, cache


def oracle_forward(task, model, batch9, p64, batch_norm=False, share=1):
    x, mask, pe, lap, degree, labels, edge_index, batch, fi = (None if t is None else t.cpu() for t in batch9)
    kw = dict(num_layers=len(model.encoder.layers), num_heads=model.encoder.num_heads,
              order=model.encoder.order, batch_norm=batch_norm, heads_share_graph=bool(share),
              x_lap_pos_enc=None if lap is None else lap.double())
    pe64 = None if pe is None else pe.double()
    dg64 = None if degree is None else degree.double()
    if task == 'molhiv':
        logit, prob, coeff = O.graph_transformer_gengcn_molhiv(x, edge_index, batch, fi, mask, pe64,
                                                               dg64, p64, **kw)
        return logit, O.molhiv_loss(logit, labels.double()), coeff, prob
    if task == 'sbm':
        logit, coeff = O.graph_transformer_gengcn_sbm(x.double(), edge_index, batch, fi, mask, pe64,
                                                      dg64, p64, **kw)
        return logit, torch.nn.functional.cross_entropy(logit, labels), coeff, None
    out, coeff = O.graph_transformer_gengcn(x.double(), edge_index, batch, fi, mask, pe64, dg64, p64, **kw)
    if task == 'tu':
        return out, torch.nn.functional.cross_entropy(out, labels.view(-1)), coeff, None
    return out, torch.nn.functional.l1_loss(out, labels.double().view(out.shape)), coeff, None


def params64(model):
    return {k: v.detach().cpu().double().clone().requires_grad_(True)
            for k, v in model.state_dict().items() if v.dtype.is_floating_point}


def check_task_step(task, dev, run_ctx, batch_norm=False, mode='cheb', lr=1e-3, lap_dim=0):
    """forward output, loss, every parameter gradient and the parameters after ONE optimiser step
    (Adam / AdamW as the reference scripts configure them) against the fp64 oracle."""
    model, batch9, cache = build_case(task, dev, batch_norm=batch_norm, mode=mode, lap_dim=lap_dim)
    p64 = params64(model)
    crit = T.make_criterion(task, nb_class=3 if task in ('tu', 'sbm') else 1)
    opt = T.make_optimizer(task, model.parameters(), lr=lr)
    with run_ctx():
        loss, out = T.task_loss(task, model, crit, batch9, cache)
        out_ref, loss_ref, coeff_ref, prob_ref = oracle_forward(task, model, batch9, p64, batch_norm)
        KC.assert_close(task + ' output', out.cpu(), out_ref)
        KC.assert_close(task + ' loss', loss.cpu(), loss_ref)
        loss.backward()
    loss_ref.backward()
    names = dict(model.named_parameters())
    for k, p in names.items():
        g_ref = p64[k].grad
        if p.grad is None:
            assert g_ref is None or float(g_ref.abs().max()) == 0.0, k
            continue
        KC.assert_close('grad ' + k, p.grad.cpu(), g_ref, tol=3e-5)
    # one optimiser step on both sides (the oracle side: torch's CPU Adam/AdamW in fp64)
    before = {k: p.detach().cpu().double().clone() for k, p in names.items()}
    ref_params = [p64[k] for k in names if p64[k].grad is not None]
    opt_ref = T.make_optimizer(task, ref_params, lr=lr)
    opt_ref.step()
    opt.step()
    # Adam's first update is lr * g / (|g| + 1e-8): where the gradient is rounding noise (a bias in
    # front of a BatchNorm has an exactly-zero gradient) its SIGN is noise too, so elements are
    # compared where the gradient is significant and bounded by one lr step elsewhere
    for k, p in names.items():
        got, ref = p.detach().cpu().double(), p64[k].detach()
        g_ref = p64[k].grad
        if g_ref is None:
            assert torch.equal(got, before[k]), k
            continue
        sig = g_ref.abs() > 1e-3 * max(1.0, float(g_ref.abs().max()))
        if sig.any():
            KC.assert_close('param after step ' + k, got[sig], ref[sig], tol=3e-5)
        assert float((got - before[k]).abs().max()) <= lr * 1.01 + abs(lr) * 1e-4 * float(before[k].abs().max()), k
    return float(loss.detach())


def check_molhiv_outputs(dev, run_ctx):
    """3-tuple (logits, reg, sigmoid(logits)) of the molhiv shell, the max-cosine regulariser, and
    the LeakyReLU(True) == identity quirk (oracle applies negative_slope 1.0)."""
    model, batch9, cache = build_case('molhiv', dev)
    x, mask, pe, _, degree, labels, edge_index, batch, fi = batch9
    with run_ctx(), torch.no_grad():
        logit, reg, prob, coeff = model(x, edge_index, batch, fi, mask, pe, degree=degree,
                                        regularization=1.0, return_filter_coeff=True, graph_cache=cache)
    p64 = params64(model)
    with torch.no_grad():
        logit_ref, _, coeff_ref, prob_ref = oracle_forward('molhiv', model, batch9, p64)
    assert logit.shape == (x.shape[0],)
    KC.assert_close('molhiv logits', logit.cpu(), logit_ref)
    KC.assert_close('molhiv sigmoid', prob.cpu(), prob_ref)
    KC.assert_close('molhiv regulariser', reg.cpu(), O.regularisation_max_cos(coeff_ref), tol=1e-4)


def check_sbm_padded_equals_gather(dev, run_ctx):
    """the capturable padded-logits loss equals the reference's boolean-gather loss"""
    model, batch9, cache = build_case('sbm', dev)
    crit = T.make_criterion('sbm', nb_class=3)
    with run_ctx():
        l1, _ = T.task_loss('sbm', model, crit, batch9, cache)
        b = list(batch9)
        b[5] = T.pad_node_labels(batch9[5], batch9[8], batch9[0].shape[0], batch9[0].shape[1])
        l2, logits = T.task_loss('sbm', model, crit, tuple(b), cache, padded_node_labels=True)
    assert logits.dim() == 3
    KC.assert_close('padded vs gathered loss', l2.cpu(), l1.detach().cpu().double())
    # weighted loss of the reference's model.loss()
    with run_ctx():
        out, _ = model(*[batch9[i] for i in (0, 6, 7, 8, 1, 2)], degree=batch9[4], graph_cache=cache)
        lw = model.loss(out, batch9[5])
    KC.assert_close('weighted SBM loss', lw.cpu(), O.sbm_weighted_loss(out.detach().cpu().double(),
                                                                      batch9[5].cpu(), 3))


def check_config5_step(dev, run_ctx, bsz=320, n_min=20, n_max=64, layers=2):
    """BASELINE config 5 all at once (exercised piecewise until round 3): the molhiv shell
    (transformer/models.py:598-742) with ``--lappe --lap-dim 8``, d_model = 64 / 4 heads (the fused stack's shape), one
    N_pad <= 64 bucket of `bsz` graphs (320: more graphs than workgroups, the kernels walk - the fp64 oracle's cost grows
    quadratically with the batch, 1024 graphs take it minutes; bench.py times the 1024-graph bucket), on bf16 STORAGE (layers.set_storage_dtype: fused bf16 stack, fp32 statistics,
    parameter gradients and filter stage) - logits, loss and every parameter gradient of one step against the fp64 oracle
    on the same fp32 inputs and master weights.  Bars: those of the bf16 leg of the bench (tests/bench_checks.py)."""
    import bench_checks as BC
    from feta_tmlr_amd.transformer.layers import set_storage_dtype
    from feta_tmlr_amd.transformer.position_encoding import LapEncoding
    torch.manual_seed(0)
    d, heads, lap_dim = 64, 4, 8
    model = M.DiffGraphTransformerGenGCNMolHiv(9, 1, d, heads, dim_feedforward=2 * d, dropout=0.0, nb_layers=layers,
                                               batch_norm=False, filter_order=4, heads_share_graph=True,
                                               filter_mode='spectral', lap_pos_enc=True, lap_pos_enc_dim=lap_dim)
    ds = D.SyntheticGraphDataset('mutag', bsz, seed=0, n_min=n_min, n_max=n_max, features='atom', labels='binary',
                                 nan_label_frac=0.3)
    ds.samples[0].y, ds.samples[1].y, ds.samples[2].y = float('nan'), 1.0, 0.0
    with torch.no_grad():
        model.encoder.spectral_gnns.bias.normal_(0, 0.1)
        model.encoder.gcn.bias.normal_(0, 0.1)
        model.embedding_lap_pos_enc.bias.normal_(0, 0.1)
    LapEncoding(lap_dim, normalization='sym').apply_to(ds)
    n_pad = max(g.num_nodes for g in ds.samples)
    assert n_pad <= 64
    batch9, cache = D.collate(ds.samples, k_eig=16, device=dev)
    model = model.to(dev)
    p64 = params64(model)
    set_storage_dtype(model, torch.bfloat16)
    crit = T.make_criterion('molhiv', nb_class=1)
    with run_ctx():
        loss, out = T.task_loss('molhiv', model, crit, batch9, cache)
        loss.backward()
    # the K = 16 eigenbasis operator is the bench's operator, not a reference operator: its oracle is the
    # eigenbasis formulation on the very U, lambda the kernels read
    x, mask, pe, lap, degree, labels, edge_index, batch, fi = (None if t is None else t.cpu() for t in batch9)
    c = cache.to(torch.device('cpu'))
    logit, prob, coeff = O.graph_transformer_gengcn_molhiv(
        x, edge_index, batch, fi, mask, pe.double(), degree.double(), p64, num_layers=layers, num_heads=heads, order=4,
        batch_norm=False, heads_share_graph=True, x_lap_pos_enc=lap.double(), eig=(c.u.double(), c.lam.double()),
        collapsed=True)
    loss_ref = O.molhiv_loss(logit, labels.double())
    loss_ref.backward()
    errs = {'out': KC.assert_close('config 5 logits', out.float().cpu(), logit, tol=BC.BF16_MODEL_TOL),
            'loss': KC.assert_close('config 5 loss', loss.float().cpu(), loss_ref, tol=BC.BF16_MODEL_TOL)}
    for k, p in model.named_parameters():
        g_ref = p64[k].grad
        if p.grad is None:
            assert g_ref is None or float(g_ref.abs().max()) == 0.0, k
            continue
        if float(g_ref.norm()) < 1e-9:
            continue
        errs[k] = BC.rel_fro(p.grad, g_ref)
        assert errs[k] <= BC.bf16_grad_tol(k), 'config 5 gradient %s: relative Frobenius error %.3f' % (k, errs[k])
    return errs
