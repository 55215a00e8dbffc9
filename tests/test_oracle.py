"""Pins the CPU oracle (the reference has no golden vectors for this path - SURVEY F9): three
independent formulations of the filter agree in fp64, the collapsed coefficient generator equals
the un-collapsed restatement of the reference's algorithm, and closed-form known answers."""
import numpy as np
import pytest
import torch

from feta_tmlr_amd.transformer import data as D
from oracle import feta_oracle as O

F64 = torch.float64


def _graph(n, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(D.molecule_graph(rng, n))


@pytest.mark.parametrize('n,order,seed', [(23, 4, 0), (9, 5, 1), (37, 4, 2), (2, 3, 3)])
def test_three_formulations_agree(n, order, seed):
    g = torch.Generator().manual_seed(seed)
    ei = _graph(n, seed)
    x = torch.randn(n, 16, generator=g, dtype=F64)
    w = torch.randn(order, 16, 16, generator=g, dtype=F64)
    bias = torch.randn(16, generator=g, dtype=F64)
    y1 = O.cheb_conv_dynamic_edges(x, ei, w.unsqueeze(1), torch.zeros(n, dtype=torch.long), bias)
    lh = O.lhat_dense(ei, n, F64)
    y2 = O.cheb_filter_dense(x, lh, w, bias)
    u, lam = O.eig_basis(lh, n)
    y3 = O.spec_filter_eig(x, u, lam, w, bias)
    assert (y1 - y2).abs().max() < 1e-12
    assert (y1 - y3).abs().max() < 1e-11


def test_truncated_eigenbasis_is_a_different_operator():
    """K < n does not reproduce the reference operator (hence parity runs use K = N)."""
    n, order = 23, 4
    g = torch.Generator().manual_seed(0)
    ei = _graph(n, 0)
    x = torch.randn(n, 16, generator=g, dtype=F64)
    w = torch.randn(order, 16, 16, generator=g, dtype=F64)
    lh = O.lhat_dense(ei, n, F64)
    y = O.cheb_filter_dense(x, lh, w, None)
    u, lam = O.eig_basis(lh, 16)
    assert (O.spec_filter_eig(x, u, lam, w, None) - y).abs().max() > 1e-2


def test_lhat_is_minus_normalised_adjacency():
    n = 12
    ei = _graph(n, 4)
    a = torch.zeros(n, n, dtype=F64)
    a[ei[0], ei[1]] = 1.0
    dis = a.sum(1).pow(-0.5)
    ref = -(dis[:, None] * a * dis[None, :])
    assert (O.lhat_dense(ei, n, F64) - ref).abs().max() < 1e-14


def test_lhat_numpy_matches_oracle():
    for seed in range(3):
        ei = _graph(17, seed)
        assert np.abs(D.lhat_numpy(ei.numpy(), 17) - O.lhat_dense(ei, 17, F64).numpy()).max() < 1e-14


def test_empty_graph_known_answer():
    """No edges: L_hat = 0, T_k = 1,0,-1,0,... so out = X W0 - X W2 + bias."""
    n = 7
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 8, generator=g, dtype=F64)
    w = torch.randn(4, 8, 8, generator=g, dtype=F64)
    bias = torch.randn(8, generator=g, dtype=F64)
    ei = torch.zeros(2, 0, dtype=torch.long)
    y = O.cheb_conv_dynamic_edges(x, ei, w.unsqueeze(1), torch.zeros(n, dtype=torch.long), bias)
    assert (y - (x @ w[0] - x @ w[2] + bias)).abs().max() < 1e-14


def test_order_one_known_answer():
    n = 9
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 8, generator=g, dtype=F64)
    w = torch.randn(1, 8, 8, generator=g, dtype=F64)
    y = O.cheb_conv_dynamic_edges(x, _graph(n, 0), w.unsqueeze(1), torch.zeros(n, dtype=torch.long), None)
    assert (y - x @ w[0]).abs().max() < 1e-14


def test_path_graph_spectrum_known_answer():
    """P_n: eigenvalues of D^-1/2 A D^-1/2 are cos(pi k/(n-1)), so lambda_hat = -cos(pi k/(n-1))."""
    n = 8
    src = torch.arange(n - 1)
    ei = torch.cat([torch.stack([src, src + 1]), torch.stack([src + 1, src])], dim=1)
    _, lam = O.eig_basis(O.lhat_dense(ei, n, F64), n)
    ref = np.sort(-np.cos(np.pi * np.arange(n) / (n - 1)))
    assert np.abs(lam.numpy() - ref).max() < 1e-12


def test_directed_edges_keep_source_to_target_flow():
    """propagate() adds x[source] into target; a transposed restatement fails this."""
    ei = torch.tensor([[0, 1], [1, 2]])          # 0 -> 1, 1 -> 2
    lh = O.lhat_dense(ei, 3, F64)
    # deg (scattered on the source row) = [1, 1, 0]; w(0->1) = -1, w(1->2) = -(1 * 0) = 0
    ref = torch.zeros(3, 3, dtype=F64)
    ref[1, 0] = -1.0
    assert torch.equal(lh, ref)


@pytest.mark.parametrize('zero_diag', [False, True])
def test_collapsed_coefficients_equal_faithful(zero_diag):
    g = torch.Generator().manual_seed(0)
    bsz, h, n, c = 3, 2, 11, 24
    nb = torch.tensor([11, 5, 8])
    mask = torch.arange(n)[None, :] >= nb[:, None]
    a = torch.rand(bsz, h, n, n, generator=g, dtype=F64).masked_fill(mask[:, None, None, :], 0.0)
    if zero_diag:
        a[:, 0].diagonal(dim1=-2, dim2=-1).zero_()
    a = a / a.sum(-1, keepdim=True)
    gw = torch.randn(c, c, generator=g, dtype=F64)
    gb = torch.randn(c, generator=g, dtype=F64)
    lw = torch.randn(c, c, generator=g, dtype=F64)
    lb = torch.randn(c, generator=g, dtype=F64)
    f = O.get_filter_coefficients_faithful(a, mask, gw, gb, lw, lb)
    k = O.get_filter_coefficients_collapsed(a, mask, gw, gb, lw, lb)
    assert f.shape == (h, bsz, c)
    assert (f - k).abs().max() < 1e-12


def test_uniform_attention_known_answer():
    """attn = 1/n on the real block: w_ij = 1/n, deg_j = 1, c_j = 1 => pooled = tanh(colsum(W) + b)."""
    n, c = 6, 10
    mask = torch.zeros(1, n, dtype=torch.bool)
    a = torch.full((1, 1, n, n), 1.0 / n, dtype=F64)
    g = torch.Generator().manual_seed(1)
    gw = torch.randn(c, c, generator=g, dtype=F64)
    gb = torch.randn(c, generator=g, dtype=F64)
    out = O.get_filter_coefficients_faithful(a, mask, gw, gb, torch.eye(c, dtype=F64), torch.zeros(c, dtype=F64))
    assert (out[0, 0] - torch.tanh(gw.sum(0) + gb)).abs().max() < 1e-13


def test_attention_rows_normalised_and_masked():
    g = torch.Generator().manual_seed(0)
    n, b, h, dh = 9, 2, 2, 4
    qkv = torch.randn(n, b, 3 * h * dh, generator=g, dtype=F64)
    mask = torch.arange(n)[None, :] >= torch.tensor([9, 4])[:, None]
    _, a, oh = O.attention_core(qkv, None, mask, h)
    assert a.shape == (b, h, n, n) and oh.shape == (b, n, h, dh)
    assert (a.sum(-1) - 1).abs().max() < 1e-12
    assert float(a[1, :, :, 4:].abs().max()) == 0.0


def test_heads_beyond_zero_see_no_graph_in_literal_mode():
    """with the un-replicated edge_index only head 0 is filtered on the graph."""
    ds = D.SyntheticGraphDataset('mutag', 2, in_dim=4, seed=0, n_min=5, n_max=9)
    (b9, cache) = D.collate(ds.samples)
    mask, ei, batch, fi = b9[1], b9[6], b9[7], b9[8]
    bsz, n, h, dh, order = 2, mask.shape[1], 2, 4, 3
    g = torch.Generator().manual_seed(0)
    oh = torch.randn(bsz, n, h, dh, generator=g, dtype=F64)
    coeff = torch.randn(h, bsz, order * dh * dh, generator=g, dtype=F64)
    y = O.filter_stage_faithful(oh, coeff, ei, fi, batch, None, order, (n, bsz, h * dh), False)
    y = y.view(n, bsz, h, dh)
    for b in range(bsz):
        m = int(cache.n_real[b])
        w = coeff[1, b].reshape(order, dh, dh)
        ref = oh[b, :m, 1] @ (w[0] - w[2])
        assert (y[:m, b, 1] - ref).abs().max() < 1e-13
        assert float(y[m:, b].abs().max() if m < n else 0.0) == 0.0
