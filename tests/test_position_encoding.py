"""Encodings derived from one eigendecomposition (feta_tmlr_amd/transformer/position_encoding.py)
against the reference's definitions (transformer/position_encoding.py:55-161) evaluated directly."""
import numpy as np
import pytest
import scipy.linalg

from feta_tmlr_amd.transformer import data as D
from feta_tmlr_amd.transformer import position_encoding as PE


def _ds(n=6, seed=0):
    return D.SyntheticGraphDataset('zinc', n, in_dim=4, seed=seed, pos_enc=False, with_eig=False)


@pytest.mark.parametrize('norm', [None, 'sym', 'rw'])
def test_diffusion_matches_expm(norm):
    ds = _ds()
    PE.DiffusionEncoding(None, beta=0.7, normalization=norm).apply_to(ds)
    for g in ds:
        ref = scipy.linalg.expm(-0.7 * PE.laplacian_dense(g.edge_index, g.num_nodes, norm))
        assert np.abs(g.pe - ref).max() < 1e-5


@pytest.mark.parametrize('norm', [None, 'sym', 'rw'])
def test_pstep_matches_matrix_power(norm):
    ds = _ds()
    PE.PStepRWEncoding(None, p=3, beta=0.25, normalization=norm).apply_to(ds)
    for g in ds:
        m = np.eye(g.num_nodes) - 0.25 * PE.laplacian_dense(g.edge_index, g.num_nodes, norm)
        assert np.abs(g.pe - np.linalg.matrix_power(m, 3)).max() < 1e-5


def test_lap_encoding_columns_are_eigenvectors_in_ascending_order():
    ds = _ds()
    PE.LapEncoding(5, normalization='sym').apply_to(ds)
    for g in ds:
        n = g.num_nodes
        lap = PE.laplacian_dense(g.edge_index, n, 'sym')
        lam = np.linalg.eigvalsh(lap)
        assert g.lap_pe.shape == (n, 5)
        for c in range(min(5, n - 1)):
            v = g.lap_pe[:, c].astype(np.float64)
            assert np.abs(lap @ v - lam[c + 1] * v).max() < 1e-5   # first eigenvector dropped


def test_lap_encoding_zero_pads_small_graphs():
    ds = D.SyntheticGraphDataset('zinc', 2, in_dim=4, seed=1, pos_enc=False, with_eig=False, n_min=3, n_max=3)
    PE.LapEncoding(8, normalization='sym').apply_to(ds)
    for g in ds:
        assert g.lap_pe.shape == (3, 8) and float(np.abs(g.lap_pe[:, 2:]).max()) == 0.0


def test_spectral_encoding_reconstructs_lhat_and_feeds_collate():
    ds = _ds(4)
    PE.SpectralEncoding().apply_to(ds)
    for g in ds:
        lhat = D.lhat_numpy(g.edge_index, g.num_nodes)
        assert np.abs((g.u * g.lam) @ g.u.T - lhat).max() < 1e-12
        assert g.lam.min() >= -1 - 1e-12 and g.lam.max() <= 1 + 1e-12
    _, cache = D.collate(ds.samples, k_eig=8)
    assert cache.u.shape[2] == 8 and cache.lam.shape == (4, 8)


def test_adj_full_and_cache(tmp_path):
    ds = _ds(3)
    PE.AdjEncoding(None).apply_to(ds)
    g = ds[0]
    assert g.pe.sum() == g.edge_index.shape[1]
    enc = PE.DiffusionEncoding(str(tmp_path / 'pe'), beta=1.0, normalization='sym', zero_diag=True)
    enc.apply_to(ds, split='train')
    first = [g.pe.copy() for g in ds]
    assert all(float(np.abs(np.diag(p)).max()) == 0.0 for p in first)
    enc.apply_to(ds, split='train')     # second call is served from the .npz cache
    assert all(np.array_equal(a, g.pe) for a, g in zip(first, ds))
    assert PE.POSENCODINGS['diffusion'] is PE.DiffusionEncoding


def test_bucket_batches_cover_every_graph_once():
    ds = D.SyntheticGraphDataset('molhiv', 200, in_dim=4, seed=0, pos_enc=False, with_eig=False)
    batches = D.bucket_batches(ds.samples, 32)
    seen = sorted(i for _, idx in batches for i in idx)
    assert seen == list(range(200))
    for npad, idx in batches:
        assert npad in D.BUCKETS and len(idx) <= 32
        assert all(ds[i].num_nodes <= npad for i in idx)
        smaller = [bk for bk in D.BUCKETS if bk < npad]
        if smaller:
            assert all(ds[i].num_nodes > smaller[-1] for i in idx)
