"""Per-graph positional / spectral encodings, all derived from ONE symmetric eigendecomposition
of the graph Laplacian.

The reference computes each encoding separately on the host and caches the result as a pickle
(transformer/position_encoding.py:11-52): scipy ``expm(-beta L)`` per graph for the diffusion
kernel (:65-72), sparse matrix powers for the p-step kernel (:83-93) and a general ``np.linalg.eig``
for the Laplacian eigenvectors (:127-161).  Here a graph is decomposed once, ``L = U diag(lam) U^T``
(``numpy.linalg.eigh``, fp64), and every encoding is a function of (U, lam):

    diffusion   U exp(-beta lam) U^T          (:71)
    p-step      U (1 - beta lam)^p U^T        (:89-92)
    LapEncoding columns 1..dim of U           (:137-161: ascending, first eigenvector dropped,
                                               zero-padded columns when the graph is small)
    spectral    (U[:, :K], lam - 1)           the eigenbasis of L_hat = L_sym - I consumed by
                                               feta_spec_filter_fwd/bwd (filter_mode='spectral')

Class names, constructor arguments and ``apply_to`` follow the reference; graphs are
``GraphSample``s (x, edge_index, ...) instead of torch-geometric ``Data`` objects.  The cache is an
``.npz`` per split instead of a pickle.
"""
import os

import numpy as np


def laplacian_dense(edge_index, n, normalization=None):
    """Dense Laplacian with torch-geometric ``get_laplacian`` semantics (self loops removed,
    degree on the source row, duplicate edges summed): None: D - A; 'sym': I - D^-1/2 A D^-1/2;
    'rw': I - D^-1 A."""
    s, t = np.asarray(edge_index)
    keep = s != t
    s, t = s[keep], t[keep]
    a = np.zeros((n, n))
    np.add.at(a, (s, t), 1.0)
    deg = a.sum(1)
    if normalization is None:
        return np.diag(deg) - a
    if normalization == 'sym':
        dis = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1e-300)), 0.0)
        return np.eye(n) - dis[:, None] * a * dis[None, :]
    if normalization == 'rw':
        dinv = np.where(deg > 0, 1.0 / np.maximum(deg, 1e-300), 0.0)
        return np.eye(n) - dinv[:, None] * a
    raise ValueError('Invalid normalization')


def decompose(edge_index, n, normalization=None):
    """-> (lam ascending [n], U [n, n]) with L = U diag(lam) U^T.  'rw' is decomposed through its
    similar symmetric matrix (L_rw = D^-1/2 L_sym D^1/2), so U is then not orthogonal:
    returns (lam, V, V_inv) in that case."""
    if normalization == 'rw':
        s, t = np.asarray(edge_index)
        keep = s != t
        deg = np.bincount(s[keep], minlength=n).astype(np.float64)
        lam, u = np.linalg.eigh(laplacian_dense(edge_index, n, 'sym'))
        d = np.sqrt(np.maximum(deg, 1e-300))
        dis = np.where(deg > 0, 1.0 / d, 1.0)
        dsq = np.where(deg > 0, d, 1.0)
        return lam, dis[:, None] * u, u.T * dsq[None, :]
    lam, u = np.linalg.eigh(laplacian_dense(edge_index, n, normalization))
    return lam, u, u.T


class PositionEncoding(object):
    """reference: transformer/position_encoding.py:11-52 (apply_to / save / load / compute_pe)."""

    attr = 'pe'

    def __init__(self, savepath=None, zero_diag=False):
        self.savepath = savepath
        self.zero_diag = zero_diag

    def apply_to(self, dataset, split='train'):
        saved = self.load(split)
        computed = []
        out = []
        for i, g in enumerate(dataset):
            pe = self.compute_pe(g) if saved is None else saved[i]
            if saved is None:
                computed.append(pe)
            if self.zero_diag:
                pe = pe.copy()
                np.fill_diagonal(pe, 0.0)
            setattr(g, self.attr, pe)
            out.append(pe)
        setattr(dataset, self.attr + '_list', out)
        if saved is None:
            self.save(computed, split)
        return dataset

    def _file(self, split):
        return None if self.savepath is None else self.savepath + '.' + split + '.npz'

    def save(self, pos_enc, split):
        f = self._file(split)
        if f is not None and not os.path.isfile(f):
            np.savez_compressed(f, **{'g%d' % i: p for i, p in enumerate(pos_enc)})

    def load(self, split):
        f = self._file(split)
        if f is None or not os.path.isfile(f):
            return None
        z = np.load(f)
        return [z['g%d' % i] for i in range(len(z.files))]

    def compute_pe(self, graph):
        raise NotImplementedError


class DiffusionEncoding(PositionEncoding):
    """expm(-beta L) (reference :55-72)."""

    def __init__(self, savepath=None, beta=1., use_edge_attr=False, normalization=None, zero_diag=False):
        super().__init__(savepath, zero_diag)
        if use_edge_attr:
            raise NotImplementedError('edge attributes are not used on the FeTA path')
        self.beta = beta
        self.normalization = normalization

    def compute_pe(self, graph):
        lam, v, vinv = decompose(graph.edge_index, graph.num_nodes, self.normalization)
        return ((v * np.exp(-self.beta * lam)) @ vinv).astype(np.float32)


class PStepRWEncoding(PositionEncoding):
    """(I - beta L)^p (reference :75-93)."""

    def __init__(self, savepath=None, p=1, beta=0.5, use_edge_attr=False, normalization=None, zero_diag=False):
        super().__init__(savepath, zero_diag)
        if use_edge_attr:
            raise NotImplementedError('edge attributes are not used on the FeTA path')
        self.p = p
        self.beta = beta
        self.normalization = normalization

    def compute_pe(self, graph):
        lam, v, vinv = decompose(graph.edge_index, graph.num_nodes, self.normalization)
        return ((v * (1.0 - self.beta * lam) ** max(self.p, 1)) @ vinv).astype(np.float32)   # :89-92: p - 1 products


class AdjEncoding(PositionEncoding):
    """Dense adjacency (reference :96-105)."""

    def __init__(self, savepath=None, normalization=None, zero_diag=False):
        super().__init__(savepath, zero_diag)
        self.normalization = normalization

    def compute_pe(self, graph):
        n = graph.num_nodes
        a = np.zeros((n, n), np.float32)
        s, t = np.asarray(graph.edge_index)
        a[s, t] = 1.0
        return a


class FullEncoding(PositionEncoding):
    """All ones (reference :107-115)."""

    def compute_pe(self, graph):
        return np.ones((graph.num_nodes, graph.num_nodes), np.float32)


class LapEncoding(PositionEncoding):
    """Laplacian eigenvector node features (reference :118-169): eigenvectors in ascending order of
    eigenvalue, the first one dropped, ``dim`` columns, zero-padded when the graph has fewer.
    The sign of every column is arbitrary (the reference's np.linalg.eig has the same freedom and
    its training loop flips signs at random, experiments/run_transformer_gengcn.py:126-131)."""

    attr = 'lap_pe'

    def __init__(self, dim, use_edge_attr=False, normalization=None):
        super().__init__(None, False)
        if use_edge_attr:
            raise NotImplementedError('edge attributes are not used on the FeTA path')
        self.pos_enc_dim = dim
        self.normalization = normalization

    def compute_pe(self, graph):
        n = graph.num_nodes
        _, v, _ = decompose(graph.edge_index, n, self.normalization)
        cols = v[:, 1:self.pos_enc_dim + 1]
        out = np.zeros((n, self.pos_enc_dim), np.float32)
        out[:, :cols.shape[1]] = cols
        return out

    def apply_to(self, dataset, split=None):
        return super().apply_to(dataset, split or 'train')


class SpectralEncoding(PositionEncoding):
    """Eigenbasis of the scaled Laplacian L_hat = L_sym - I that ChebConvDynamic filters on
    (transformer/ChebNetDynamic.py:108-130): sets graph.u [n, n] and graph.lam [n] (ascending),
    which ``data.collate(..., k_eig=K)`` truncates / pads into the [B,N,K] / [B,K] kernel inputs."""

    attr = 'spectral'

    def __init__(self):
        super().__init__(None, False)

    def compute_pe(self, graph):
        lam, u, _ = decompose(graph.edge_index, graph.num_nodes, 'sym')
        graph.u, graph.lam = u, lam - 1.0
        return lam - 1.0


POSENCODINGS = {
    'diffusion': DiffusionEncoding,
    'pstep': PStepRWEncoding,
    'adj': AdjEncoding,
}


# ---- the same encodings for a whole padded batch, on the device ---------------------------------

DEVICE_EIGH_MAX_N = 256   # feta_eigh_sym_supported (N <= 192: matrix in LDS; up to 256: in an L2-resident workspace)


def device_spectrum(edge_index, batch, node_off, n_real, n_pad, k_eig=None):
    """Batch producer of the spectral inputs: edge list of the collated batch ->
    (lhat [B,N,N], u [B,N,K], lam [B,K]) with K = k_eig or N, all on the device of the inputs.
    Two launches: feta_lhat_from_edges (ChebConvDynamic.__norm__ semantics) and feta_eigh_sym, one
    workgroup per graph; the reference decomposes graph by graph on the host (:127-161).  The
    spectrum is that of Lhat = L_sym - I, which is what ``filter_mode='spectral'`` consumes."""
    from .. import functional as FF
    lhat = FF.lhat_from_edges(edge_index, batch, node_off, n_real.shape[0], n_pad)
    if n_pad <= DEVICE_EIGH_MAX_N:
        u, lam = FF.eigh_sym(lhat, n_real, shift=2.0, k=k_eig)
        return lhat, u, lam
    # beyond the kernels' FETA_MAX_NODES = 256 (no BASELINE shape: the largest ogbg-molhiv graph has 222 nodes):
    # graph by graph on the host, as the reference does for every graph (:137)
    import torch
    k = n_pad if k_eig is None else int(k_eig)
    lh, ns = lhat.cpu().double().numpy(), n_real.cpu().tolist()
    u = np.zeros((len(ns), n_pad, k), np.float32)
    lam = np.zeros((len(ns), k), np.float32)
    for b, nb in enumerate(ns):
        w, v = np.linalg.eigh(lh[b, :nb, :nb])
        kk = min(k, nb)
        piv = np.abs(v[:, :kk]).argmax(0)
        sign = np.where(v[piv, np.arange(kk)] < 0, -1.0, 1.0)     # the sign rule of feta_eigh_sym
        u[b, :nb, :kk] = v[:, :kk] * sign
        lam[b, :kk] = w[:kk]
    return lhat, torch.from_numpy(u).to(lhat.device), torch.from_numpy(lam).to(lhat.device)


def device_kernel_pe(u, lam, n_real, kind='diffusion', beta=1.0, p=1, zero_diag=False):
    """pe [B,N,N] of the batch from the FULL spectrum (K = N) of Lhat: the 'sym'-normalised
    DiffusionEncoding / PStepRWEncoding (:55-93) as U f(lam + 1) U^T - one launch for the batch."""
    from .. import functional as FF
    if u.shape[2] != u.shape[1]:
        raise ValueError('a kernel of the whole Laplacian needs the full spectrum: K = %d, N = %d'
                         % (u.shape[2], u.shape[1]))
    return FF.spectral_kernel(u, lam, n_real, kind, beta=beta, p=p, lam_offset=1.0, zero_diag=zero_diag)


def device_lap_encoding(u, n_real, dim):
    """lap_pe [B,N,dim]: eigenvector columns 1..dim (ascending, the first dropped, zero columns when
    the graph has fewer: LapEncoding.compute_pe, :137-161) from the device decomposition."""
    import torch
    b, n, k = u.shape
    out = torch.zeros((b, n, dim), dtype=u.dtype, device=u.device)
    take = min(dim, k - 1)
    if take > 0:
        out[:, :, :take] = u[:, :, 1:1 + take]
    return out
