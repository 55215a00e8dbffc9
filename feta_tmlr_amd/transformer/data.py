"""Padded-graph batches for the FeTA block: synthetic generators of the BASELINE shapes
and a collate that emits the reference's wire format.

The reference collate (``GraphDataset_v2.collate_fn``, transformer/data.py:161-225) returns
the 9-tuple ``padded_x, mask, pos_enc, lap_pos_enc, degree, labels, edge_index, batch,
feature_indices`` built from Python lists; ``collate`` below returns the same tuple (same
shapes, dtypes and zero padding) from numpy arrays, plus a ``GraphBatchCache`` with what the
MI355X kernels consume directly: ``n_real`` (int32 node counts, replaces the host sync at
transformer/models.py:246), graph offsets, and optionally the Laplacian eigenbasis
``U, lam`` (the LapEncoding-style offline product, transformer/position_encoding.py:127-161).

Synthetic graphs follow SURVEY 8(d): molecule-like random trees with ring closures
(MUTAG/ZINC/molhiv shapes) and 5-block SBM graphs (PATTERN shape).  No dataset is read.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

SHAPES = {
    # name: (n_min, n_max, batch_size, kind)
    'mutag': (10, 28, 32, 'mol'),
    'zinc': (9, 37, 128, 'mol'),
    'pattern': (44, 188, 64, 'sbm'),
    'molhiv': (2, 222, 1024, 'mol'),
}


def molecule_graph(rng, n):
    """Random tree (parent among the last 3 placed nodes) + 2 ring closures, symmetrised,
    no self loops.  Returns edge_index [2, E] int64 with both directions of every edge."""
    edges = set()
    for v in range(1, n):
        p = int(rng.integers(max(0, v - 3), v))
        edges.add((p, v))
    for _ in range(2 if n >= 5 else 0):
        a, b = (int(t) for t in rng.integers(0, n, size=2))
        if a != b and abs(a - b) > 1:
            edges.add((min(a, b), max(a, b)))
    e = np.array(sorted(edges), dtype=np.int64).reshape(-1, 2)
    return np.concatenate([e, e[:, ::-1]], axis=0).T.copy()


def sbm_graph(rng, n, blocks=5, p_in=0.5, p_out=0.35, return_blocks=False):
    """Stochastic block model, symmetrised (PATTERN shape)."""
    lab = rng.integers(0, blocks, size=n)
    same = lab[:, None] == lab[None, :]
    prob = np.where(same, p_in, p_out)
    upper = np.triu(rng.random((n, n)) < prob, k=1)
    src, dst = np.nonzero(upper)
    e = np.stack([src, dst], axis=1).astype(np.int64)
    ei = np.concatenate([e, e[:, ::-1]], axis=0).T.copy()
    return (ei, lab.astype(np.int64)) if return_blocks else ei


@dataclass
class GraphSample:
    """One graph, the fields the reference reads off a PyG ``Data`` (g.x, g.edge_index, g.y,
    g.pe, g.lap_pe, g.degree: transformer/data.py:130-140)."""
    x: np.ndarray                      # [n, f] float32 node features (integer-valued for 'atom')
    edge_index: np.ndarray             # [2, E] int64
    y: object = 0.0                    # graph label (float / int / nan) or [n] int64 node labels
    pe: Optional[np.ndarray] = None    # [n, n] relative positional kernel
    lap_pe: Optional[np.ndarray] = None
    degree: Optional[np.ndarray] = None
    u: Optional[np.ndarray] = None     # [n, n] eigenvectors of Lhat (ascending), float64
    lam: Optional[np.ndarray] = None   # [n]

    @property
    def num_nodes(self):
        return self.x.shape[0]


def lhat_numpy(edge_index, n):
    """Dense Lhat = -D^-1/2 A D^-1/2 with the edge-list semantics of
    ChebConvDynamic.__norm__ (transformer/ChebNetDynamic.py:108-130): Lhat[t, s] over edges s->t,
    degree on the source row, duplicates summed."""
    s, t = edge_index
    keep = s != t
    s, t = s[keep], t[keep]
    deg = np.bincount(s, minlength=n).astype(np.float64)
    dis = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1e-300)), 0.0)
    m = np.zeros((n, n))
    np.add.at(m, (t, s), -dis[s] * dis[t])
    return m


def diffusion_kernel(edge_index, n, beta=1.0):
    """expm(-beta L_sym) (DiffusionEncoding, transformer/position_encoding.py:65-72) through the
    eigendecomposition of the symmetric L = I + Lhat."""
    lam, u = np.linalg.eigh(np.eye(n) + lhat_numpy(edge_index, n))
    return (u * np.exp(-beta * lam)) @ u.T


class SyntheticGraphDataset:
    """Seeded list of GraphSample of one BASELINE shape."""

    def __init__(self, shape='zinc', num_graphs=128, in_dim=64, seed=0, pos_enc=True,
                 with_degree=True, with_eig=True, n_min=None, n_max=None, features='normal',
                 labels='regression', nb_class=2, nan_label_frac=0.0):
        """features: 'normal' N(0,1) [n, in_dim] | 'atom' integer columns within ATOM_FEATURE_DIMS
        (ogbg-molhiv wire format).  labels: 'regression' float | 'class' int in [0, nb_class) |
        'binary' {0., 1.} with a fraction of NaN (the unlabeled molhiv graphs,
        experiments/run_transformer_gengcn_molhiv.py:177) | 'node' per-node block id (SBM)."""
        lo, hi, _, kind = SHAPES[shape]
        lo = lo if n_min is None else n_min
        hi = hi if n_max is None else n_max
        rng = np.random.default_rng(seed)
        self.samples: List[GraphSample] = []
        for _ in range(num_graphs):
            if shape == 'molhiv':
                n = int(np.clip(np.round(rng.lognormal(3.2, 0.35)), lo, hi))
            else:
                n = int(rng.integers(lo, hi + 1))
            blocks = None
            if kind == 'sbm':
                ei, blocks = sbm_graph(rng, n, return_blocks=True)
            else:
                ei = molecule_graph(rng, n)
            if features == 'atom':
                from .models import ATOM_FEATURE_DIMS
                xf = np.stack([rng.integers(0, dmax, size=n) for dmax in ATOM_FEATURE_DIMS], 1).astype(np.float32)
            else:
                xf = rng.standard_normal((n, in_dim)).astype(np.float32)
            yv = float(rng.standard_normal())
            if labels == 'class':
                y = int(rng.integers(0, nb_class))
            elif labels == 'binary':
                y = float(rng.integers(0, 2))
                if rng.random() < nan_label_frac:
                    y = float('nan')
            elif labels == 'node':
                y = blocks if blocks is not None else rng.integers(0, nb_class, size=n).astype(np.int64)
            else:
                y = yv
            g = GraphSample(x=xf, edge_index=ei, y=y)
            if pos_enc:
                g.pe = diffusion_kernel(ei, n).astype(np.float32)
            if with_degree:
                deg = np.bincount(ei[0], minlength=n).astype(np.float32)
                g.degree = 1.0 / np.sqrt(1.0 + deg)          # transformer/data.py:145
            if with_eig:
                g.lam, g.u = np.linalg.eigh(lhat_numpy(ei, n))
            self.samples.append(g)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        return self.samples[i]


@dataclass
class GraphBatchCache:
    """Per-batch graph structure in the layout the kernels read (device tensors)."""
    n_real: torch.Tensor                    # [B] int32
    node_off: torch.Tensor                  # [B] int32, first global node id of each graph
    n_pad: int
    u: Optional[torch.Tensor] = None        # [B, N, K] float32, rows >= n_real and cols >= n_real zero
    lam: Optional[torch.Tensor] = None      # [B, K]
    lhat: Optional[torch.Tensor] = None     # [B, N, N] dense scaled Laplacian (filled lazily)
    extra: dict = field(default_factory=dict)

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return GraphBatchCache(mv(self.n_real), mv(self.node_off), self.n_pad, mv(self.u),
                               mv(self.lam), mv(self.lhat),
                               {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.extra.items()})


def collate(samples, k_eig=None, n_pad=None, device='cpu', seq_first_degree=True):
    """-> (padded_x, mask, pos_enc, lap_pos_enc, degree, labels, edge_index, batch,
    feature_indices), cache   — tuple layout of transformer/data.py:224."""
    bsz = len(samples)
    ns = [g.num_nodes for g in samples]
    n = max(ns) if n_pad is None else n_pad
    f = samples[0].x.shape[1]
    x = np.zeros((bsz, n, f), np.float32)
    mask = np.ones((bsz, n), bool)
    use_pe = samples[0].pe is not None
    use_lap = samples[0].lap_pe is not None
    use_deg = samples[0].degree is not None
    pe = np.zeros((bsz, n, n), np.float32) if use_pe else None
    lap = np.zeros((bsz, n, samples[0].lap_pe.shape[1]), np.float32) if use_lap else None
    deg = np.zeros((bsz, n), np.float32) if use_deg else None
    eis, bat, fi = [], [], []
    off = 0
    offs = []
    for i, g in enumerate(samples):
        m = ns[i]
        x[i, :m] = g.x
        mask[i, :m] = False
        if use_pe:
            pe[i, :m, :m] = g.pe
        if use_lap:
            lap[i, :m, :g.lap_pe.shape[1]] = g.lap_pe
        if use_deg:
            deg[i, :m] = g.degree
        eis.append(g.edge_index + off)
        bat.append(np.full((m,), i, np.int64))
        fi.append(np.stack([np.full((m,), i, np.int64), np.arange(m, dtype=np.int64)], 1))
        offs.append(off)
        off += m
    u = lam = None
    if k_eig is not None:
        u = np.zeros((bsz, n, k_eig), np.float32)
        lam = np.zeros((bsz, k_eig), np.float32)
        for i, g in enumerate(samples):
            kk = min(k_eig, ns[i])
            u[i, :ns[i], :kk] = g.u[:, :kk]
            lam[i, :kk] = g.lam[:kk]
    t = lambda a: None if a is None else torch.from_numpy(a).to(device)
    y0 = samples[0].y
    if isinstance(y0, np.ndarray) and y0.ndim >= 1 and y0.shape[0] == ns[0]:
        labels = t(np.concatenate([np.asarray(g.y) for g in samples]))       # node labels: transformer/data.py:456
    elif isinstance(y0, (int, np.integer)):
        labels = torch.tensor([int(g.y) for g in samples], dtype=torch.int64, device=device)
    else:
        labels = torch.tensor([g.y for g in samples], dtype=torch.float32, device=device)
    batch9 = (t(x), t(mask), t(pe), t(lap), t(deg), labels,
              t(np.concatenate(eis, axis=1)), t(np.concatenate(bat)), t(np.concatenate(fi)))
    cache = GraphBatchCache(n_real=t(np.array(ns, np.int32)), node_off=t(np.array(offs, np.int32)),
                            n_pad=n, u=t(u), lam=t(lam))
    if use_deg and seq_first_degree:
        # the degree scale per row of the seq-first [N*B, d] activation view (row = node * B + graph),
        # so that no transpose kernel runs per step
        cache.extra['degree_rows'] = t(np.ascontiguousarray(deg.T).reshape(-1))
    return batch9, cache


def attach_device_spectrum(batch9, cache, k_eig=None, pos_enc=None, beta=1.0, p=1, zero_diag=False,
                           lap_dim=None):
    """Fill the spectral inputs of a collated batch ON THE DEVICE instead of per graph on the host
    (SURVEY 8f N2-N4): ``cache.lhat``, ``cache.u`` [B,N,K], ``cache.lam`` [B,K] from the batch's edge
    list (position_encoding.device_spectrum), and optionally the relative kernel ``pos_enc``
    ('diffusion' | 'pstep', 'sym' normalisation) and ``lap_dim`` Laplacian eigenvector features, which
    replace entries 2 / 3 of the 9-tuple.  The batch must already live on the GPU
    (``collate(..., device=...)``; graphs need no ``u`` / ``lam`` / ``pe`` of their own)."""
    from . import position_encoding as PE
    x, mask, pe, lap, deg, labels, edge_index, batch, fi = batch9
    n = cache.n_pad
    full = pos_enc is not None or lap_dim is not None
    lhat, u, lam = PE.device_spectrum(edge_index, batch, cache.node_off, cache.n_real, n,
                                      None if full else k_eig)
    if pos_enc is not None:
        pe = PE.device_kernel_pe(u, lam, cache.n_real, pos_enc, beta=beta, p=p, zero_diag=zero_diag)
    if lap_dim is not None:
        lap = PE.device_lap_encoding(u, cache.n_real, lap_dim)
    cache.lhat = lhat
    k = n if k_eig is None else min(k_eig, n)
    cache.u = u if k == u.shape[2] else u[:, :, :k].contiguous()
    cache.lam = lam if k == lam.shape[1] else lam[:, :k].contiguous()
    return (x, mask, pe, lap, deg, labels, edge_index, batch, fi), cache


BUCKETS = (16, 32, 48, 64, 128, 256)


def bucket_batches(samples, batch_size, buckets=BUCKETS, shuffle_rng=None):
    """Variable-N batching for wide size distributions (ogbg-molhiv: 2..222 nodes, BASELINE config 5):
    graphs are grouped by the smallest padded size in ``buckets`` that holds them, and every batch
    is cut from one group, so the padded area (and the N^2 attention work) follows the graphs.
    -> list of (n_pad, [sample indices]); the reference pads every batch to its own maximum
    (transformer/data.py:165), which is the special case of one bucket per batch."""
    groups = {}
    for i, g in enumerate(samples):
        n = g.num_nodes
        npad = next((bk for bk in buckets if n <= bk), None)
        if npad is None:
            raise ValueError('graph %d has %d nodes, more than the largest bucket %d' % (i, n, buckets[-1]))
        groups.setdefault(npad, []).append(i)
    out = []
    for npad in sorted(groups):
        idx = groups[npad]
        if shuffle_rng is not None:
            idx = list(shuffle_rng.permutation(idx))
        for k in range(0, len(idx), batch_size):
            out.append((npad, idx[k:k + batch_size]))
    return out


# ---- N3: collate without a per-graph Python loop, staged through pinned host buffers -----------------------------

class PackedGraphs:
    """A whole split as flat arrays (one-time conversion of the GraphSample list): node features / degree of all
    graphs back to back with per-graph offsets, edge lists likewise.  A batch is then a handful of vectorised gathers
    instead of the reference's per-graph loop over Python lists (transformer/data.py:197-219)."""

    def __init__(self, samples):
        ns = np.array([g.num_nodes for g in samples], np.int64)
        es = np.array([g.edge_index.shape[1] for g in samples], np.int64)
        self.num_graphs = len(samples)
        self.n = ns
        self.node_off = np.concatenate([[0], np.cumsum(ns)])
        self.edge_off = np.concatenate([[0], np.cumsum(es)])
        self.x = np.concatenate([g.x for g in samples], axis=0).astype(np.float32)
        self.edges = np.concatenate([g.edge_index for g in samples], axis=1).astype(np.int64)   # graph-local node ids
        self.degree = (np.concatenate([g.degree for g in samples]).astype(np.float32)
                       if samples[0].degree is not None else None)
        y0 = samples[0].y
        self.node_labels = isinstance(y0, np.ndarray) and y0.ndim >= 1 and y0.shape[0] == ns[0]
        if self.node_labels:
            self.y = np.concatenate([np.asarray(g.y) for g in samples]).astype(np.int64)
        elif isinstance(y0, (int, np.integer)):
            self.y = np.array([int(g.y) for g in samples], np.int64)
        else:
            self.y = np.array([g.y for g in samples], np.float32)


class BatchStager:
    """Builds the reference's 9-tuple (+ GraphBatchCache) for a list of graph ids of a PackedGraphs:
      * vectorised fill of PREALLOCATED PINNED host buffers (x, mask, degree and its seq-first row layout, labels,
        edge_index, batch, feature_indices, n_real, node_off): no per-graph Python loop, no allocation per batch;
      * one asynchronous host-to-device copy per field on a copy stream, into preallocated device buffers (two
        sets, used alternately, so that staging batch i + 1 overlaps the step on batch i);
      * the dense per-graph matrices are NOT built on the host: pe ('diffusion' | 'pstep'), U / lambda (k_eig) and
        the Laplacian eigenvector features (lap_dim) come from the device (attach_device_spectrum: edge list ->
        Lhat -> feta_eigh_sym -> feta_spectral_kernel), which replaces the reference's offline pickle cache
        (transformer/position_encoding.py:35-49) and its [B,N,N] host tensors.
    Reference: GraphDataset_v2.collate_fn, transformer/data.py:161-225 (same tuple layout, dtypes, zero padding)."""

    def __init__(self, packed, max_batch, n_pad, device, pos_enc=None, k_eig=None, lap_dim=None, beta=1.0, p=1,
                 zero_diag=False):
        self.pk, self.bmax, self.n_pad, self.device = packed, int(max_batch), int(n_pad), torch.device(device)
        self.pos_enc, self.k_eig, self.lap_dim, self.beta, self.p, self.zero_diag = pos_enc, k_eig, lap_dim, beta, p, zero_diag
        self.cuda = self.device.type == 'cuda'
        f = packed.x.shape[1]
        nmax_nodes = self.bmax * self.n_pad
        emax = int(np.max(packed.edge_off[1:] - packed.edge_off[:-1])) * self.bmax if packed.num_graphs else 0
        pin = self.cuda

        def host(shape, dtype):
            return torch.zeros(shape, dtype=dtype, pin_memory=pin)

        self.sets = []
        for _ in range(2):
            h = dict(x=host((self.bmax, self.n_pad, f), torch.float32), mask=host((self.bmax, self.n_pad), torch.bool),
                     degree=host((self.bmax, self.n_pad), torch.float32),
                     degree_rows=host((self.n_pad * self.bmax,), torch.float32),
                     labels=host((nmax_nodes if packed.node_labels else self.bmax,),
                                 torch.int64 if packed.y.dtype == np.int64 else torch.float32),
                     edge_index=host((2, emax), torch.int64), batch=host((nmax_nodes,), torch.int64),
                     fi=host((nmax_nodes, 2), torch.int64), n_real=host((self.bmax,), torch.int32),
                     node_off=host((self.bmax,), torch.int32))
            d = {k: (torch.empty_like(v, device=self.device) if self.cuda else v) for k, v in h.items()}
            self.sets.append((h, d))
        self.turn = 0
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        # one event per buffer set, recorded behind that set's H2D copies: the pinned host buffers of a set are the
        # SOURCE of asynchronous copies, so a host that runs two or more batches ahead must not refill them before
        # the copy engine has read them (stream-side ordering alone does not stop the host)
        self.copied = [None, None]

    def stage(self, graph_ids):
        """-> (batch9, cache) on the device; the copies are enqueued on the copy stream and the current stream is
        made to wait for them (no host synchronisation)."""
        pk, n_pad = self.pk, self.n_pad
        ids = np.asarray(graph_ids, np.int64)
        bsz = len(ids)
        assert 0 < bsz <= self.bmax
        turn = self.turn
        h, d = self.sets[turn]
        self.turn ^= 1
        if self.copied[turn] is not None:
            self.copied[turn].synchronize()     # the copies enqueued from this set two calls ago have been read
        ns = pk.n[ids]
        assert int(ns.max()) <= n_pad, 'graph with %d nodes in a batch padded to %d' % (int(ns.max()), n_pad)
        n_tot = int(ns.sum())
        offs = np.concatenate([[0], np.cumsum(ns)])[:-1]
        # node (b, i) of the batch <- global node id of the split: one gather for every per-node field
        b_of = np.repeat(np.arange(bsz), ns)
        i_of = np.arange(n_tot) - np.repeat(offs, ns)
        src = np.repeat(pk.node_off[ids], ns) + i_of
        x = h['x'].numpy()
        x[:bsz].fill(0.0)
        x[b_of, i_of] = pk.x[src]
        mask = h['mask'].numpy()
        mask[:bsz] = np.arange(n_pad)[None, :] >= ns[:, None]
        deg = None
        if pk.degree is not None:
            deg = h['degree'].numpy()
            deg[:bsz].fill(0.0)
            deg[b_of, i_of] = pk.degree[src]
            rows = h['degree_rows'].numpy()[:n_pad * bsz].reshape(n_pad, bsz)
            rows[:] = deg[:bsz].T                                  # row = node * B + graph (seq-first activations)
        # edges: graph-local ids + the graph's offset in the batch
        es = pk.edge_off[ids + 1] - pk.edge_off[ids]
        e_tot = int(es.sum())
        e_src = np.repeat(pk.edge_off[ids], es) + (np.arange(e_tot) - np.repeat(np.concatenate([[0], np.cumsum(es)])[:-1], es))
        ei = h['edge_index'].numpy()
        ei[:, :e_tot] = pk.edges[:, e_src] + np.repeat(offs, es)[None, :]
        h['batch'].numpy()[:n_tot] = b_of
        fi = h['fi'].numpy()
        fi[:n_tot, 0] = b_of
        fi[:n_tot, 1] = i_of
        h['n_real'].numpy()[:bsz] = ns
        h['node_off'].numpy()[:bsz] = offs
        lab = h['labels'].numpy()
        if pk.node_labels:
            lab[:n_tot] = pk.y[src]
            n_lab = n_tot
        else:
            lab[:bsz] = pk.y[ids]
            n_lab = bsz
        views = dict(x=(slice(0, bsz),), mask=(slice(0, bsz),), degree=(slice(0, bsz),),
                     degree_rows=(slice(0, n_pad * bsz),), labels=(slice(0, n_lab),),
                     edge_index=(slice(None), slice(0, e_tot)), batch=(slice(0, n_tot),), fi=(slice(0, n_tot),),
                     n_real=(slice(0, bsz),), node_off=(slice(0, bsz),))
        out = {}
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            self.copy_stream.wait_stream(cur)       # the device buffers of this set are free again
            with torch.cuda.stream(self.copy_stream):
                for k, sl in views.items():
                    if k in ('degree', 'degree_rows') and deg is None:
                        continue
                    out[k] = d[k][sl]
                    out[k].copy_(h[k][sl], non_blocking=True)
                if self.copied[turn] is None:
                    self.copied[turn] = torch.cuda.Event()
                self.copied[turn].record(self.copy_stream)
            cur.wait_stream(self.copy_stream)
        else:
            for k, sl in views.items():
                if k in ('degree', 'degree_rows') and deg is None:
                    continue
                out[k] = h[k][sl].clone()
        batch9 = (out['x'], out['mask'], None, None, out.get('degree'), out['labels'], out['edge_index'], out['batch'],
                  out['fi'])
        cache = GraphBatchCache(n_real=out['n_real'], node_off=out['node_off'], n_pad=n_pad)
        if 'degree_rows' in out:
            cache.extra['degree_rows'] = out['degree_rows']
        if self.pos_enc is not None or self.k_eig is not None or self.lap_dim is not None:
            batch9, cache = attach_device_spectrum(batch9, cache, k_eig=self.k_eig, pos_enc=self.pos_enc,
                                                   beta=self.beta, p=self.p, zero_diag=self.zero_diag,
                                                   lap_dim=self.lap_dim)
        return batch9, cache
