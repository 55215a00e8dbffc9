"""FeTA encoder and model shell with the reference's class names, constructor arguments,
forward signatures and state-dict keys (transformer/models.py:103-368, 487-595), running the
attention core, the coefficient generator and the spectral filter in the MI355X kernels.

What differs from the reference by construction (DESIGN.md lists the parity stance of each):
  * no host sync and no Python loop per (head, graph) block: node counts travel as a device
    int32 array, the dense attention graph of get_filter_coefficients is never materialised
    (reference: transformer/models.py:246,252-264,280-282);
  * per-node weight copies, head stacking, gather and scatter are folded into the filter
    kernel (reference: transformer/ChebNetDynamic.py:148-149, transformer/models.py:178-186,200-202);
  * ``heads_share_graph=False`` (default) reproduces the reference's un-replicated
    ``edge_index`` for the stacked heads (transformer/models.py:186); ``True`` filters every head
    on the graph (the behaviour of the in-tree DGL variants);
  * ``filter_mode='cheb'`` is the reference operator (direct recursion); ``'spectral'`` is the
    eigenbasis form using U, lambda_hat from ``graph_cache`` (exact when K spans the graph).
"""
import math

import torch
from torch import nn
import torch.nn.functional as F

from .. import functional as FF
from ..fused_stack import StackTail, fused_encoder_stack, stack_supported
from ..parallel import mark_row_constant
from .ChebNetDynamic import ChebConvDynamic
from .data import GraphBatchCache
from .layers import DiffTransformerEncoderLayer, clone_layers, linear_rows, n_real_from_mask


class DenseGCNParams(nn.Module):
    """Parameters of the reference's ``self.gcn = GCNConv(C, C)`` (transformer/models.py:144):
    ``weight [in, out]`` glorot, ``bias [out]`` zeros (vendored text transformer/GenGCN.py:340-356).
    On the FeTA path its input is all ones, so only colsum(weight) and bias are ever used."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        stdv = math.sqrt(6.0 / (in_channels + out_channels))
        self.weight.data.uniform_(-stdv, stdv)


class DiffTransformerEncoder(nn.Module):
    """transformer/models.py:88-100."""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = clone_layers(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, pe, degree=None, mask=None, src_key_padding_mask=None, return_attn=False):
        output = src
        attn = None
        for mod in self.layers:
            output, attn = mod(output, pe=pe, degree=degree, src_mask=mask,
                               src_key_padding_mask=src_key_padding_mask)
        if self.norm is not None:
            output = self.norm(output)
        return (output, attn) if return_attn else output


class DiffTransformerEncoderGenGCN(nn.Module):
    """transformer/models.py:103-368 (gnn_type='ChebConvDynamic')."""

    def __init__(self, d_model, num_heads, encoder_layer, num_layers, norm=None, num_coefficients=4,
                 laplacian_norm='sym', gnn_type='ChebConvDynamic', last_layer_filter=True,
                 learn_only_filter_order_coeff=False, use_skip_conn=True,
                 heads_share_graph=False, filter_mode='cheb'):
        super().__init__()
        if gnn_type != 'ChebConvDynamic':
            raise NotImplementedError("only gnn_type='ChebConvDynamic' is on the FeTA hot path")
        assert filter_mode in ('cheb', 'spectral')
        self.layers = clone_layers(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        dh = d_model // num_heads
        self.order = num_coefficients                                           # :127,130
        self.filter_in_channels = dh
        self.filter_out_channels = dh
        if learn_only_filter_order_coeff:
            self.num_coefficients = num_coefficients                            # :126-128
        else:
            self.num_coefficients = self.order * dh * dh                        # :133
        self.spectral_gnns = ChebConvDynamic(dh, dh, self.order, normalization=laplacian_norm,
                                             learn_only_filter_order_coeff=learn_only_filter_order_coeff)
        self.gcn = DenseGCNParams(self.num_coefficients, self.num_coefficients)  # :144
        mark_row_constant(self.gcn.weight)   # all-ones input => every row of its gradient is equal
        self.linear = nn.Linear(self.num_coefficients, self.num_coefficients)    # :145
        self.linear_cat = nn.Linear(2 * d_model, d_model)                        # :146
        self.gnn_type = gnn_type
        self.num_heads = num_heads
        self.last_layer_filter = last_layer_filter
        self.learn_only_filter_order_coeff = learn_only_filter_order_coeff
        self.use_skip_conn = use_skip_conn
        self.heads_share_graph = heads_share_graph
        self.filter_mode = filter_mode
        self.storage_dtype = torch.float32   # torch.bfloat16: bf16 storage path (layers.set_storage_dtype)
        self.spectral_k = None    # eigenpairs kept when the eigenbasis is computed here (None: all N_pad)
        self.fused_stack = True   # layer stacks (BatchNorm or LayerNorm) run as one autograd node when the dims allow
        self.keep_stack_boundary = False   # set by trainers that use backward_head / backward_stack
        self._stack_boundary = None
        self._stack_grads = None

    # -- A2 ---------------------------------------------------------------------------------
    def get_filter_coefficients(self, attn_weights, edge_index=None, feature_indices=None,
                                batch=None, masks=None, n_real=None):
        """-> [H, B, C]  (transformer/models.py:240-287; edge_index / feature_indices / batch are
        ignored there as well, :248-249)."""
        if n_real is None:
            n_real = n_real_from_mask(masks)
        return self._coefficients(attn_weights, n_real)

    def _coefficients(self, attn_weights, n_real):
        pooled = FF.filter_coefficients(attn_weights.detach(), n_real, self.gcn.weight, self.gcn.bias)
        coeff = FF.dense_linear(pooled, self.linear.weight, self.linear.bias)    # :284
        return coeff.reshape(self.num_heads, attn_weights.shape[0], -1)          # :285

    def _coefficients_and_filter(self, attn_weights, out_each_head, cache, pending=None, cat=None):
        """get_filter_coefficients + filter of one layer (:173, :186-202) with ``self.linear`` folded into the
        filter's autograd node (functional.FilterFromPooledFn).  -> (coeff [H,B,C], out_filtered [N,B,d])"""
        bsz, n, h, dh = out_each_head.shape
        pooled = FF.filter_coefficients(attn_weights.detach().float(), cache.n_real, self.gcn.weight, self.gcn.bias,
                                        pending)
        if self.filter_mode == 'cheb':
            graph, mode = (cache.lhat,), 'cheb'
        else:
            if cache.u is None:
                raise ValueError("filter_mode='spectral' needs graph_cache.u / graph_cache.lam "
                                 '(collate(..., k_eig=K))')
            graph, mode = (cache.u, cache.lam), 'spec'
        y, coeff = FF.filter_from_pooled(out_each_head, pooled, self.linear.weight, self.linear.bias,
                                         self.spectral_gnns.bias, cache.n_real, graph, mode, self.order,
                                         self.heads_share_graph, pending=pending,
                                         gemm_bf16=self.storage_dtype != torch.float32, cat=cat)
        return coeff.reshape(h, bsz, -1), y.permute(1, 0, 2, 3).reshape(n, bsz, h * dh)

    # -- A3 ---------------------------------------------------------------------------------
    def filter(self, coeff_all_heads, out_each_head, cache):
        """coeff [H,B,C], out_each_head [B,N,H,dh] -> out_filtered [N,B,d] (zeros on padded rows);
        replaces transformer/models.py:178-186,200-202,346-360."""
        bsz, n, h, dh = out_each_head.shape
        w = coeff_all_heads.reshape(h * bsz, -1)
        if self.learn_only_filter_order_coeff:
            w = self.spectral_gnns.group_weights(w.reshape(h * bsz, self.order).permute(1, 0))
        if self.filter_mode == 'cheb':
            y = FF.cheb_filter(out_each_head, cache.lhat, w, self.spectral_gnns.bias, cache.n_real,
                               self.order, self.heads_share_graph)
        else:
            if cache.u is None:
                raise ValueError("filter_mode='spectral' needs graph_cache.u / graph_cache.lam "
                                 '(collate(..., k_eig=K))')
            y = FF.spec_filter(out_each_head, cache.u, cache.lam, w, self.spectral_gnns.bias,
                               cache.n_real, self.order, self.heads_share_graph)
        return y.permute(1, 0, 2, 3).reshape(n, bsz, h * dh)

    def _graph_cache(self, graph_cache, edge_index, batch, key_padding_mask, n_pad):
        if graph_cache is None:
            n_real = n_real_from_mask(key_padding_mask)
            off = (torch.cumsum(n_real, 0) - n_real).to(torch.int32)
            graph_cache = GraphBatchCache(n_real=n_real, node_off=off, n_pad=n_pad)
        if self.filter_mode == 'cheb' and graph_cache.lhat is None:
            graph_cache.lhat = FF.lhat_from_edges(edge_index, batch, graph_cache.node_off,
                                                  graph_cache.n_real.shape[0], n_pad)
        if self.filter_mode == 'spectral' and graph_cache.u is None:
            # no eigenbasis came with the batch: decompose Lhat of every graph on the device (edge list ->
            # Lhat -> feta_eigh_sym, two launches per batch; spectral_k = None keeps the full basis, which
            # makes the filter the exact Chebyshev operator)
            from . import position_encoding as PE
            graph_cache.lhat, graph_cache.u, graph_cache.lam = PE.device_spectrum(
                edge_index, batch, graph_cache.node_off, graph_cache.n_real, n_pad, self.spectral_k)
        return graph_cache

    # -- two-phase backward (data-parallel overlap) ---------------------------------------------
    def head_parameters(self):
        """Parameters of the filter stage (gcn, linear, linear_cat, spectral_gnns): ~96 % of the
        gradient bytes, and the FIRST gradients backward produces."""
        mods = (self.gcn, self.linear, self.linear_cat, self.spectral_gnns)
        return [p for m in mods for p in m.parameters() if p.requires_grad]

    def stack_parameters(self):
        return [p for p in self.layers.parameters() if p.requires_grad]

    def stack_flat_grad(self):
        """The flat buffer that holds every gradient of the layer stack after a backward through the
        fused BatchNorm stack (all stack_parameters().grad are views of it), or None."""
        from ..fused_stack import STACK_FLAT_GRAD
        return STACK_FLAT_GRAD.get(self.layers[0]) if len(self.layers) else None

    def backward_head(self, out, grad_out):
        """Phase 1 of backward after a fused-stack forward: gradients of head_parameters() (assigned
        to .grad) and of the stack outputs (kept for backward_stack).  A data-parallel trainer starts
        the all-reduce of the head bucket here and runs backward_stack() underneath it."""
        bnd = self._stack_boundary
        if bnd is None:
            raise RuntimeError('backward_head needs keep_stack_boundary = True and a forward through '
                               'the fused BatchNorm stack')
        head = self.head_parameters()
        grads = torch.autograd.grad(out, list(bnd) + head, grad_outputs=grad_out, allow_unused=True)
        for p, g in zip(head, grads[2:]):
            p.grad = g
        self._stack_grads = grads[:2]

    def backward_stack(self):
        """Phase 2: backward through the encoder-layer stack (accumulates stack_parameters().grad)."""
        bnd, grads = self._stack_boundary, self._stack_grads
        self._stack_boundary = self._stack_grads = None   # do not keep the autograd graph alive
        torch.autograd.backward(list(bnd), list(grads))

    def forward(self, src, pe, edge_index, feature_indices, batch, degree=None, mask=None,
                src_key_padding_mask=None, eigenvalues=None, graph_cache=None):
        """-> (output [N,B,d], attn [B,H,N,N] of the last layer, coefficients [B, H*n_filtered, C]).
        ``graph_cache`` (optional, this package): GraphBatchCache from ``data.collate`` holding
        n_real / Lhat / U, lambda so that nothing is derived from edge_index per call."""
        output = src
        n = src.shape[0]
        cache = self._graph_cache(graph_cache, edge_index, batch, src_key_padding_mask, n)
        coefficients = []
        allout_filtered = None
        attn = None
        degree_rows = None
        if degree is not None:   # degree [B,N] -> one value per row of the [N*B, d] view, once
            degree_rows = cache.extra.get('degree_rows')   # collate(..., seq_first_degree=True) emits it
            if degree_rows is None or degree_rows.shape[0] != degree.numel():
                degree_rows = degree.transpose(0, 1).reshape(-1).contiguous()
        lowp = self.storage_dtype != torch.float32
        if lowp:
            if self.filter_mode != 'spectral' or self.learn_only_filter_order_coeff:
                raise NotImplementedError("the bf16 storage path runs filter_mode='spectral' with matrix coefficients")
            output = output.to(self.storage_dtype)
            pe = None if pe is None else pe.to(self.storage_dtype)    # once for all layers
        fused = (self.fused_stack and self.last_layer_filter and mask is None
                 and stack_supported(self.layers, src.shape[-1]))
        if lowp and fused:
            # bf16 storage: the stack runs the bf16 instantiations of the four fused kernels where the shape has them
            # (fused_stack.lowp_stack_supported) and hands fp32 tensors to the filter stage, which is the fp32 path
            # below, unchanged; other shapes take the op-by-op bf16 path (general bf16 kernels + library bf16 GEMMs)
            from .. import _lib
            from ..fused_stack import lowp_stack_supported
            fused = lowp_stack_supported(_lib.backend(output)[0], self.layers, n, src.shape[1], src.shape[-1])
        lowp = lowp and not fused      # from here on: the op-by-op bf16 path
        # BatchNorm stack whose only consumer is linear_cat: the last BatchNorm is applied inside linear_cat's kernels
        tail = None
        if (fused and self.layers[0].batch_norm and self.use_skip_conn and self.norm is None
                and src.shape[-1] % 16 == 0 and FF.row_linear_supported(2 * src.shape[-1], self.linear_cat.weight.shape[0])):
            tail = StackTail()
        # linear_cat's weight-gradient partials are reduced inside the launch of the filter's backward (PendingSums)
        # (one filter stage per forward only: a parameter that collects several contributions is summed by autograd
        # as they arrive)
        pending = FF.PendingSums() if (not lowp and self.last_layer_filter) else None
        if (pending is not None and fused and not self.learn_only_filter_order_coeff and self.linear.bias is not None):
            # s = colsum(gcn.weight) of the coefficient generator: computed by trailing workgroups of the stack's
            # first launch instead of a launch of its own
            pending.s = torch.empty(self.gcn.weight.shape[1], dtype=torch.float32, device=src.device)
            pending.fwd_sums = [(self.gcn.weight.detach(), pending.s)]
            # ... and the generator's forward kernel in the launch of the last layer's feed-forward half
            pending.coeff_fwd_req = self.gcn.bias
        cat = None
        for layer_num, mod in enumerate(self.layers):
            last = layer_num + 1 == self.num_layers
            filt = last or not self.last_layer_filter                            # :169-171
            if fused:
                # every layer in one autograd node (feta_tmlr_amd/fused_stack.py); the loop body
                # below then only runs the filter stage of the last layer
                if not last:
                    continue
                output, concat, attn = fused_encoder_stack(output, pe, degree_rows, cache.n_real,
                                                           self.layers, need_attn=True, tail=tail, pending=pending)
                if self.keep_stack_boundary:   # for backward_head / backward_stack
                    self._stack_boundary = (output, concat)
                elif pending is not None and concat.requires_grad:
                    # one backward pass runs the filter stage and then the stack: the stack's reduction launch takes
                    # the stage's column sums (two passes: the head gradients must be final when the first returns)
                    pending.stack_armed = True
                nn_, bb_, dd_ = concat.shape
                out_each_head = concat.view(nn_, bb_, self.num_heads, dd_ // self.num_heads).permute(1, 0, 2, 3)
            else:
                output, attn, out_each_head = mod(output, pe=pe, degree=degree, src_mask=mask,
                                                  src_key_padding_mask=src_key_padding_mask,
                                                  need_heads=True, n_real=cache.n_real,
                                                  need_weights=filt, degree_rows=degree_rows)
            if not filt:
                continue
            if self.learn_only_filter_order_coeff or self.linear.bias is None:
                coeff_all_heads = self.get_filter_coefficients(attn, masks=src_key_padding_mask,
                                                               n_real=cache.n_real)   # :173
                out_filtered = self.filter(coeff_all_heads, out_each_head, cache)     # :186-202
            else:
                # linear_cat (:223-224) can ride in the filter's launch when this is the one filter stage of the forward
                # and its operands are the row-linear kernels' (functional.CatFold)
                cat = None
                if (last and self.last_layer_filter and self.use_skip_conn and not lowp and self.norm is None
                        and output.dtype == torch.float32 and self.linear_cat.weight.shape[0] == output.shape[-1]
                        and (tail is not None or not (fused and self.layers[0].batch_norm))):
                    cat = FF.CatFold(output, self.linear_cat.weight, self.linear_cat.bias, tail)
                coeff_all_heads, out_filtered = self._coefficients_and_filter(attn, out_each_head, cache, pending, cat)
            coefficients.append(coeff_all_heads)                                  # :198
            if self.use_skip_conn and allout_filtered is not None:
                allout_filtered = allout_filtered + out_filtered                  # :209-213
            else:
                allout_filtered = out_filtered
            if not self.use_skip_conn:
                output = allout_filtered                                          # :215-216
        if self.use_skip_conn and allout_filtered is not None:
            nn_, bb_, dd_ = output.shape
            wc = self.linear_cat.weight
            done = cat.out if (cat is not None and cat.out is not None) else None    # (computed by the filter's launch)
            if lowp:    # plain library bf16 GEMM on bf16 copies of the master weights; the encoder hands back fp32
                dt = self.storage_dtype
                output = F.linear(torch.cat((output, allout_filtered.to(dt)), dim=-1), wc.to(dt),
                                  self.linear_cat.bias.to(dt)).float()
            elif tail is not None:
                output = FF.row_linear_cat_bn(output.reshape(nn_ * bb_, dd_), allout_filtered.reshape(nn_ * bb_, dd_),
                                              wc, self.linear_cat.bias, tail, pending, done=done, fold=cat).view(nn_, bb_, -1)
            elif (dd_ % 16 == 0 and FF.row_linear_supported(2 * dd_, wc.shape[0])):
                # [output | allout_filtered] W^T + b without materialising the concatenation (:223-224)
                output = FF.row_linear_cat(output.reshape(nn_ * bb_, dd_), allout_filtered.reshape(nn_ * bb_, dd_),
                                           wc, self.linear_cat.bias, pending, done=done, fold=cat).view(nn_, bb_, -1)
            else:
                cat = torch.cat((output, allout_filtered), dim=-1)               # :223
                output, _ = linear_rows(cat.reshape(nn_ * bb_, 2 * dd_), wc, self.linear_cat.bias)   # :224
                output = output.view(nn_, bb_, -1)
        if output.dtype != torch.float32:
            output = output.float()
        if self.norm is not None:
            output = self.norm(output)
        if len(coefficients) == 1:      # last_layer_filter: a view, no copy of the [H, B, C] block
            coeffs = coefficients[0].permute(1, 0, 2)
        else:
            coeffs = torch.cat(coefficients, dim=0).permute(1, 0, 2) if coefficients else None
        return output, attn, coeffs                                               # :238


class GlobalAvg1D(nn.Module):
    """transformer/models.py:586-595."""

    def forward(self, x, mask=None):
        if mask is None:
            return x.mean(dim=1)
        m = (~mask).float().unsqueeze(-1)
        return (x * m).sum(dim=1) / m.sum(dim=1)


class _InputEmbeddingFn(torch.autograd.Function):
    """y = x W^T for the node-feature embedding ``nn.Linear(in_size, d_model, bias=False)`` of the task shells
    (transformer/models.py:521-522) when x is data (no gradient).  Plain library GEMMs; the point is the weight
    gradient: dW = dy^T x contracts over all N*B rows into a d_model x in_size result, which the BLAS heuristics
    run as ONE workgroup (38 us at the ZINC batch) - here it is split over the leading (node) dimension as a
    batched GEMM plus one small sum (10 us)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        return torch.matmul(x, w.t())

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dw = torch.bmm(dy.transpose(1, 2), x).sum(0)     # [S, d, B] x [S, B, f] -> [S, d, f] -> [d, f]
        return None, dw


def input_embedding(linear, x):
    """x [S, B, in_size] -> [S, B, d_model] through ``linear`` (an nn.Linear)."""
    if linear.bias is None and x.dim() == 3 and not x.requires_grad and torch.is_grad_enabled() and linear.weight.requires_grad:
        return _InputEmbeddingFn.apply(x, linear.weight)
    return linear(x)


class DiffGraphTransformerGenGCN(nn.Module):
    """transformer/models.py:487-551 (graph-level regression / classification shell)."""

    def __init__(self, in_size, nb_class, d_model, nb_heads, dim_feedforward=2048, dropout=0.1,
                 nb_layers=4, batch_norm=False, lap_pos_enc=False, lap_pos_enc_dim=0,
                 filter_order=4, gnn_type='ChebConvDynamic', last_layer_filter=True,
                 learn_only_filter_order_coeff=False, heads_share_graph=False, filter_mode='cheb',
                 tie_qk=False):
        super().__init__()
        self.lap_pos_enc = lap_pos_enc
        self.lap_pos_enc_dim = lap_pos_enc_dim
        if lap_pos_enc and lap_pos_enc_dim > 0:
            self.embedding_lap_pos_enc = nn.Linear(lap_pos_enc_dim, d_model)
        self.embedding = nn.Linear(in_features=in_size, out_features=d_model, bias=False)
        encoder_layer = DiffTransformerEncoderLayer(d_model, nb_heads, dim_feedforward, dropout,
                                                    batch_norm=batch_norm, tie_qk=tie_qk)
        self.encoder = DiffTransformerEncoderGenGCN(
            d_model, nb_heads, encoder_layer, nb_layers, num_coefficients=filter_order,
            gnn_type=gnn_type, last_layer_filter=last_layer_filter,
            learn_only_filter_order_coeff=learn_only_filter_order_coeff,
            heads_share_graph=heads_share_graph, filter_mode=filter_mode)
        # the reference also registers an unused GCNConv(d,d) here (:508, forward block commented
        # out :534-541); kept so that checkpoints load and gradient buckets see grad=None params.
        self.gcn = DenseGCNParams(d_model, d_model)
        self.pooling = GlobalAvg1D()
        self.classifier = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(True),
                                        nn.Linear(d_model, nb_class))

    def forward(self, x, edge_index, batch, feature_indices, masks, pe, x_lap_pos_enc=None,
                degree=None, regularization=0.0, return_filter_coeff=False, graph_cache=None):
        output = input_embedding(self.embedding, x.permute(1, 0, 2))                              # :521-522
        if self.lap_pos_enc and x_lap_pos_enc is not None:
            output = output + self.embedding_lap_pos_enc(x_lap_pos_enc.transpose(0, 1))   # :523-526
        output, attn, filter_coeff = self.encoder(output, pe, edge_index, feature_indices, batch,
                                                  degree=degree, src_key_padding_mask=masks,
                                                  graph_cache=graph_cache)       # :527
        pooled = self.pooling(output.permute(1, 0, 2), masks)                    # :528,532
        reg = 0
        if regularization > 0:
            reg = torch.norm(filter_coeff, p=2, dim=[1, 2]).mean()               # :578 (what :554-584 return)
        out = self.classifier(pooled)
        if return_filter_coeff:
            return out, reg, filter_coeff
        return out, reg                                                           # :548-551


def regularisation_max_cos(coeff):
    """transformer/models.py:727-742, :1078-1093: largest off-diagonal cosine between the heads'
    coefficient vectors of a graph, summed over graphs."""
    gm = torch.bmm(coeff, coeff.permute(0, 2, 1))
    gm = gm * (1.0 - torch.eye(coeff.shape[1], device=coeff.device, dtype=coeff.dtype)).unsqueeze(0)
    v1 = torch.norm(coeff, p=2, dim=2)
    reg = gm / torch.bmm(v1.unsqueeze(-1), v1.unsqueeze(1))
    return reg.max(dim=1).values.max(dim=1).values.sum()


# ogb.utils.features.get_atom_feature_dims(): atomic number, chirality, degree, formal charge,
# number of H, radical electrons, hybridisation, aromatic, in-ring.  ogb is not pinned by the
# reference (README.md:21-33); later ogb releases use 5 chirality / 7 hybridisation classes -
# pass ``feature_dims`` to match a checkpoint.
ATOM_FEATURE_DIMS = (119, 4, 12, 12, 10, 6, 6, 2, 2)
BOND_FEATURE_DIMS = (5, 6, 2)


class AtomEncoder(nn.Module):
    """Stand-alone counterpart of ``ogb.graphproppred.mol_encoder.AtomEncoder`` (imported at
    transformer/models.py:12, used at :619,646): one embedding table per integer atom feature,
    xavier-uniform, summed.  State-dict keys ``atom_embedding_list.{i}.weight`` as in ogb."""

    list_name = 'atom_embedding_list'

    def __init__(self, emb_dim, feature_dims=ATOM_FEATURE_DIMS):
        super().__init__()
        tables = nn.ModuleList()
        for dim in feature_dims:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            tables.append(emb)
        setattr(self, self.list_name, tables)

    def forward(self, x):
        tables = getattr(self, self.list_name)
        out = 0
        for i in range(x.shape[1]):
            out = out + tables[i](x[:, i])
        return out


class BondEncoder(AtomEncoder):
    """ogb BondEncoder; registered but never called by the reference (transformer/models.py:620-621)."""

    list_name = 'bond_embedding_list'

    def __init__(self, emb_dim, feature_dims=BOND_FEATURE_DIMS):
        super().__init__(emb_dim, feature_dims)


class DiffGraphTransformerGenGCNMolHiv(nn.Module):
    """transformer/models.py:598-742 (ogbg-molhiv shell, BASELINE config 5): AtomEncoder embedding
    of integer node features, FeTA encoder, masked mean pooling, 2-layer classifier whose
    ``nn.LeakyReLU(True)`` is the identity (negative_slope = True = 1.0, :637), logits + sigmoid."""

    def __init__(self, in_size, nb_class, d_model, nb_heads, dim_feedforward=2048, dropout=0.1,
                 nb_layers=4, batch_norm=False, lap_pos_enc=False, lap_pos_enc_dim=0,
                 filter_order=4, gnn_type='ChebConvDynamic', last_layer_filter=True,
                 learn_only_filter_order_coeff=False, use_skip_conn=True, use_default_encoder=False,
                 heads_share_graph=False, filter_mode='cheb', tie_qk=False,
                 atom_feature_dims=ATOM_FEATURE_DIMS):
        super().__init__()
        self.lap_pos_enc = lap_pos_enc
        self.lap_pos_enc_dim = lap_pos_enc_dim
        if lap_pos_enc and lap_pos_enc_dim > 0:
            self.embedding_lap_pos_enc = nn.Linear(lap_pos_enc_dim, d_model)
        self.d_model = d_model
        self.embedding = AtomEncoder(d_model, atom_feature_dims)                 # :619
        self.edge_embeddings = BondEncoder(d_model)                              # :621, unused
        encoder_layer = DiffTransformerEncoderLayer(d_model, nb_heads, dim_feedforward, dropout,
                                                    batch_norm=batch_norm, tie_qk=tie_qk)
        self.use_default_encoder = use_default_encoder
        if use_default_encoder:
            # the reference passes (d_model, nb_heads, layer, n) to a 2-argument constructor here
            # (:626) and would raise; the plain encoder is built with the arguments it takes
            self.encoder = DiffTransformerEncoder(encoder_layer, nb_layers)
        else:
            self.encoder = DiffTransformerEncoderGenGCN(
                d_model, nb_heads, encoder_layer, nb_layers, num_coefficients=filter_order,
                gnn_type=gnn_type, last_layer_filter=last_layer_filter,
                learn_only_filter_order_coeff=learn_only_filter_order_coeff,
                use_skip_conn=use_skip_conn, heads_share_graph=heads_share_graph,
                filter_mode=filter_mode)
        self.gcn = DenseGCNParams(d_model, d_model)                              # :630, unused
        self.pooling = GlobalAvg1D()
        self.classifier = nn.Sequential(nn.Linear(d_model, d_model), nn.LeakyReLU(True),
                                        nn.Linear(d_model, nb_class))            # :635-639
        self.sigmoid = nn.Sigmoid()

    def forward(self, x, edge_index, batch, feature_indices, masks, pe, x_lap_pos_enc=None,
                degree=None, regularization=0.0, return_filter_coeff=False, graph_cache=None):
        emb = self.embedding(x.reshape(-1, x.shape[-1]).to(torch.long))          # :645-646
        output = emb.reshape(x.shape[0], x.shape[1], self.d_model).permute(1, 0, 2)   # :648,667
        if self.lap_pos_enc and x_lap_pos_enc is not None:
            output = output + self.embedding_lap_pos_enc(x_lap_pos_enc.transpose(0, 1))
        if self.use_default_encoder:
            output = self.encoder(output, pe, degree=degree, src_key_padding_mask=masks)
            filter_coeff = None
        else:
            output, attn, filter_coeff = self.encoder(output, pe, edge_index, feature_indices, batch,
                                                      degree=degree, src_key_padding_mask=masks,
                                                      graph_cache=graph_cache)   # :674
        pooled = self.pooling(output.permute(1, 0, 2), masks)                    # :675,679
        reg = self.regularisation(filter_coeff) if regularization > 0 else 0     # :715-718
        cls_out = self.classifier(pooled)                                        # :720
        if return_filter_coeff:
            return cls_out.squeeze(), reg, self.sigmoid(cls_out).squeeze(), filter_coeff
        return cls_out.squeeze(), reg, self.sigmoid(cls_out).squeeze()           # :722-725

    def regularisation(self, coeff):
        return regularisation_max_cos(coeff)


class DiffGraphTransformerGenGCNSBM(nn.Module):
    """transformer/models.py:1008-1110 (PATTERN/CLUSTER node classification, BASELINE config 4):
    no pooling - the classifier runs on every node and the logits of the real nodes are returned
    graph-major, ``[N_tot, nb_class]`` (:1069-1071)."""

    def __init__(self, in_size, nb_class, d_model, nb_heads, dim_feedforward=2048, dropout=0.1,
                 nb_layers=4, batch_norm=False, lap_pos_enc=False, lap_pos_enc_dim=0,
                 filter_order=4, gnn_type='ChebConvDynamic', last_layer_filter=True,
                 learn_only_filter_order_coeff=False, heads_share_graph=False, filter_mode='cheb',
                 tie_qk=False):
        super().__init__()
        self.lap_pos_enc = lap_pos_enc
        self.lap_pos_enc_dim = lap_pos_enc_dim
        if lap_pos_enc and lap_pos_enc_dim > 0:
            self.embedding_lap_pos_enc = nn.Linear(lap_pos_enc_dim, d_model)
        self.embedding = nn.Linear(in_features=in_size, out_features=d_model, bias=False)
        encoder_layer = DiffTransformerEncoderLayer(d_model, nb_heads, dim_feedforward, dropout,
                                                    batch_norm=batch_norm, tie_qk=tie_qk)
        self.encoder = DiffTransformerEncoderGenGCN(
            d_model, nb_heads, encoder_layer, nb_layers, num_coefficients=filter_order,
            gnn_type=gnn_type, last_layer_filter=last_layer_filter,
            learn_only_filter_order_coeff=learn_only_filter_order_coeff,
            heads_share_graph=heads_share_graph, filter_mode=filter_mode)
        self.classifier = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(True),
                                        nn.Linear(d_model, nb_class))
        # the reference's loss() reads self.n_classes / self.device, which its constructor never
        # sets (:1101) - it would raise; they are derived here
        self.n_classes = nb_class

    def forward(self, x, edge_index, batch, feature_indices, masks, pe, x_lap_pos_enc=None,
                degree=None, regularization=0.0, return_filter_coeff=False, graph_cache=None,
                padded_logits=False):
        """padded_logits=True returns the logits of every position [B, N_pad, nb_class] instead of
        the boolean gather of the real nodes (whose length differs per batch: not capturable)."""
        output = input_embedding(self.embedding, x.permute(1, 0, 2))                              # :1042-1043
        if self.lap_pos_enc and x_lap_pos_enc is not None:
            output = output + self.embedding_lap_pos_enc(x_lap_pos_enc.transpose(0, 1))
        output, attn, filter_coeff = self.encoder(output, pe, edge_index, feature_indices, batch,
                                                  degree=degree, src_key_padding_mask=masks,
                                                  graph_cache=graph_cache)       # :1048
        reg = self.regularisation(filter_coeff) if regularization > 0 else 0
        cls_output = self.classifier(output.permute(1, 0, 2))                    # :1069
        if not padded_logits:
            cls_output = cls_output[~masks]                                      # :1070-1071
        if return_filter_coeff:
            return cls_output, reg, filter_coeff
        return cls_output, reg

    def regularisation(self, coeff):
        return regularisation_max_cos(coeff)

    def loss(self, pred, label):
        """:1095-1110, class weights (V - |class|) / V of the classes present, without the host
        round trips of bincount/nonzero/unique."""
        v = label.shape[0]
        sizes = torch.bincount(label, minlength=self.n_classes)
        weight = (v - sizes).to(pred.dtype) / v * (sizes > 0).to(pred.dtype)
        return F.cross_entropy(pred, label, weight=weight)
