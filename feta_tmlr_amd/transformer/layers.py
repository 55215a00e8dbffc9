"""DiffTransformerEncoderLayer / DiffMultiheadAttention for the MI355X.

The reference imports ``DiffTransformerEncoderLayer`` from ``transformer/layers.py``
(transformer/models.py:4) but ships a copy of gckn/layers.py under that name, so there is no
source to follow.  The contract is taken from the call sites:
  ctor  (d_model, nb_heads, dim_feedforward, dropout, batch_norm=)   transformer/models.py:505-506
  call  mod(src, pe=, degree=, src_mask=, src_key_padding_mask=, need_heads=True)
        -> (src', attn [B,H,N,N], out_each_head [B,N,H,dh])          transformer/models.py:166-167,179,244
        -> (src', attn) without need_heads                            transformer/models.py:92-93
  .self_attn returns a tuple whose [1] is attn                        experiments/visu_attention.py:326-329
and the body from the upstream GraphiT layer the README credits (README.md:129): scaled
dot-product scores, padded keys masked, exp(s - rowmax), multiplied by the positional kernel
``pe`` (broadcast over heads), normalised by clamp(rowsum, 1e-6) - the same form as the in-tree
DGL variants (LSPE/layers/graphit_gt_layer.py:39-43,120-131,164) - then out_proj, the optional
``degree`` scaling, residual + norm + FFN + residual + norm.  Choices the call sites do not pin are
constructor flags (``tie_qk``, ``in_proj_bias``, ``stab`` = 'rowmax' | 'clamp5': the DGL variants clamp the score to
+-5 instead of subtracting the row maximum, LSPE/layers/graphit_gt_layer.py:39-43) rather than guesses.

Kernels: the score/softmax/weighted-sum runs in feta_attn_fwd/bwd; every linear with its
element-wise neighbours (bias, relu, degree scaling, residual) and the BatchNorm statistics runs in
feta_rowlin_fwd/bwd; BatchNorm in feta_bn_apply_fwd / feta_bn_bwd.  LayerNorm and (training-mode)
dropout, when configured, stay PyTorch ops on the same stream.
"""
import copy

import torch
from torch import nn
import torch.nn.functional as F

from .. import functional as FF


def n_real_from_mask(key_padding_mask):
    """[B,N] bool (True = pad) -> int32 node counts on the same device (no host sync).
    Padding must be a suffix, as produced by the reference collate (transformer/data.py:210)."""
    return (~key_padding_mask).sum(dim=-1, dtype=torch.int32)


def linear_rows(x2d, weight, bias, rowscale=None, residual=None, relu=False, want_stats=False, stats_shift=None):
    """feta_rowlin on [M, KI] rows when the dims are instantiated, else the same arithmetic with
    rocBLAS + PyTorch ops (still on the GPU).  -> (y, stats or None)"""
    if FF.row_linear_supported(x2d.shape[1], weight.shape[0]):
        return FF.row_linear(x2d, weight, bias, rowscale, residual, relu, want_stats, stats_shift)
    y = F.linear(x2d, weight, bias)
    if relu:
        y = F.relu(y)
    if rowscale is not None:
        y = y * rowscale.unsqueeze(-1)
    if residual is not None:
        y = y + residual
    return y, None


class DiffMultiheadAttention(nn.Module):
    """Parameter names follow nn.MultiheadAttention (in_proj_weight [3d,d], out_proj.*) so that
    reference checkpoints (``encoder.layers.{i}.self_attn.*``) load."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=False, tie_qk=False, stab='rowmax'):
        super().__init__()
        assert embed_dim % num_heads == 0
        assert stab in ('rowmax', 'clamp5')
        self.stab = stab     # exp(s - rowmax) (upstream GraphiT) | exp(clamp(s, -5, 5)) (the in-tree DGL witnesses)
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.dropout = dropout
        self.tie_qk = tie_qk
        self.batch_first = False
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        if bias:
            self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter('in_proj_bias', None)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)   # torch 1.6: always biased
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        if self.in_proj_bias is not None:
            nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def core(self, query, pe, key_padding_mask, need_weights, n_real):
        """in_proj + attention core -> (concat [N,B,d] before out_proj, attn or None)."""
        n, b, d = query.shape
        if n_real is None:
            if key_padding_mask is None:
                n_real = torch.full((b,), n, dtype=torch.int32, device=query.device)
            else:
                n_real = n_real_from_mask(key_padding_mask)
        qkv, _ = linear_rows(query.reshape(n * b, d), self.in_proj_weight, self.in_proj_bias)
        return FF.attention_core(qkv.view(n, b, 3 * d), pe, n_real, self.num_heads,
                                 need_attn=need_weights, tie_qk=self.tie_qk, batch_first=False,
                                 dropout_p=self.dropout if self.training else 0.0, stab=self.stab)

    def forward(self, query, key, value, pe=None, key_padding_mask=None, need_weights=True,
                attn_mask=None, need_heads=False, n_real=None):
        if key is not query or value is not query:
            raise NotImplementedError('self-attention only (query is key is value)')
        if attn_mask is not None:
            raise NotImplementedError('attn_mask is not used on the FeTA path')
        n, b, d = query.shape
        concat, attn = self.core(query, pe, key_padding_mask, need_weights, n_real)
        out, _ = linear_rows(concat.reshape(n * b, d), self.out_proj.weight, self.out_proj.bias)
        out = out.view(n, b, d)
        if need_heads:
            heads = concat.view(n, b, self.num_heads, self.head_dim).permute(1, 0, 2, 3)
            return out, attn, heads
        return out, attn


class DiffTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu',
                 batch_norm=False, tie_qk=False, in_proj_bias=False, stab='rowmax'):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError('relu only')
        self.self_attn = DiffMultiheadAttention(d_model, nhead, dropout=dropout, bias=in_proj_bias,
                                                tie_qk=tie_qk, stab=stab)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.batch_norm = batch_norm
        if batch_norm:
            self.norm1 = nn.BatchNorm1d(d_model)
            self.norm2 = nn.BatchNorm1d(d_model)
        else:
            self.norm1 = nn.LayerNorm(d_model)
            self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.storage_dtype = torch.float32   # torch.bfloat16: the bf16 storage path (set_storage_dtype)

    def _forward_lowp(self, src, pe, degree_rows, n_real, need_heads, need_weights):
        """bf16 STORAGE path (BASELINE configs 3 / 5; the reference has no reduced-precision mode): activations,
        pe and attn live in HBM as bf16; the attention core runs in the bf16 kernels (feta_attn_fwd/bwd_bf16, bf16
        MFMA, fp32 softmax statistics), the linears are plain library bf16 GEMMs on bf16 copies of the fp32 master
        weights (gradients arrive on the masters in fp32), BatchNorm / LayerNorm statistics are computed in fp32."""
        dt = self.storage_dtype
        n, b, d = src.shape
        m = n * b
        a = self.self_attn
        cast = lambda p: None if p is None else p.to(dt)
        x0 = src.to(dt)
        qkv = F.linear(x0, cast(a.in_proj_weight), cast(a.in_proj_bias))
        concat, attn = FF.attention_core(qkv, pe, n_real, a.num_heads, need_attn=need_weights, tie_qk=a.tie_qk,
                                         batch_first=False, dropout_p=a.dropout if self.training else 0.0, stab=a.stab)
        src2 = F.linear(concat, cast(a.out_proj.weight), cast(a.out_proj.bias))
        if degree_rows is not None:
            src2 = src2 * degree_rows.view(n, b, 1).to(dt)
        y1 = x0 + self.dropout1(src2)
        x1 = self.norm1(y1.reshape(m, d).float()).to(dt).view(n, b, d)
        h = self.dropout(F.relu(F.linear(x1, cast(self.linear1.weight), cast(self.linear1.bias))))
        y2 = x1 + self.dropout2(F.linear(h, cast(self.linear2.weight), cast(self.linear2.bias)))
        out = self.norm2(y2.reshape(m, d).float()).to(dt).view(n, b, d)
        if need_heads:
            return out, attn, concat.view(n, b, a.num_heads, a.head_dim).permute(1, 0, 2, 3)
        return out, attn

    def _norm(self, mod, y, stats):
        """y [M, d] -> normalised [M, d].  BatchNorm statistics run over all N*B rows, padded ones
        included, exactly as nn.BatchNorm1d on the [N*B, d] view does."""
        if not self.batch_norm:
            if (isinstance(mod, nn.LayerNorm) and mod.elementwise_affine and y.dim() == 2
                    and FF.layer_norm_rows_supported(y.shape[1])):
                return FF.layer_norm_rows(y, mod.weight, mod.bias, mod.eps)
            return mod(y)
        if mod.training and mod.momentum is not None and y.shape[1] % 4 == 0 and y.shape[1] <= 256:
            return FF.batch_norm_train(y, stats, mod.weight, mod.bias, mod.running_mean,
                                       mod.running_var, mod.momentum, mod.eps, mod.num_batches_tracked)
        return mod(y)

    def forward(self, src, pe=None, degree=None, src_mask=None, src_key_padding_mask=None,
                need_heads=False, n_real=None, need_weights=True, degree_rows=None):
        """src [N,B,d] seq-first.  ``degree_rows`` (optional, this package) is ``degree`` already laid
        out per row of the [N*B, d] view, so the encoder computes it once for all layers."""
        if src_mask is not None:
            raise NotImplementedError('attn_mask is not used on the FeTA path')
        n, b, d = src.shape
        m = n * b
        if self.storage_dtype != torch.float32:
            if degree is not None and degree_rows is None:
                degree_rows = degree.transpose(0, 1).reshape(m).contiguous()
            if n_real is None:
                n_real = (n_real_from_mask(src_key_padding_mask) if src_key_padding_mask is not None else
                          torch.full((b,), n, dtype=torch.int32, device=src.device))
            return self._forward_lowp(src, pe, degree_rows, n_real, need_heads, need_weights)
        fuse_drop = not (self.training and self.dropout1.p > 0.0)
        x0 = src.reshape(m, d)
        concat, attn = self.self_attn.core(src, pe, src_key_padding_mask, need_weights, n_real)
        if degree is not None and degree_rows is None:
            degree_rows = degree.transpose(0, 1).reshape(m).contiguous()
        op = self.self_attn.out_proj
        if fuse_drop:
            # y1 = x0 + degree * (concat W_o^T + b_o), BN statistics of y1 in the same launch
            y1, st1 = linear_rows(concat.reshape(m, d), op.weight, op.bias, rowscale=degree_rows,
                                  residual=x0, want_stats=self.batch_norm,
                                  stats_shift=self.norm1.running_mean if self.batch_norm else None)
        else:
            src2, _ = linear_rows(concat.reshape(m, d), op.weight, op.bias, rowscale=degree_rows)
            y1, st1 = x0 + self.dropout1(src2), None
        x1 = self._norm(self.norm1, y1, st1)
        if fuse_drop:
            h, _ = linear_rows(x1, self.linear1.weight, self.linear1.bias, relu=True)
            y2, st2 = linear_rows(h, self.linear2.weight, self.linear2.bias, residual=x1,
                                  want_stats=self.batch_norm,
                                  stats_shift=self.norm2.running_mean if self.batch_norm else None)
        else:
            h, _ = linear_rows(x1, self.linear1.weight, self.linear1.bias, relu=True)
            src2, _ = linear_rows(self.dropout(h), self.linear2.weight, self.linear2.bias)
            y2, st2 = x1 + self.dropout2(src2), None
        out = self._norm(self.norm2, y2, st2).view(n, b, d)
        if need_heads:
            heads = concat.view(n, b, self.self_attn.num_heads, self.self_attn.head_dim).permute(1, 0, 2, 3)
            return out, attn, heads
        return out, attn


def set_storage_dtype(module, dtype):
    """Switch every FeTA encoder / encoder layer below `module` to a storage dtype: torch.float32 (the reference's
    arithmetic, fused kernels) or torch.bfloat16 (bf16 activations / pe / U / attn / per-block filter weights, bf16
    MFMA, fp32 statistics and accumulators, fp32 master weights and gradients)."""
    assert dtype in (torch.float32, torch.bfloat16)
    for mod in module.modules():
        if hasattr(mod, 'storage_dtype'):
            mod.storage_dtype = dtype
    return module


def clone_layers(layer, n):
    """nn.TransformerEncoder semantics: n deep copies (identical initial weights)."""
    return nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
