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
constructor flags (``tie_qk``, ``in_proj_bias``) rather than guesses.

The score/softmax/weighted-sum runs in feta_attn_fwd/bwd; projections and norms are rocBLAS /
PyTorch ops on the same stream.
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


class DiffMultiheadAttention(nn.Module):
    """Parameter names follow nn.MultiheadAttention (in_proj_weight [3d,d], out_proj.*) so that
    reference checkpoints (``encoder.layers.{i}.self_attn.*``) load."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=False, tie_qk=False):
        super().__init__()
        assert embed_dim % num_heads == 0
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

    def forward(self, query, key, value, pe=None, key_padding_mask=None, need_weights=True,
                attn_mask=None, need_heads=False, n_real=None):
        if key is not query or value is not query:
            raise NotImplementedError('self-attention only (query is key is value)')
        if attn_mask is not None:
            raise NotImplementedError('attn_mask is not used on the FeTA path')
        if self.training and self.dropout > 0.0:
            raise NotImplementedError('attention-probability dropout is not built; the FeTA '
                                      'scripts default to --dropout 0.0 '
                                      '(experiments/run_transformer_gengcn.py:47)')
        n, b, _ = query.shape
        if n_real is None:
            if key_padding_mask is None:
                n_real = torch.full((b,), n, dtype=torch.int32, device=query.device)
            else:
                n_real = n_real_from_mask(key_padding_mask)
        qkv = F.linear(query, self.in_proj_weight, self.in_proj_bias)
        concat, attn = FF.attention_core(qkv, pe, n_real, self.num_heads, need_attn=need_weights,
                                         tie_qk=self.tie_qk, batch_first=False)
        out = self.out_proj(concat)
        if need_heads:
            heads = concat.view(n, b, self.num_heads, self.head_dim).permute(1, 0, 2, 3)
            return out, attn, heads
        return out, attn


class DiffTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu',
                 batch_norm=False, tie_qk=False, in_proj_bias=False):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError('relu only')
        self.self_attn = DiffMultiheadAttention(d_model, nhead, dropout=dropout, bias=in_proj_bias,
                                                tie_qk=tie_qk)
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

    def _norm(self, mod, x):
        if self.batch_norm:   # statistics over all N*B rows, padded ones included
            shp = x.shape
            return mod(x.reshape(-1, shp[-1])).view(shp)
        return mod(x)

    def forward(self, src, pe=None, degree=None, src_mask=None, src_key_padding_mask=None,
                need_heads=False, n_real=None, need_weights=True):
        res = self.self_attn(src, src, src, pe=pe, key_padding_mask=src_key_padding_mask,
                             attn_mask=src_mask, need_heads=need_heads, n_real=n_real,
                             need_weights=need_weights)
        src2, attn = res[0], res[1]
        if degree is not None:
            src2 = degree.transpose(0, 1).contiguous().unsqueeze(-1) * src2
        src = self._norm(self.norm1, src + self.dropout1(src2))
        src2 = self.linear2(self.dropout(F.relu(self.linear1(src))))
        src = self._norm(self.norm2, src + self.dropout2(src2))
        if need_heads:
            return src, attn, res[2]
        return src, attn


def clone_layers(layer, n):
    """nn.TransformerEncoder semantics: n deep copies (identical initial weights)."""
    return nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
