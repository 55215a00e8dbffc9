"""ChebConvDynamic with the reference's constructor / forward signature
(transformer/ChebNetDynamic.py:80-81,132-133), computed by the MI355X kernels.

The reference operates on the gathered node list ``x [M, C]`` with a batched ``edge_index``
and gathers a per-node copy of its group's weights (``repeat_interleave``, :148-149, ~49 MB per
ZINC batch).  Here the node list is re-packed into padded per-group tiles, the batched edge
list becomes one dense scaled Laplacian per group (feta_lhat_from_edges) and the filter runs
as one wave per group with the weights read once (feta_cheb_filter_fwd/bwd).  Groups that
``edge_index`` does not cover get Lhat = 0 exactly as in the reference.

The fused encoder (transformer/models.py of this package) does not go through this class's
re-packing; it calls the same kernels on the padded activations directly.
"""
import math

import torch
from torch import nn

from .. import functional as FF


class ChebConvDynamic(nn.Module):
    def __init__(self, in_channels, out_channels, K, normalization='sym', bias=True,
                 learn_only_filter_order_coeff=False, **kwargs):
        super().__init__()
        assert K > 0                                                  # reference :85
        assert normalization in [None, 'sym', 'rw'], 'Invalid normalization'   # :86
        if normalization != 'sym':
            raise NotImplementedError("only normalization='sym' (lambda_max = 2) is built")
        if in_channels != out_channels:
            raise NotImplementedError('the kernels implement the square per-head filter '
                                      '(in_channels == out_channels), as used by FeTA')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.normalization = normalization
        self.order = K
        self.learn_only_filter_order_coeff = learn_only_filter_order_coeff
        if learn_only_filter_order_coeff:
            self.weight = nn.Parameter(torch.empty(K, in_channels, out_channels))   # :91-92
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))                    # :95-96
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.learn_only_filter_order_coeff:      # glorot, reference :20-23,103-104
            stdv = math.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))
            self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.fill_(0)                 # :25-27,105

    def group_weights(self, filter_coeff):
        """[P, G, in, out] (or [P, G] scalars in learn_only_filter_order_coeff mode) -> [G, P*in*out]."""
        if self.learn_only_filter_order_coeff:
            w = filter_coeff.permute(1, 0)[:, :, None, None] * self.weight[None]    # :165,173,181
        else:
            w = filter_coeff.permute(1, 0, 2, 3)
        return w.reshape(w.shape[0], -1)

    def forward(self, x, edge_index, filter_coeff, edge_weight=None, batch=None, lambda_max=None):
        """x [M, C] gathered nodes, grouped by ascending ``batch``; filter_coeff [P, G, C, C]."""
        if self.normalization != 'sym' and lambda_max is None:
            raise ValueError('You need to pass `lambda_max` to `forward() in`'
                             'case the normalization is non-symmetric.')     # reference :135-137
        if edge_weight is not None or lambda_max is not None:
            raise NotImplementedError('edge_weight / lambda_max are not used on the FeTA path')
        if batch is None:
            raise ValueError('batch is required (the reference raises NameError without it, '
                             'transformer/ChebNetDynamic.py:146-156)')
        groups = filter_coeff.shape[1]
        m = x.shape[0]
        batch = batch.long()
        counts = torch.bincount(batch, minlength=groups)
        n_pad = int(counts.max().item())             # host sync: only on this re-packing path
        off = torch.cumsum(counts, 0) - counts
        local = torch.arange(m, device=x.device) - off[batch]
        xp = torch.zeros((groups, n_pad, 1, self.in_channels), dtype=x.dtype, device=x.device)
        xp[batch, local, 0] = x
        lhat = FF.lhat_from_edges(edge_index, batch, off.to(torch.int32), groups, n_pad)
        y = FF.cheb_filter(xp, lhat, self.group_weights(filter_coeff), self.bias,
                           counts.to(torch.int32), self.order, heads_share_graph=True,
                           batch_first=True)
        return y[batch, local, 0]

    def __repr__(self):
        return '{}({}, {}, K={}, normalization={})'.format(
            self.__class__.__name__, self.in_channels, self.out_channels, self.order,
            self.normalization)
