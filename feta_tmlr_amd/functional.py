"""torch.autograd.Function wrappers over the C ABI (include/feta_hip.h).

Token tensors are handed to the kernels as [B, N, H, dh] *views* with explicit strides, so the
reference's seq-first [N, B, d] activations (transformer/models.py:521-527) are consumed and
produced in place: no transposes, no gathers, no scatter into zero-initialised buffers.
"""
import torch

from . import _lib


def _token_view(t, batch_first, heads):
    """[N,B,d] (seq-first) or [B,N,d] (batch-first) -> [B,N,H,dh] view (no copy)."""
    d = t.shape[-1]
    v = t.view(t.shape[0], t.shape[1], heads, d // heads)
    return v if batch_first else v.permute(1, 0, 2, 3)


def _dense_like(t, batch_first):
    """grad arriving as a [B,N,H,dh]-shaped tensor -> same values in the storage order the
    kernels were given (dense (H,dh), 16-byte aligned rows)."""
    if batch_first:
        return t.contiguous()
    return t.permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3)


def _new_token(b, n, h, dh, batch_first, ref):
    """same storage dtype as ref (fp32, or bf16 on the bf16 storage path)"""
    if batch_first:
        return torch.empty((b, n, h, dh), dtype=ref.dtype, device=ref.device)
    return torch.empty((n, b, h, dh), dtype=ref.dtype, device=ref.device).permute(1, 0, 2, 3)


class AttentionCoreFn(torch.autograd.Function):
    """qkv (projected, [N,B,3d] or [B,N,3d]) -> (concat heads in the same token layout,
    attn [B,H,N,N] or None).  C ABI: feta_attn_fwd / feta_attn_bwd."""

    @staticmethod
    def forward(ctx, qkv, pe, n_real, num_heads, need_attn, tie_qk, batch_first, drop=None, clamp5=False):
        abi, stream = _lib.backend(qkv, pe, n_real)
        ctx.set_materialize_grads(False)   # no zero tensor for the non-differentiable attn output
        qkv = qkv.contiguous()
        l0, l1, d3 = qkv.shape
        d = d3 // 3
        dh = d // num_heads
        b, n = (l0, l1) if batch_first else (l1, l0)
        v5 = qkv.view(l0, l1, 3, num_heads, dh)
        sel = (lambda i: v5[:, :, i]) if batch_first else (lambda i: v5[:, :, i].permute(1, 0, 2, 3))
        q, k, v = sel(0), (sel(0) if tie_qk else sel(1)), sel(2)
        out = _new_token(b, n, num_heads, dh, batch_first, qkv)
        attn = torch.empty((b, num_heads, n, n), dtype=qkv.dtype, device=qkv.device) if need_attn else None
        stats = torch.empty((b, num_heads, n, 2), dtype=torch.float32, device=qkv.device)
        pe_c = None if pe is None else pe.to(qkv.dtype).contiguous()   # (bf16 storage: pe travels as bf16 too)
        scale = float(dh) ** -0.5
        abi.attn_fwd(q, k, v, pe_c, n_real, out, attn, stats, scale, stream, drop=drop, clamp5=clamp5)
        ctx.clamp5 = bool(clamp5)
        ctx.save_for_backward(qkv, pe_c, n_real, out, stats)
        ctx.cfg = (num_heads, tie_qk, batch_first, scale)
        ctx.drop = drop
        concat = (out if batch_first else out.permute(1, 0, 2, 3)).reshape(l0, l1, d)
        if attn is not None:
            ctx.mark_non_differentiable(attn)   # enters A2 detached (transformer/models.py:282)
        return concat, attn

    @staticmethod
    def backward(ctx, dconcat, _dattn):
        qkv, pe_c, n_real, out, stats = ctx.saved_tensors
        if dconcat is None:
            return (None,) * 9
        num_heads, tie_qk, batch_first, scale = ctx.cfg
        abi, stream = _lib.backend(qkv)
        l0, l1, d3 = qkv.shape
        d = d3 // 3
        dh = d // num_heads
        b, n = (l0, l1) if batch_first else (l1, l0)
        v5 = qkv.view(l0, l1, 3, num_heads, dh)
        dqkv = torch.empty_like(qkv)
        g5 = dqkv.view(l0, l1, 3, num_heads, dh)
        if batch_first:
            sel, gsel = (lambda i: v5[:, :, i]), (lambda i: g5[:, :, i])
        else:
            sel = lambda i: v5[:, :, i].permute(1, 0, 2, 3)
            gsel = lambda i: g5[:, :, i].permute(1, 0, 2, 3)
        q, k, v = sel(0), (sel(0) if tie_qk else sel(1)), sel(2)
        dout = _token_view(dconcat.to(qkv.dtype).contiguous(), batch_first, num_heads)
        delta = torch.empty((b, num_heads, n), dtype=torch.float32, device=qkv.device)
        abi.attn_bwd(q, k, v, pe_c, n_real, out, dout, stats, delta, gsel(0), gsel(1), gsel(2),
                     scale, stream, drop=ctx.drop, clamp5=ctx.clamp5)
        if tie_qk:
            dqkv[..., :d] += dqkv[..., d:2 * d]
            dqkv[..., d:2 * d] = 0
        return dqkv, None, None, None, None, None, None, None, None


class FilterCoefficientsFn(torch.autograd.Function):
    """attn [B,H,N,N] (detached) + GCN parameters -> pooled [H*B, C]
    (transformer/models.py:240-283 collapsed; C ABI: feta_colsum, feta_coeff_fwd/bwd)."""

    @staticmethod
    def forward(ctx, attn, n_real, gcn_weight, gcn_bias, pending=None):
        abi, stream = _lib.backend(attn, gcn_weight)
        ctx.pending = pending
        ctx.params = (gcn_weight, gcn_bias)
        if pending is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]):
            pending.coeff_armed = True      # this node's backward is the last of the filter stage: it flushes
        attn = attn.contiguous()
        b, h, n, _ = attn.shape
        c = gcn_weight.shape[1]
        dev = attn.device
        if pending is not None and pending.s is not None:
            s = pending.s                 # requested by the encoder ...
            left = pending.take_fwd()     # ... and computed inside the layer stack's first launch, or not yet
            if left:
                abi.colsum_multi(left, stream)
            pending.s = None
        else:
            s = torch.empty(c, dtype=torch.float32, device=dev)
            abi.colsum(gcn_weight.contiguous(), s, stream)
        gb = gcn_bias.contiguous()
        if pending is not None and pending.coeff_fwd_out is not None:
            cj, pooled = pending.coeff_fwd_out      # ran in the launch of the last layer's feed-forward half
            pending.coeff_fwd_out = None
        else:
            cj = torch.empty((h * b, n), dtype=torch.float32, device=dev)
            pooled = torch.empty((h * b, c), dtype=torch.float32, device=dev)
            abi.coeff_fwd(attn, n_real, s, gb, cj, pooled, stream)
        if pending is not None:
            pending.coeff_fwd_req = None
        ctx.save_for_backward(cj, n_real, s, gb)
        ctx.dims = (b, n, h, gcn_weight.shape[0])
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        cj, n_real, s, gb = ctx.saved_tensors
        b, n, h, rows = ctx.dims
        abi, stream = _lib.backend(cj)
        c = s.shape[0]
        groups = abi.coeff_bwd_groups(b, h)
        partial = torch.empty((groups, 2, c), dtype=torch.float32, device=cj.device)
        dsdb = torch.empty((2, c), dtype=torch.float32, device=cj.device)   # contiguous: one reduction
        ds, db = dsdb[0], dsdb[1]
        # s = 1^T W  =>  every row of dW equals ds: the reduction launch writes the dense rows itself (an
        # expanded view would be copied into a dense .grad by autograd: one more 4 MB kernel per step)
        dw = torch.empty((rows, c), dtype=torch.float32, device=cj.device)
        pend = ctx.pending
        if (pend is not None and pend.stack_armed and not pend.stack_done and PendingSums.untouched(*ctx.params)):
            # the layer stack's backward comes after this node: its first launch carries this node's kernel in trailing
            # workgroups (feta_ffn_bwd_coeff), its reduction launch the sums
            pend.coeff_bwd_req = (cj, n_real, s, gb, dpooled.contiguous(), partial, b, n, h)
            p2 = partial.view(groups, 2 * c)
            pend.add(p2[:, :c], ds, dw, owners=[(ctx.params[0], dw)])
            pend.add(p2[:, c:], db, owners=[(ctx.params[1], db)])
            return None, None, dw, db, None
        waiting = pend.take() if pend is not None else []
        if waiting:
            # ONE reduction launch for this node's partials and every column sum the nodes before it left pending
            abi.coeff_bwd(cj, n_real, s, gb, dpooled.contiguous(), partial, None, None, b, n, h, stream)
            p2 = partial.view(groups, 2 * c)
            abi.colsum_multi([(p2[:, :c], ds, dw), (p2[:, c:], db)] + waiting, stream)
        else:
            abi.coeff_bwd(cj, n_real, s, gb, dpooled.contiguous(), partial, ds, db, b, n, h, stream, dw_dense=dw)
        return None, None, dw, db, None


class DenseLinearFn(torch.autograd.Function):
    """y = x W^T + b for the C x C ``self.linear`` of the coefficient generator
    (transformer/models.py:284): the three GEMMs stay rocBLAS (plain library GEMMs), the bias
    gradient is one feta_colsum instead of a generic reduction kernel."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        with _lib.tuned_gemm():
            return torch.addmm(bias, x, w.t())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        abi, stream = _lib.backend(dy)
        dy = dy.contiguous()
        db = torch.empty(dy.shape[1], dtype=dy.dtype, device=dy.device)
        abi.colsum(dy, db, stream)
        with _lib.tuned_gemm():
            return dy.mm(w), dy.t().mm(x), db


def dense_linear(x, w, bias):
    return DenseLinearFn.apply(x, w, bias)


class _FilterFn(torch.autograd.Function):
    """x [B,N,H,dh] view, coeff [H*B, P*dh*dh], bias [dh] -> y (same token layout as x),
    zero on padded rows.  mode 'cheb': graph = lhat [B,N,N]; mode 'spec': graph = (u, lam)."""

    @staticmethod
    def forward(ctx, x, coeff, bias, n_real, g0, g1, mode, order, share, batch_first):
        abi, stream = _lib.backend(x, coeff)
        b, n, h, dh = x.shape
        xs = _dense_like(x, batch_first)
        coeff = coeff.contiguous()
        y = _new_token(b, n, h, dh, batch_first, x)
        if mode == 'cheb':
            abi.cheb_filter_fwd(xs, g0, coeff, bias, n_real, y, order, share, stream)
        else:
            abi.spec_filter_fwd(xs, g0, g1, coeff, bias, n_real, y, order, share, stream)
        ctx.save_for_backward(xs, coeff, n_real, g0, g1)
        ctx.cfg = (mode, order, share, batch_first, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, coeff, n_real, g0, g1 = ctx.saved_tensors
        mode, order, share, batch_first, has_bias = ctx.cfg
        abi, stream = _lib.backend(xs)
        b, n, h, dh = xs.shape
        dys = _dense_like(dy.to(xs.dtype), batch_first)
        dx = _new_token(b, n, h, dh, batch_first, xs)
        dcoeff = torch.empty_like(coeff)
        dbp = torch.empty((b * h, dh), dtype=torch.float32, device=xs.device)
        if mode == 'cheb':
            abi.cheb_filter_bwd(xs, g0, coeff, n_real, dys, dx, dcoeff, dbp, order, share, stream)
        else:
            abi.spec_filter_bwd(xs, g0, g1, coeff, n_real, dys, dx, dcoeff, dbp, order, share, stream)
        dbias = None
        if has_bias:
            dbias = torch.empty(dh, dtype=torch.float32, device=xs.device)
            abi.colsum(dbp, dbias, stream)
        return dx, dcoeff, dbias, None, None, None, None, None, None, None


import os as _os
# fp32 compute: above this many multiply-adds per product the C x C linear runs as library GEMMs.  Round 3 measured the
# LDS-tiled fp32 kernels of csrc/lin.hip at the BASELINE shape (2^29 MACs per product) against them: forward 15.7 vs
# 12.8 us, dX + dW (+ db + pending sums, one launch) 32.8 vs 25.8 us - the chip holds ~1.6 GHz under fp32 MFMA load, so
# the floor is 10.2 / 20.4 us and the library sits at 80 % of it; FETA_LIN_OWN_MAX_MACS=2147483648 selects ours (A/B).
# bf16 compute (bf16 storage legs) always takes the tiled kernels: 9.3 / 16.5 us, no cast launches.
LIN_OWN_GEMM_MAX_MACS = int(_os.environ.get('FETA_LIN_OWN_MAX_MACS', str(1 << 27)))
# bf16 compute (the bf16 storage legs): from this many rows H*B the three products run as LIBRARY bf16 GEMMs with fp32 output
# (torch.mm(..., out_dtype=float32) on bf16 copies of the operands - the same arithmetic as the tiled kernels of
# csrc/lin.hip: operands rounded to bf16, fp32 accumulation and result).  The tiled kernels read fp32 operands and win
# while the products are small (rows 512: 9.7 / 16.9 us against 33 / 52 us with the cast launches); at config 5's 4096
# rows they sit at 200 / 143 TFLOP/s (42.9 / 120.3 us) and the library, casts included, takes 35.5 / 60.3 us
# (tools/scratch/bf16_gemm_probe.py, eager).  Inside the captured step: 1024 rows 0.3683 vs 0.3651 ms (tiled kernels stay), 2048
# rows 0.547 vs 0.570 ms, 4096 rows (config 5) 1.08 vs 1.145 ms.
LIN_LIB_BF16_MIN_ROWS = int(_os.environ.get('FETA_LIN_LIB_BF16_MIN_ROWS', '2048'))


# The coefficient generator's kernels ride as trailing workgroups of a feed-forward launch while the batch leaves CUs idle
# there.  As a role they inherit the host kernel's two waves per SIMD; from this many (graph, head) blocks on, the stand-alone
# launches (eight waves per SIMD) are the faster form even with their launch cost: config 5 (4096 blocks) 1.040 -> 1.022 ms
# per step, the BASELINE batch (512 blocks) and config 4 (256 blocks of 128 nodes) the other way round.
COEFF_ROLE_MAX_BLOCKS = int(_os.environ.get('FETA_COEFF_ROLE_MAX', 2048))


class PendingSums:
    """Column sums (split-K partial buffer -> gradient) that one backward node of the filter stage leaves for a LATER
    node of the same backward pass which has a launch to carry them (trailing workgroups of feta_lin_bwd, or the
    reduction launch of the coefficient generator's backward): every launch of a captured step costs ~4.5 us whatever
    its size.  The encoder creates one per forward; a node that will flush arms it in forward iff autograd is going
    to run its backward, and only then do the nodes that run BEFORE it in backward defer: linear_cat -> filter
    (the filter output is linear_cat's operand) -> coefficient generator (pooled is the filter node's operand)."""

    def __init__(self):
        self.armed = False         # FilterFromPooledFn will run a backward (it takes what linear_cat leaves)
        self.coeff_armed = False   # FilterCoefficientsFn will (it takes whatever is left: the stage's last node)
        self.stack_armed = False   # the fused layer stack will, in the same backward pass (set by the encoder), and
        self.stack_done = False    # ... has not yet: its one reduction launch takes what the stage's last node leaves
        self.items = []
        self.owners = []
        self._callback_queued = False
        # forward direction: column sums of parameters that a launch of the layer stack carries in trailing workgroups
        # (s = colsum(gcn.weight) of the coefficient generator); whoever needs them first runs them if nobody did
        self.fwd_sums = []
        self.s = None
        # ... and the coefficient generator's kernels themselves: its forward shares the launch of the last layer's
        # feed-forward half (it needs that layer's attention matrix only), its backward kernel the first launch of the
        # stack's backward.  coeff_fwd_req = gcn bias (set by the encoder); coeff_fwd_out = (cj, pooled) once run;
        # coeff_bwd_req = the argument tuple of abi.coeff_bwd left by FilterCoefficientsFn.backward
        self.coeff_fwd_req = None
        self.coeff_fwd_out = None
        self.coeff_bwd_req = None

    @staticmethod
    def untouched(*params):
        """A gradient may be completed after its node returned only if nothing reads it before the flush: the
        parameter has no .grad to accumulate into (autograd then keeps the returned tensor itself) and no hooks."""
        # AccumulateGrad keeps the returned tensor without copying it only while grad mode is off (an ordinary
        # backward pass): under create_graph=True it accumulates a copy, and anomaly mode reads the buffer when the
        # node returns - both would see the un-reduced buffer
        if torch.is_grad_enabled() or torch.is_anomaly_enabled():
            return False
        for p in params:
            if p is None:
                continue
            if p.grad is not None or p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
                return False
        return True

    def add(self, partial, out, bcast=None, owners=()):
        """owners: [(parameter, the gradient tensor its node returns)] whose storage this sum completes later: checked
        when the backward pass ends (_finish_pass) - had autograd copied the returned tensor after all (a hook on the
        AccumulateGrad node, which untouched() cannot see), the finished values are copied over that copy."""
        # (a second tensor object on the same storage: autograd keeps a returned gradient without copying it only
        # if nobody else holds a reference to that tensor object)
        self.items.append((partial, out.detach(), None if bcast is None else bcast.detach()))
        for p, g in owners:
            if p is not None and g is not None:
                self.owners.append((p, g.detach()))
        if not self._callback_queued:
            # safety net: whatever no later node took (autograd pruned it from this pass) is reduced when the
            # backward pass ends
            torch.autograd.Variable._execution_engine.queue_callback(self._finish_pass)
            self._callback_queued = True

    def take(self):
        items, self.items = self.items, []
        return items

    def take_fwd(self):
        sums, self.fwd_sums = self.fwd_sums, []
        return sums

    def coeff_fwd_role(self, attn, n_real):
        """Called by the layer stack for the launch that follows the last attention: -> the argument tuple of the
        coefficient generator's forward (outputs allocated here), or None if nobody asked / s is not ready."""
        if self.coeff_fwd_req is None or self.s is None or self.fwd_sums or attn is None:
            return None
        if attn.shape[-1] > 64:      # (beyond the role's LDS tile budget the stand-alone launch is the faster form:
            return None              # csrc/feta_coeff.h - one 1024-thread workgroup per block, staged rows)
        if attn.shape[0] * attn.shape[1] > COEFF_ROLE_MAX_BLOCKS:
            return None
        gcn_bias, self.coeff_fwd_req = self.coeff_fwd_req, None
        b, h, n, _ = attn.shape
        cj = torch.empty((h * b, n), dtype=torch.float32, device=attn.device)
        pooled = torch.empty((h * b, self.s.shape[0]), dtype=torch.float32, device=attn.device)
        self.coeff_fwd_out = (cj, pooled)
        return (attn, n_real, self.s, gcn_bias.detach().contiguous(), cj, pooled)

    def take_coeff_bwd(self):
        req, self.coeff_bwd_req = self.coeff_bwd_req, None
        return req

    def _finish_pass(self):
        self._callback_queued = False
        req = self.take_coeff_bwd()
        items = self.take()
        if items:
            abi, stream = _lib.backend(items[0][0])
            if req is not None:
                cj, n_real, s, gb, dpooled, partial, b, n, h = req
                abi.coeff_bwd(cj, n_real, s, gb, dpooled, partial, None, None, b, n, h, stream)
            abi.colsum_multi(items, stream)
        owners, self.owners = self.owners, []
        for p, g in owners:
            # .grad is None: torch.autograd.grad() handed the tensor itself to the caller - nothing to repair
            if p.grad is not None and p.grad.data_ptr() != g.data_ptr():
                with torch.no_grad():
                    p.grad.copy_(g.view_as(p.grad))


USE_CAT_FOLD = _os.environ.get('FETA_CAT_FOLD', '1') != '0'


class CatFold:
    """linear_cat folded into the launch of the eigenbasis filter (feta_spec_filter_cat_fwd, ABI 10).  The encoder fills
    in what linear_cat would read - the stack output `y2` ([N, B, d]; with `tail`: the pre-norm rows whose last BatchNorm the
    consumer finalizes, fused_stack.StackTail), `w`, `bias` - and hands the object to filter_from_pooled; if the filter's
    forward takes the fused kernel it leaves linear_cat's output in `out`, and row_linear_cat[_bn] then launches nothing
    in forward.  0: FETA_CAT_FOLD=0 (A/B timing).
    Backward (feta_spec_filter_cat_bwd, ABI 11; FETA_CAT_FOLD_BWD=0: off): linear_cat's autograd node runs first - it
    launches nothing either, allocates its outputs (dxn, the BatchNorm-backward sums of the StackTail contract, one partial row
    per graph for dW_cat) and leaves them with dout in `bwd`; the filter's node, which autograd runs next (its incoming
    gradient is linear_cat's dx2, a placeholder nobody reads), fills them in its one launch.  Only when the partial rows
    have a reduction launch to ride in (PendingSums armed) - else linear_cat's own backward kernel runs as before."""

    def __init__(self, y2, w, bias, tail=None):
        self.y2, self.w, self.bias, self.tail = y2.detach(), w, bias, tail
        self.out = None
        self.y = None          # the filter's output (token view), saved by the fused forward for the fused backward
        self.bwd_ok = False    # the backward fold is possible for this shape
        self.bwd = None        # linear_cat's backward -> the filter's backward: dict(dout, dxn, partial, gs, prm)

    def defer_backward(self, dy, y2, x2, w, prm, pending, owners_of, has_bias):
        """Called by linear_cat's backward: -> (dx1, dx2, dW view, db view, gs) with nothing launched, or None if the
        backward fold does not apply."""
        if not (self.bwd_ok and USE_CAT_FOLD_BWD and pending is not None and self.bwd is None and self.y is not None):
            return None
        m, k1 = y2.shape
        ki, no = k1 + x2.shape[1], w.shape[0]
        nb = _lib.backend(y2)[0].spec_cat_bwd_rows(self.y.shape[0])       # one row per graph, or per workgroup of a walked batch
        dx1, dx2 = torch.empty_like(y2), torch.empty_like(x2)    # dx2: a placeholder (the filter's node ignores it)
        partial = torch.empty((nb, no * ki + no), dtype=torch.float32, device=y2.device)
        dwdb = torch.empty(no * ki + no, dtype=torch.float32, device=y2.device)
        gs = torch.empty((nb, 2, k1), dtype=torch.float32, device=y2.device) if prm is not None else None
        self.bwd = dict(dout=dy, dxn=dx1, partial=partial, gs=gs, prm=prm)
        pending.add(partial, dwdb, owners=[(owners_of[0], dwdb[:no * ki].view(no, ki)),
                                           (owners_of[1], dwdb[no * ki:] if has_bias else None)])
        return dx1, dx2, dwdb[:no * ki].view(no, ki), (dwdb[no * ki:] if has_bias else None), gs


USE_CAT_FOLD_BWD = _os.environ.get('FETA_CAT_FOLD_BWD', '1') != '0'


class FilterFromPooledFn(torch.autograd.Function):
    """``self.linear`` of the coefficient generator (transformer/models.py:284) and the dynamic filter
    (transformer/models.py:346-360, transformer/ChebNetDynamic.py:132-189) as ONE autograd node:
    coeff = pooled W_lin^T + b_lin (rocBLAS), y = filter(x; coeff).  -> (y, coeff [H*B, C]).
    Same kernels as DenseLinearFn + _FilterFn; what the merge buys is one reduction launch for both bias
    gradients (colsum of the per-block filter-bias partials and colsum of dcoeff = the gradient of b_lin become
    available together) instead of two, and no dcoeff hand-over through autograd."""

    @staticmethod
    def forward(ctx, x, pooled, lin_w, lin_b, bias, n_real, g0, g1, mode, order, share, batch_first, pending,
                gemm_bf16=False, cat=None):
        abi, stream = _lib.backend(x, pooled)
        ctx.gemm_bf16 = bool(gemm_bf16)
        b, n, h, dh = x.shape
        xs = _dense_like(x, batch_first)
        # fp32 master precision (also what a regulariser sees)
        pooled, lin_w = pooled.contiguous(), lin_w.contiguous()
        r_, k_, n_ = pooled.shape[0], pooled.shape[1], lin_w.shape[0]
        # csrc/lin.hip (one 16 x 16 tile per wave straight from L2, gradient products and column sums in one launch)
        # where the products are launch-bound; the library GEMM where they are compute-bound (C = 1024 at the
        # BASELINE shape: 1 GFLOP each, ~12 us at half the fp32 matrix peak; lin.hip is L2-bound there, 31 us)
        ctx.lib_bf16 = bool(ctx.gemm_bf16 and r_ >= LIN_LIB_BF16_MIN_ROWS and pooled.is_cuda)
        ctx.own_gemm = (not ctx.lib_bf16 and abi.lin_supported(r_, k_, n_)
                        and (r_ * k_ * n_ <= LIN_OWN_GEMM_MAX_MACS or ctx.gemm_bf16))
        ctx.lp = None
        if ctx.own_gemm:
            coeff = torch.empty((pooled.shape[0], lin_w.shape[0]), dtype=torch.float32, device=pooled.device)
            abi.lin_fwd(pooled, lin_w, lin_b, coeff, stream, bf16=ctx.gemm_bf16)
        elif ctx.lib_bf16:
            p16, w16 = pooled.to(torch.bfloat16), lin_w.to(torch.bfloat16)
            coeff = torch.mm(p16, w16.t(), out_dtype=torch.float32)
            if lin_b is not None:
                coeff += lin_b
            ctx.lp = (p16, w16)       # (the backward's operands: no second pair of cast launches)
        else:
            with _lib.tuned_gemm():
                coeff = torch.addmm(lin_b, pooled, lin_w.t())
        ctx.pending = pending
        # pooled's gradient flows into FilterCoefficientsFn: if that node flushes, it runs after this one
        ctx.defer = pending is not None and pending.coeff_armed and ctx.needs_input_grad[1] and not ctx.own_gemm
        ctx.params = (lin_b, bias)
        if pending is not None and any(ctx.needs_input_grad):
            pending.armed = True
        # bf16 storage path: the per-block weights the filter kernel reads are bf16 copies of them
        cw = coeff if x.dtype == torch.float32 else coeff.to(x.dtype)
        y = _new_token(b, n, h, dh, batch_first, x)
        fold = (cat is not None and USE_CAT_FOLD and mode == 'spec' and x.dtype == torch.float32 and not batch_first
                and cat.w.shape == (h * dh, 2 * h * dh) and abi.spec_cat_supported(n, h, dh, order, g0.shape[2], share))
        if fold:
            # linear_cat rides in this launch (CatFold): out = [xn | y] W_cat^T + b
            out = _new_token(b, n, h, dh, batch_first, x)
            y2v = cat.y2.detach().view(n, b, h, dh).permute(1, 0, 2, 3)
            kw = {}
            if cat.tail is not None:
                t, nm = cat.tail, cat.tail.norm
                kw = dict(y2_stats=t.st2, Gx=t.G2, gamma=t.gamma, beta=t.beta, bn_out=t.prm2, rmean=nm.running_mean,
                          rvar=nm.running_var, nbt=nm.num_batches_tracked, momentum=float(nm.momentum), eps=float(nm.eps))
            abi.spec_filter_cat_fwd(xs, g0, g1, cw, bias, n_real, y, order, share, stream, y2=y2v, w_cat=cat.w.detach().contiguous(),
                                    b_cat=None if cat.bias is None else cat.bias.detach(), out=out, **kw)
            cat.out = out.permute(1, 0, 2, 3).reshape(n * b, h * dh)      # (a view: [N, B, d] rows)
            cat.y = y.detach()      # (another tensor object: the returned y gets this node as grad_fn - kept in ctx.cat it would be a cycle)
            cat.bwd_ok = bool(abi.spec_cat_bwd_supported(n, h, dh, order, g0.shape[2], share))
        elif mode == 'cheb':
            abi.cheb_filter_fwd(xs, g0, cw, bias, n_real, y, order, share, stream)
        else:
            abi.spec_filter_fwd(xs, g0, g1, cw, bias, n_real, y, order, share, stream)
        ctx.save_for_backward(xs, cw, pooled, lin_w, n_real, g0, g1)
        ctx.cfg = (mode, order, share, batch_first, bias is not None)
        ctx.cat = cat if fold else None
        ctx.set_materialize_grads(False)
        return y, coeff

    @staticmethod
    def backward(ctx, dy, dcoeff_ext):
        xs, coeff, pooled, lin_w, n_real, g0, g1 = ctx.saved_tensors
        mode, order, share, batch_first, has_bias = ctx.cfg
        abi, stream = _lib.backend(xs)
        b, n, h, dh = xs.shape
        waiting = ctx.pending.take() if ctx.pending is not None else []
        sums = []           # (in, out) column sums still to run
        if dy is None:      # only the coefficients were used downstream
            dx, dbias = None, None
            dcoeff = dcoeff_ext.contiguous()
            db_lin = torch.empty(dcoeff.shape[1], dtype=torch.float32, device=xs.device)
        else:
            st = None
            if ctx.cat is not None and ctx.cat.bwd is not None:
                st, ctx.cat.bwd = ctx.cat.bwd, None
            dx = _new_token(b, n, h, dh, batch_first, xs)
            dcoeff = torch.empty_like(coeff)
            dbp = torch.empty((b * h, dh), dtype=torch.float32, device=xs.device)
            if st is not None:
                # linear_cat's backward rides in this launch (CatFold): dy is its placeholder - the kernel starts from dout
                cat = ctx.cat
                tv = lambda t: t.view(n, b, h, dh).permute(1, 0, 2, 3)
                abi.spec_filter_cat_bwd(xs, g0, g1, coeff, n_real, cat.y, dx, dcoeff, dbp, order, share, stream,
                                        dout=tv(st['dout']), y2=tv(cat.y2.detach()), w_cat=cat.w.detach().contiguous(),
                                        dxn=tv(st['dxn']), partial=st['partial'], y2_bn=st['prm'], gs=st['gs'])
            else:
                dys = _dense_like(dy.to(xs.dtype), batch_first)
            if st is not None:
                pass
            elif mode == 'cheb':
                abi.cheb_filter_bwd(xs, g0, coeff, n_real, dys, dx, dcoeff, dbp, order, share, stream)
            else:
                abi.spec_filter_bwd(xs, g0, g1, coeff, n_real, dys, dx, dcoeff, dbp, order, share, stream)
            if dcoeff.dtype != torch.float32:
                dcoeff = dcoeff.float()      # the C x C linear's gradients accumulate in fp32
            if dcoeff_ext is not None:   # a regulariser on the coefficients (transformer/models.py:554-584)
                dcoeff += dcoeff_ext
            both = torch.empty(dh + dcoeff.shape[1], dtype=torch.float32, device=xs.device)
            dbias, db_lin = both[:dh], both[dh:]
            sums.append((dbp, dbias))
        sums += waiting
        if ctx.own_gemm:
            # dpooled, dW_lin, db_lin and every pending column sum in ONE launch (csrc/lin.hip)
            dpooled = torch.empty_like(pooled) if ctx.needs_input_grad[1] else None
            dw_lin = torch.empty_like(lin_w)
            abi.lin_bwd(pooled, lin_w, dcoeff, dpooled, dw_lin, db_lin, stream, pairs=sums, bf16=ctx.gemm_bf16)
        else:
            sums.append((dcoeff, db_lin))
            if ctx.defer and PendingSums.untouched(*ctx.params):
                for pr in sums:
                    ctx.pending.add(*pr)
                # (dbias is None when only the coefficients were used downstream - a regulariser on them, dy is None)
                ctx.pending.owners += [(p_, g_.detach()) for p_, g_ in ((ctx.params[0], db_lin), (ctx.params[1], dbias))
                                       if p_ is not None and g_ is not None]
            else:
                abi.colsum_multi(sums, stream)
            if ctx.lib_bf16:
                p16, w16 = ctx.lp
                d16 = dcoeff.to(torch.bfloat16)
                dpooled = torch.mm(d16, w16, out_dtype=torch.float32)
                dw_lin = torch.mm(d16.t(), p16, out_dtype=torch.float32)
            else:
                with _lib.tuned_gemm():
                    dpooled, dw_lin = dcoeff.mm(lin_w), dcoeff.t().mm(pooled)
        if not has_bias:
            dbias = None
        return (dx, dpooled, dw_lin, db_lin, dbias) + (None,) * 10


def filter_from_pooled(x, pooled, lin_w, lin_b, bias, n_real, graph, mode, order, heads_share_graph=False,
                       batch_first=False, pending=None, gemm_bf16=False, cat=None):
    """x [B,N,H,dh] view, pooled [H*B, C] -> (y, coeff [H*B, C]); graph = (lhat,) | (u, lam)."""
    g0 = graph[0].contiguous()
    g1 = graph[1].contiguous() if len(graph) > 1 else None
    if x.dtype != torch.float32:
        if mode != 'spec':
            raise NotImplementedError("the bf16 storage path runs the eigenbasis filter (filter_mode='spectral')")
        g0 = g0.to(x.dtype)      # U travels as bf16; lambda and t_k(lambda) stay fp32
    return FilterFromPooledFn.apply(x, pooled, lin_w, lin_b, bias, n_real, g0, g1, mode, order,
                                    bool(heads_share_graph), batch_first, pending, bool(gemm_bf16), cat)


class RowLinearFn(torch.autograd.Function):
    """y = relu?(x W^T + b) * rowscale? + residual?  on [M, KI] rows, plus (optionally) the
    per-block partial BatchNorm statistics of y.  C ABI: feta_rowlin_fwd / feta_rowlin_bwd."""

    @staticmethod
    def forward(ctx, x, w, bias, rowscale, residual, relu, want_stats, stats_shift=None):
        abi, stream = _lib.backend(x, w)
        ctx.set_materialize_grads(False)   # no zero tensor for the non-differentiable stats output
        assert not (relu and residual is not None), 'relu mask is taken from the saved output'
        x = x.contiguous()
        w = w.contiguous()
        m, no = x.shape[0], w.shape[0]
        y = torch.empty((m, no), dtype=torch.float32, device=x.device)
        stats = None
        if want_stats:
            # shifted partial sums + the shift row (csrc/feta_rowops.h): relative to the consumer BatchNorm's running mean
            stats = torch.empty((abi.rowlin_blocks(m) + 1, 2, no), dtype=torch.float32, device=x.device)
        res = None if residual is None else residual.contiguous()
        if want_stats and stats_shift is not None:
            d = abi.rowlin_ex(m, x.shape[1], no, relu=relu, x=x, w=w, bias=bias, rowscale=rowscale, residual=res, y=y,
                              stats=stats, stats_shift=stats_shift)
            abi.rowlin_fwd_ex(d, stream)
        else:
            abi.rowlin_fwd(x, w, bias, rowscale, res, y, stats, relu, stream)
        ctx.save_for_backward(x, w, rowscale, y if relu else None)
        ctx.cfg = (bias is not None, residual is not None)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        x, w, rowscale, ysaved = ctx.saved_tensors
        if dy is None:
            return (None,) * 8
        has_bias, has_res = ctx.cfg
        abi, stream = _lib.backend(x)
        m, ki = x.shape
        no = w.shape[0]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        partial = torch.empty((abi.rowlin_chunks(m), no * ki + no), dtype=torch.float32, device=x.device)
        dwdb = torch.empty(no * ki + no, dtype=torch.float32, device=x.device)
        abi.rowlin_bwd(x, w, dy, rowscale, ysaved, dx, partial, dwdb, stream)
        dw = dwdb[:no * ki].view(no, ki)
        db = dwdb[no * ki:] if has_bias else None
        return dx, dw, db, None, (dy if has_res else None), None, None, None


class RowLinearCatFn(torch.autograd.Function):
    """y = [x1 | x2] W^T + b without materialising the concatenation (linear_cat,
    transformer/models.py:223-224); backward writes dx1 and dx2 directly."""

    @staticmethod
    def forward(ctx, x1, x2, w, bias, pending=None, done=None, fold=None):
        abi, stream = _lib.backend(x1, x2, w)
        ctx.defer = pending if (pending is not None and pending.armed and x2.requires_grad) else None
        ctx.fold = fold if done is not None else None
        ctx.params = (w, bias)
        x1, x2, w = x1.contiguous(), x2.contiguous(), w.contiguous()
        m, k1 = x1.shape
        ki, no = k1 + x2.shape[1], w.shape[0]
        if done is not None:     # the filter's launch computed it (CatFold)
            y = done
        else:
            y = torch.empty((m, no), dtype=torch.float32, device=x1.device)
            d = abi.rowlin_ex(m, ki, no, x=x1, x2=x2, x_split=k1, w=w, bias=bias, y=y)
            abi.rowlin_fwd_ex(d, stream)
        ctx.save_for_backward(x1, x2, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, w = ctx.saved_tensors
        abi, stream = _lib.backend(x1)
        m, k1 = x1.shape
        ki, no = k1 + x2.shape[1], w.shape[0]
        dy = dy.contiguous()
        if ctx.defer is not None and not PendingSums.untouched(*ctx.params):
            ctx.defer = None
        if ctx.fold is not None and ctx.defer is not None:
            r = ctx.fold.defer_backward(dy, x1, x2, w, None, ctx.defer, ctx.params, ctx.has_bias)
            if r is not None:
                return r[0], r[1], r[2], r[3], None, None, None
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        partial = torch.empty((abi.rowlin_chunks(m), no * ki + no), dtype=torch.float32, device=x1.device)
        dwdb = torch.empty(no * ki + no, dtype=torch.float32, device=x1.device)
        d = abi.rowlin_ex(m, ki, no, x=x1, x2=x2, x_split=k1, w=w, dy=dy, dx=dx1, dx2=dx2, partial=partial,
                          partial_ld=(no * ki + no if ctx.defer is not None else 0))
        if ctx.defer is not None:     # the filter's backward reduces the partials inside its own launch
            abi.rowlin_bwd_ex(d, None, stream)
            ctx.defer.add(partial, dwdb, owners=[(ctx.params[0], dwdb[:no * ki].view(no, ki)),
                                                 (ctx.params[1], dwdb[no * ki:] if ctx.has_bias else None)])
        else:
            abi.rowlin_bwd_ex(d, dwdb, stream)
        return dx1, dx2, dwdb[:no * ki].view(no, ki), (dwdb[no * ki:] if ctx.has_bias else None), None, None, None


class RowLinearCatBNFn(torch.autograd.Function):
    """linear_cat (transformer/models.py:223-224) over [BN(y2) | x2] where BN is the last BatchNorm of the fused
    layer stack, never materialised: forward finalizes the statistics the stack left in `tail` and normalises inside
    the operand loads; backward returns, for y2, the gradient w.r.t. the NORMALISED tensor and leaves the BatchNorm
    backward partial sums in tail.gs (fused_stack.StackTail contract)."""

    @staticmethod
    def forward(ctx, y2, x2, w, bias, tail, pending=None, done=None, fold=None):
        abi, stream = _lib.backend(y2, x2, w)
        ctx.defer = pending if (pending is not None and pending.armed and x2.requires_grad) else None
        ctx.fold = fold if done is not None else None
        ctx.params = (w, bias)
        y2, x2, w = y2.contiguous(), x2.contiguous(), w.contiguous()
        m, k1 = y2.shape
        ki, no = k1 + x2.shape[1], w.shape[0]
        if done is not None:     # the filter's launch computed it and finalized the BatchNorm (CatFold: tail.prm2 is published)
            out = done
        else:
            out = torch.empty((m, no), dtype=torch.float32, device=y2.device)
            nm = tail.norm
            d = abi.rowlin_ex(m, ki, no, x=y2, x2=x2, x_split=k1, w=w, bias=bias, y=out, x_stats=tail.st2, Gx=tail.G2,
                              x_gamma=tail.gamma, x_beta=tail.beta, x_bn_out=tail.prm2, x_rmean=nm.running_mean,
                              x_rvar=nm.running_var, x_nbt=nm.num_batches_tracked, momentum=float(nm.momentum),
                              eps=float(nm.eps))
            abi.rowlin_fwd_ex(d, stream)
        ctx.save_for_backward(y2, x2, w, tail.prm2)
        ctx.tail = tail
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        y2, x2, w, prm2 = ctx.saved_tensors
        abi, stream = _lib.backend(y2)
        m, k1 = y2.shape
        ki, no = k1 + x2.shape[1], w.shape[0]
        dy = dy.contiguous()
        if ctx.defer is not None and not PendingSums.untouched(*ctx.params):
            ctx.defer = None
        if ctx.fold is not None and ctx.defer is not None:
            r = ctx.fold.defer_backward(dy, y2, x2, w, prm2, ctx.defer, ctx.params, ctx.has_bias)
            if r is not None:
                ctx.tail.gs = r[4]      # (one row per graph, filled by the filter's launch: the stack's node runs behind it)
                return r[0], r[1], r[2], r[3], None, None, None, None
        dx1, dx2 = torch.empty_like(y2), torch.empty_like(x2)
        partial = torch.empty((abi.rowlin_chunks(m), no * ki + no), dtype=torch.float32, device=y2.device)
        dwdb = torch.empty(no * ki + no, dtype=torch.float32, device=y2.device)
        gs = torch.empty((abi.rowlin_blocks(m), 2, k1), dtype=torch.float32, device=y2.device)
        d = abi.rowlin_ex(m, ki, no, x=y2, x_bn=prm2, x2=x2, x_split=k1, w=w, dy=dy, dx=dx1, dx2=dx2, partial=partial,
                          partial_ld=(no * ki + no if ctx.defer is not None else 0), sum_y=y2, sum_bn=prm2, sum_out=gs)
        if ctx.defer is not None:     # the filter's backward reduces the partials inside its own launch
            abi.rowlin_bwd_ex(d, None, stream)
            ctx.defer.add(partial, dwdb, owners=[(ctx.params[0], dwdb[:no * ki].view(no, ki)),
                                                 (ctx.params[1], dwdb[no * ki:] if ctx.has_bias else None)])
        else:
            abi.rowlin_bwd_ex(d, dwdb, stream)
        ctx.tail.gs = gs
        return dx1, dx2, dwdb[:no * ki].view(no, ki), (dwdb[no * ki:] if ctx.has_bias else None), None, None, None, None


def row_linear_cat_bn(y2, x2, w, bias, tail, pending=None, done=None, fold=None):
    return RowLinearCatBNFn.apply(y2, x2, w, bias, tail, pending, done, fold)


class BatchNormTrainFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over the rows of y [M, D] from per-block partial statistics.
    C ABI: feta_bn_stats (when the producer did not emit them), feta_bn_apply_fwd, feta_bn_bwd."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, momentum, eps, nbt=None):
        abi, stream = _lib.backend(y)
        y = y.contiguous()
        m, d = y.shape
        if stats is None:
            stats = torch.empty((abi.rowlin_blocks(m) + 1, 2, d), dtype=torch.float32, device=y.device)
            abi.bn_stats(y, stats, stream, shift=running_mean)
        out = torch.empty_like(y)
        mean_rstd = torch.empty((2, d), dtype=torch.float32, device=y.device)
        abi.bn_apply_fwd(y, stats, gamma, beta, out, mean_rstd, running_mean, running_var,
                         float(momentum), float(eps), stream, nbt=nbt)
        ctx.save_for_backward(y, mean_rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, mean_rstd, gamma = ctx.saved_tensors
        abi, stream = _lib.backend(y)
        m, d = y.shape
        dout = dout.contiguous()
        partial = torch.empty((abi.rowlin_blocks(m), 2, d), dtype=torch.float32, device=y.device)
        dy = torch.empty_like(y)
        dgamma = torch.empty(d, dtype=torch.float32, device=y.device)
        dbeta = torch.empty(d, dtype=torch.float32, device=y.device)
        abi.bn_bwd(y, dout, mean_rstd, gamma, partial, dy, dgamma, dbeta, stream)
        return dy, None, dgamma, dbeta, None, None, None, None, None


ROWLIN_DIMS = (16, 32, 64, 128, 192, 256)


def row_linear_supported(ki, no, need_backward=True):
    return ki in ROWLIN_DIMS and (no in ROWLIN_DIMS if need_backward else no % 16 == 0)


def row_linear(x, w, bias=None, rowscale=None, residual=None, relu=False, want_stats=False, stats_shift=None):
    """x [M, KI] -> (y [M, NO], stats [G + 1, 2, NO] or None).  stats_shift [NO]: the running mean of the BatchNorm that
    will consume the statistics - they are sums of (y - shift) (csrc/feta_rowops.h)."""
    return RowLinearFn.apply(x, w, bias, rowscale, residual, relu, want_stats, stats_shift)


def row_linear_cat(x1, x2, w, bias=None, pending=None, done=None, fold=None):
    """[x1 | x2] W^T + b on [M, .] rows; needs x1.shape[1] % 16 == 0 and supported total dims."""
    return RowLinearCatFn.apply(x1, x2, w, bias, pending, done, fold)


def batch_norm_train(y, stats, gamma, beta, running_mean, running_var, momentum, eps, num_batches_tracked=None):
    return BatchNormTrainFn.apply(y, stats, gamma, beta, running_mean, running_var, momentum, eps,
                                  num_batches_tracked)


class LayerNormRowsFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of [M, D] rows (feta_layernorm_fwd/bwd): norm1 / norm2 of
    the encoder layer when batch_norm=False."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps):
        abi, stream = _lib.backend(y, gamma)
        y = y.contiguous()
        m, d = y.shape
        out = torch.empty_like(y)
        stats = torch.empty((m, 2), dtype=torch.float32, device=y.device)
        abi.layernorm_fwd(y, gamma.contiguous(), beta.contiguous(), float(eps), out, stats, stream)
        ctx.save_for_backward(y, stats, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, stats, gamma = ctx.saved_tensors
        abi, stream = _lib.backend(dout)
        m, d = y.shape
        dy = torch.empty_like(y)
        partial = torch.empty((abi.layernorm_blocks(m), 2, d), dtype=torch.float32, device=y.device)
        dgdb = torch.empty((2, d), dtype=torch.float32, device=y.device)
        abi.layernorm_bwd(dout.contiguous(), y, stats, gamma.contiguous(), dy, partial, dgdb, stream)
        return dy, dgdb[0], dgdb[1], None


def layer_norm_rows_supported(d):
    return d % 4 == 0 and 4 <= d <= 256


def layer_norm_rows(y, gamma, beta, eps):
    return LayerNormRowsFn.apply(y, gamma, beta, eps)


class DropoutState:
    """(seed, offset) of the attention-probability dropout masks: every masked forward takes the next offset, so
    masks differ between layers and steps and are reproducible from the seed (the kernels derive the mask from
    (seed, offset, b, h, query, key): include/feta_hip.h, feta_attn_fwd_drop).

    Host mode (default): (seed, offset) are Python ints handed to the kernels as arguments.
    Device mode (begin_device_mode; used by train.GraphedTrainStep): the key lives in an int64[2] DEVICE tensor which
    the kernels read when they RUN (feta_attn_*_drop_dev); call k of a step uses offset state[1] + k and the step
    itself advances state[1] by its number of masked calls (end_step, a captured in-place add) - a hipGraph replay
    therefore draws the masks the eager loop would have drawn at that step."""
    seed = None
    offset = 0
    _dev = None      # int64 [seed, offset] on the device of the captured steps (device mode), else None
    _keys = {}       # device -> THE key tensor of that device.  Captured graphs hold its raw address (the kernels read
    #                  it at run time, the captured end_step add writes it), so it is created once per device and never
    #                  replaced or freed: a second GraphedTrainStep (one per padded-size bucket) shares it - and with it
    #                  ONE offset stream, so buckets never replay each other's (seed, offset) pairs
    _calls = 0       # masked calls so far in the current step (device mode)

    @classmethod
    def manual_seed(cls, seed):
        cls.seed, cls.offset = int(seed) & (2 ** 63 - 1), 0
        cls._sync_device()

    @classmethod
    def _sync_device(cls):
        """host (seed, offset) -> every key tensor that exists (graphs captured on any device follow a re-seed)"""
        for key in cls._keys.values():
            key.copy_(torch.tensor([cls.seed, cls.offset], dtype=torch.int64))
        cls._calls = 0

    @classmethod
    def device_mode(cls):
        return cls._dev is not None

    @classmethod
    def begin_device_mode(cls, device):
        if cls.seed is None:
            cls.manual_seed(torch.initial_seed())
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        key = cls._keys.get(device)
        if key is None:
            key = cls._keys[device] = torch.zeros(2, dtype=torch.int64, device=device)
        key.copy_(torch.tensor([cls.seed, cls.offset], dtype=torch.int64))
        cls._dev = key
        cls._calls = 0

    @classmethod
    def end_device_mode(cls):
        """Back to host-held keys.  The key tensors stay alive (graphs captured in device mode may still be replayed:
        they keep reading and advancing their device's key)."""
        cls._dev = None
        cls._calls = 0

    @classmethod
    def next(cls):
        """-> (seed, offset) as ints, or (state tensor, offset_add) in device mode"""
        if cls._dev is not None:
            cls._calls += 1
            return cls._dev, cls._calls
        if cls.seed is None:
            cls.manual_seed(torch.initial_seed())
        cls.offset += 1
        return cls.seed, cls.offset

    @classmethod
    def end_step(cls):
        """Device mode: advance the device offset by the masked calls of this step (an in-place add on the current
        stream: part of the captured step); the host mirror follows.  -> number of calls"""
        n, cls._calls = cls._calls, 0
        if cls._dev is not None and n:
            cls._dev[1:2].add_(n)
            cls.offset += n
        return n

    @classmethod
    def replayed(cls, n):
        """A captured step that made n masked calls was replayed: the host mirror follows the device offset."""
        cls.offset += n

    @classmethod
    def snapshot(cls):
        return cls.seed, cls.offset

    @classmethod
    def restore(cls, snap):
        cls.seed, cls.offset = snap
        if cls.seed is not None:
            cls._sync_device()


def attention_core(qkv, pe, n_real, num_heads, need_attn=True, tie_qk=False, batch_first=False, dropout_p=0.0,
                   stab='rowmax'):
    """dropout_p > 0: attention-probability dropout with a mask regenerated in backward (no mask tensor).  The key
    (seed, offset) is a pair of host values, or - DropoutState in device mode - a device tensor the kernels read when
    they run, which is what makes the op capturable."""
    drop = None
    if dropout_p > 0.0:
        if qkv.is_cuda and torch.cuda.is_current_stream_capturing() and not DropoutState.device_mode():
            raise RuntimeError('attention dropout inside a captured hipGraph would replay one fixed mask: capture with '
                               'the key on the device (functional.DropoutState.begin_device_mode; '
                               'train.GraphedTrainStep does)')
        drop = (float(dropout_p),) + DropoutState.next()
    if stab not in ('rowmax', 'clamp5'):
        raise ValueError("stab must be 'rowmax' or 'clamp5'")
    if stab == 'clamp5' and drop is not None:
        raise NotImplementedError('stab=clamp5 with attention dropout')
    return AttentionCoreFn.apply(qkv, pe, n_real, num_heads, need_attn, tie_qk, batch_first, drop, stab == 'clamp5')


def filter_coefficients(attn, n_real, gcn_weight, gcn_bias, pending=None):
    return FilterCoefficientsFn.apply(attn, n_real, gcn_weight, gcn_bias, pending)


def cheb_filter(x, lhat, coeff, bias, n_real, order, heads_share_graph=False, batch_first=False):
    return _FilterFn.apply(x, coeff, bias, n_real, lhat.contiguous(), None, 'cheb', order,
                           bool(heads_share_graph), batch_first)


def spec_filter(x, u, lam, coeff, bias, n_real, order, heads_share_graph=False, batch_first=False):
    return _FilterFn.apply(x, coeff, bias, n_real, u.contiguous(), lam.contiguous(), 'spec', order,
                           bool(heads_share_graph), batch_first)


def lhat_from_edges(edge_index, node_graph, node_off, num_graphs, n_pad):
    """Dense scaled Laplacian [B,N,N] of every graph of the batch (feta_lhat_from_edges)."""
    abi, stream = _lib.backend(edge_index, node_graph)
    dev = node_graph.device
    lhat = torch.zeros((num_graphs, n_pad, n_pad), dtype=torch.float32, device=dev)
    deg = torch.zeros(node_graph.shape[0], dtype=torch.float32, device=dev)
    abi.lhat_from_edges(edge_index.contiguous(), node_graph.contiguous(), node_off.contiguous(),
                        deg, lhat, stream)
    return lhat


def eigh_sym(a, n_real, shift, k=None, max_sweeps=0, tol=0.0, return_sweeps=False):
    """Batched symmetric eigendecomposition on the device (feta_eigh_sym): a [B,N,N] (lower triangle
    read, a + shift*I positive definite), n_real [B] int32 -> u [B,N,K] (ascending, zero-padded),
    lam [B,K].  Replaces the per-graph host eig of transformer/position_encoding.py:127-161."""
    abi, stream = _lib.backend(a)
    b, n, _ = a.shape
    k = n if k is None else int(k)
    if not abi.eigh_sym_supported(n):
        raise ValueError('eigh_sym: N = %d is beyond the kernels (N <= 256); decompose on the host' % n)
    u = torch.empty((b, n, k), dtype=torch.float32, device=a.device)
    lam = torch.empty((b, k), dtype=torch.float32, device=a.device)
    sweeps = torch.empty((b,), dtype=torch.int32, device=a.device) if return_sweeps else None
    abi.eigh_sym(a.contiguous(), n_real.contiguous(), float(shift), u, lam, sweeps, int(max_sweeps),
                 float(tol), stream)
    return (u, lam, sweeps) if return_sweeps else (u, lam)


SPECTRAL_MODES = {'diffusion': 0, 'pstep': 1}


def spectral_kernel(u, lam, n_real, kind='diffusion', beta=1.0, p=1, lam_offset=0.0, zero_diag=False):
    """out [B,N,N] = U f(lam + lam_offset) U^T on the real block (feta_spectral_kernel):
    'diffusion' f = exp(-beta x) (transformer/position_encoding.py:65-72), 'pstep' f = (1 - beta x)^p
    (:83-93)."""
    if kind not in SPECTRAL_MODES:
        raise ValueError('unknown spectral kernel %r' % (kind,))
    abi, stream = _lib.backend(u, lam)
    b, n, _ = u.shape
    out = torch.empty((b, n, n), dtype=torch.float32, device=u.device)
    abi.spectral_kernel(u.contiguous(), lam.contiguous(), n_real.contiguous(), SPECTRAL_MODES[kind],
                        float(beta), int(p), float(lam_offset), bool(zero_diag), out, stream)
    return out
