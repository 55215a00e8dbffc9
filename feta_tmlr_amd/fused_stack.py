"""The whole stack of DiffTransformerEncoderLayers as ONE autograd node with a hand-scheduled forward
and backward over the C ABI: FusedEncoderStackFn (BatchNorm layers, described here) and
FusedLayerNormStackFn (LayerNorm layers, described at the class).

Per layer, forward (5 launches):
    F1  qkv  = BN2_prev(y2_prev) W_in^T                       feta_rowlin_fwd_ex (finalizes BN2_prev)
    F2  concat, attn = attention core                         feta_attn_fwd
    F3  y1   = X0 + degree * (concat W_o^T + b_o), stats1     feta_rowlin_fwd_ex (residual through BN2_prev)
    F4  h    = relu(BN1(y1) W_1^T + b_1)                      feta_rowlin_fwd_ex (finalizes BN1)
    F5  y2   = BN1(y1) + h W_2^T + b_2, stats2                feta_rowlin_fwd_ex (residual through BN1)
and one feta_bn_apply_fwd_prm at the end of the stack.  BatchNorm never runs as a pass of its own:
statistics come out of the producer's epilogue, the apply happens inside the consumers' operand
loads, and normalised activations are never materialised between layers.
Backward (6 launches + the weight-gradient reductions per layer): the BatchNorm backward of the
incoming gradient is applied inside the gradient loads of feta_rowlin_bwd_ex, the residual
gradients are added in its dX epilogue, and the partial sums the next BatchNorm backward needs are
emitted there too - no stand-alone BatchNorm, add or fill kernels.

Reference semantics: DiffTransformerEncoderLayer.forward as reconstructed in
feta_tmlr_amd/transformer/layers.py (contract transformer/models.py:166-167).
"""
import os
import weakref

import torch

from . import _lib


def _views(t, l0, l1, heads, dh):
    v5 = t.view(l0, l1, 3, heads, dh)
    return [v5[:, :, i].permute(1, 0, 2, 3) for i in range(3)]


PER_LAYER = 12  # tensors per layer in the flat parameter list
# in_proj + attention + out_proj of a layer as one launch where the shape allows (csrc/block.hip);
# FETA_ATTN_BLOCK=0 keeps the three-launch sequence (A/B timing, fallback for other shapes)
USE_ATTN_BLOCK = os.environ.get('FETA_ATTN_BLOCK', '1') != '0'
USE_FFN_FUSED = os.environ.get('FETA_FFN_FUSED', '1') != '0'
# graphs beyond the one-launch block (64 < N <= 256, config 4): attention core + out_proj + degree + residual + statistics
# as one launch behind the in_proj launch (csrc/attnout.hip); 0: feta_attn_fwd -> feta_rowlin_fwd_ex
USE_ATTN_OUT = os.environ.get('FETA_ATTN_OUT', '1') != '0'
# LayerNorm of the feed-forward kernel's OUTPUT rows in its epilogue (feta_ffn.y_ln_out) where nobody downstream applies it
# on load; 0: feta_layernorm_fwd launches (A/B timing)
USE_LN_EPILOGUE = os.environ.get('FETA_LN_EPILOGUE', '1') != '0'
# backward of the FFN half (linear2 + linear1) in one launch (csrc/ffn_bwd.hip); 0: two feta_rowlin_bwd_ex launches
USE_FFN_BWD = os.environ.get('FETA_FFN_BWD', '1') != '0'
# backward of the attention sub-block (out_proj + attention + in_proj) in one launch per layer (csrc/block_bwd.hip,
# one workgroup per graph, up to 256 graphs); 0: three launches
USE_ATTN_BLOCK_BWD = os.environ.get('FETA_ATTN_BLOCK_BWD', '1') != '0'
# two workgroups per graph (one per pair of heads) where the consumer of dx is the fused FFN backward, which adds the
# two parts on load (feta_attn_block_grad.dx_b); 0: one workgroup per graph everywhere (A/B timing)
USE_ATTN_BLOCK_SPLIT = os.environ.get('FETA_ATTN_BLOCK_SPLIT', '1') != '0'
# more graphs than workgroups (B > 256): the fused attention-block backward walks several graphs per workgroup.  Until the
# lane id was laundered once per graph (csrc/block_bwd.hip: everything a lane derives from its id is invariant in the
# graph loop and was hoisted - up to 676 B of scratch per lane) that instantiation was slower than the three-launch form
# on fp32 (molhiv B = 1024, N_pad = 64: 533 k vs 649 k graphs/s); spill-free it wins: 734 k vs 663 k fp32, 901 k (777 k)
# bf16, ZINC B = 512 736 k vs 699 k.  0: three launches for fp32 stacks beyond 256 graphs (A/B timing)
USE_ATTN_BLOCK_BWD_LOOP = os.environ.get('FETA_BLOCK_BWD_LOOP', '1') != '0'


def _fused_attn_bwd(abi, b, n, d, heads, tie, dt):
    if not (USE_ATTN_BLOCK_BWD and not tie and abi.attn_block_bwd_supported(n, d, heads)):
        return False
    gb = abi.attn_block_bwd_blocks(b)
    return gb > 0 and (gb == b or dt != torch.float32 or USE_ATTN_BLOCK_BWD_LOOP)

def layer_params(layer):
    a = layer.self_attn
    return [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
            layer.norm1.weight, layer.norm1.bias, layer.linear1.weight, layer.linear1.bias,
            layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias]


# the split-K partials that are complete when the LAST launch of a stack's backward starts (every layer but the first, and
# the first layer's feed-forward half) are reduced in trailing workgroups of that launch (feta_attn_block_bwd_sums): the
# final reduction launch is left with that launch's own columns (colsum 10.5 -> 8.4 us, the launch itself 24.1 -> 25.6 us:
# 0.2749 -> 0.2735 ms per step at B = 128; at B >= 512, where that launch fills the chip, it costs 0.7 % - not taken there);
# 0: everything in the final launch (A/B timing)
USE_EARLY_COLSUM = os.environ.get('FETA_EARLY_COLSUM', '1') != '0'
USE_LN_STACK = os.environ.get('FETA_LN_STACK', '1') != '0'   # 0: LayerNorm layers run op by op (A/B timing)
# LayerNorm stacks: the CONSUMER of a pre-norm tensor applies the LayerNorm when it stages the rows (ABI 9, csrc/feta_ln.h:
# norm2 inside the next layer's attention block, norm1 inside the feed-forward kernel, their backward inside the gradient
# loads of feta_ffn_bwd / feta_attn_block_bwd) - two launches per layer and direction like the BatchNorm stack, no
# normalised tensor in HBM; 0: feta_layernorm_fwd / _bwd launches between the fused kernels (round-3 form, A/B timing)
USE_LN_ON_LOAD = os.environ.get('FETA_LN_ON_LOAD', '1') != '0'


def ln_on_load_supported(abi, layers, n, b, d_model, tie):
    """every launch of the stack is one of the four fused kernels (they carry the LayerNorm on load)"""
    if not (USE_LN_ON_LOAD and USE_ATTN_BLOCK and USE_FFN_FUSED and USE_FFN_BWD and USE_ATTN_BLOCK_BWD) or tie:
        return False
    heads = layers[0].self_attn.num_heads
    if not (abi.attn_block_supported(n, d_model, heads) and abi.attn_block_bwd_supported(n, d_model, heads)
            and abi.attn_block_bwd_blocks(b) > 0):
        return False
    return all(abi.ffn_supported(d_model, l.linear1.out_features) and abi.ffn_bwd_supported(d_model, l.linear1.out_features)
               for l in layers)


def lowp_stack_supported(abi, layers, n, b, d_model):
    """bf16 storage (layers.set_storage_dtype): the stack runs iff every launch of it is one of the four fused kernels
    (csrc/block.hip, ffn.hip, ffn_bwd.hip, block_bwd.hip - the ones instantiated for bf16 tiles - and, LayerNorm stacks,
    feta_layernorm_*_ex)."""
    if not (USE_ATTN_BLOCK and USE_FFN_FUSED and USE_FFN_BWD and USE_ATTN_BLOCK_BWD):
        return False
    l0 = layers[0]
    heads = l0.self_attn.num_heads
    if l0.self_attn.tie_qk or (not l0.batch_norm and not USE_LN_STACK):
        return False
    if not (abi.attn_block_supported(n, d_model, heads) and abi.attn_block_bwd_supported(n, d_model, heads)
            and abi.attn_block_bwd_blocks(b) > 0):
        return False
    return all(abi.ffn_supported(d_model, l.linear1.out_features) and abi.ffn_bwd_supported(d_model, l.linear1.out_features)
               for l in layers)


def stack_supported(layers, d_model):
    """BatchNorm stack (training mode: batch statistics) or LayerNorm stack (any mode), no dropout."""
    from .functional import ROWLIN_DIMS, layer_norm_rows_supported
    if not len(layers):
        return False
    bn = layers[0].batch_norm
    for l in layers:
        if l.batch_norm != bn:
            return False
        if l.training and (l.dropout1.p > 0.0 or l.dropout.p > 0.0 or l.dropout2.p > 0.0 or l.self_attn.dropout > 0.0):
            return False
        if getattr(l.self_attn, 'stab', 'rowmax') != 'rowmax':    # (the fused kernels implement exp(s - rowmax))
            return False
        if bn:
            if not l.training or l.norm1.momentum is None or l.norm2.momentum is None:
                return False
        else:
            if not USE_LN_STACK or not isinstance(l.norm1, torch.nn.LayerNorm):
                return False
            if not (l.norm1.elementwise_affine and l.norm2.elementwise_affine and layer_norm_rows_supported(d_model)):
                return False
        ff = l.linear1.out_features
        if not all(c in ROWLIN_DIMS for c in (d_model, 3 * d_model, ff)):
            return False
    return True


# TEST HOOK (tests/bench_checks.py): a list that receives the per-layer saved tensors of every fused-stack forward (the
# relu output h among them - which side of a zero-to-rounding pre-activation the kernels took); None: off
CAPTURE_SAVED = None

# first layer of a stack -> the flat gradient buffer of its last backward (kept off the modules:
# state_dict / deepcopy / pickle of a model must not see it)
STACK_FLAT_GRAD = weakref.WeakKeyDictionary()

# every consumer workgroup re-reduces the partial statistics of its producer: beyond this many rows ONE reduction launch
# runs in front of the consumers (~5 us against ~G x G x 512 bytes of L2 reads).  Measured (round 3): 512 beats 384 and
# 1024 at the batches where it matters - molhiv B = 1024 bf16 948 k / 960 k / 955 k graphs/s, ZINC B = 512 fp32
# 745 k / 773 k / 770 k (the feed-forward kernels emit 512 rows there: not capped any more)
MAX_STAT_ROWS = int(os.environ.get('FETA_MAX_STAT_ROWS', '512'))


def _cap_partials(abi, stream, st, new, shift_row=False):
    """[G (+ 1), 2, D] partial sums -> the same if G is small, else their total as [1 (+ 1), 2, D] (one extra reduction
    launch, only at batch sizes where a step takes milliseconds anyway).  shift_row: BatchNorm STATISTICS carry one more
    row, the shift their sums are relative to (csrc/feta_rowops.h); it is not a partial and travels unchanged.
    -> (buffer, number of partial rows)"""
    g = st.shape[0] - (1 if shift_row else 0)
    if g <= MAX_STAT_ROWS:
        return st, g
    tot = new(2 if shift_row else 1, 2, st.shape[2])
    if shift_row:
        # (the shift row travels as a one-row "sum" of the same launch: no copy launch)
        abi.colsum_multi([(st[:g].view(g, -1), tot[0].view(-1)), (st[g].view(1, -1), tot[1].view(-1))], stream)
    else:
        abi.colsum(st[:g].view(g, -1), tot[0].view(-1), stream)
    return tot, 1


def _partial_buffers(abi, new, m, b, n, d, heads, ff0, nl, fused_attn, ln_cols=False, ln_f=None, ln_a=None, ffn_fused=None):
    """Split-K partial buffers of a stack backward and the slot allocator.  Two buffers, because their row counts
    differ: 'f' (linear2 / linear1 of every layer) has one row per feta_rowlin_chunks(M) row chunk, 'a' (out_proj /
    in_proj) the same or - with the fused attention-block backward - one row per graph.  dwdb_all = [f columns |
    a columns | norm tail] is the flat gradient buffer every parameter gradient is a view of; ONE multi-segment
    reduction fills it at the end of backward.  -> (part_f, part_a, tf, ta, wslot)"""
    rc = abi.rowlin_chunks(m)
    ra = abi.attn_block_bwd_blocks(b) if fused_attn else rc
    # the fused feed-forward backward chooses its own split-K chunk count (feta_ffn_bwd_chunks: the row-wise kernels' chunks
    # or, where that keeps its grid within one round of resident workgroups, half as many)
    if ffn_fused is None:
        ffn_fused = USE_FFN_BWD and abi.ffn_bwd_supported(d, ff0)
    if ffn_fused:
        rc = abi.ffn_bwd_chunks(m, ff0)
    # ln_cols (LayerNorm on load): every layer's slot is followed by [dgamma | dbeta] of the LayerNorm whose backward
    # the kernel applies on its gradient load (norm2 behind the feed-forward slot, norm1 behind the attention slot)
    ln_f = ln_cols if ln_f is None else ln_f
    ln_a = ln_cols if ln_a is None else ln_a
    tf = nl * ((ff0 * d + ff0) + (d * ff0 + d) + (2 * d if ln_f else 0))
    ta = nl * ((d * d + d) + (3 * d * d + 3 * d) + (2 * d if ln_a else 0))
    part_f, part_a = new(rc, tf), new(ra, ta)
    cur = {'f': 0, 'a': 0}

    def wslot(kind, no, ki):
        """-> (pointer to this linear's partial columns, its offset in dwdb_all)"""
        off = cur[kind]
        cur[kind] += no * ki + no
        buf, base = (part_f, 0) if kind == 'f' else (part_a, tf)
        return buf.data_ptr() + 4 * off, base + off

    def raw(kind, count):
        """-> offset in dwdb_all of `count` further columns of this kind (they follow the previous slot in the row)"""
        off = cur[kind]
        cur[kind] += count
        return (0 if kind == 'f' else tf) + off

    wslot.cur = cur
    wslot.raw = raw
    return part_f, part_a, tf, ta, wslot


class StackTail:
    """Hand-over between the BatchNorm stack and the ONE consumer of its output when that consumer is linear_cat
    (transformer/models.py:223-224): the last BatchNorm is then never materialised either.  The stack returns the
    pre-norm y2 of the last layer and leaves its statistics here; linear_cat's forward kernel finalizes them and
    applies the normalisation inside its operand loads (functional.RowLinearCatBNFn), its backward kernel emits the
    partial sums (sum dout, sum dout * xhat) the BatchNorm backward needs from its dX epilogue.  Contract: the
    gradient that reaches the stack's first output is the gradient w.r.t. the NORMALISED tensor, with `gs` set.
    Saves the bn_apply_fwd and bn_bwd_reduce launches of a step."""

    def __init__(self):
        self.y2 = self.st2 = self.gamma = self.beta = self.norm = self.prm2 = self.gs = None
        self.G2 = 0


class FusedEncoderStackFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, src, pe, degree_rows, n_real, layers, need_attn, tail, pending, *params):
        abi, stream = _lib.backend(src, pe, n_real)
        ctx.set_materialize_grads(False)   # no zero tensor for the (non-differentiable) attn output
        if len(layers):
            STACK_FLAT_GRAD.pop(layers[0], None)   # the buffer of an earlier backward is stale from here on
        n, b, d = src.shape
        m = n * b
        nl = len(layers)
        heads = layers[0].self_attn.num_heads
        dh = d // heads
        tie = layers[0].self_attn.tie_qk
        scale = float(dh) ** -0.5
        dev = src.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        # bf16 storage (layers.set_storage_dtype; BASELINE configs 3 / 5): src and pe arrive as bf16, every token tensor
        # of the stack is bf16 (qkv, out, y1, h, y2 and the gradients in backward), the kernels run their bf16
        # instantiations (include/feta_hip.h: dtype = FETA_BF16).  Statistics, parameter blocks, attn, every parameter
        # gradient: fp32.  What LEAVES the stack is fp32 again - the last layer writes its y2 as fp32 (feta_ffn.y_f32) and
        # its per-head outputs once more as fp32 (feta_attn_block.out_f32) - so the filter stage behind it (coefficient
        # generator, spectral filter, linear_cat with the folded BatchNorm) runs unchanged and there is no cast launch.
        dt = src.dtype
        lowp = dt != torch.float32
        newt = lambda *s: torch.empty(s, dtype=dt, device=dev)
        if lowp and (pe is not None and pe.dtype != dt):
            raise TypeError('bf16 stack: pe must be %s as well' % dt)
        G = abi.rowlin_blocks(m)
        x_in = src.contiguous().view(m, d)
        pe_c = None if pe is None else pe.contiguous()
        block = USE_ATTN_BLOCK and abi.attn_block_supported(n, d, heads)
        if lowp and not lowp_stack_supported(abi, layers, n, b, d):
            raise NotImplementedError('bf16 storage: the fused stack needs d_model = 64, 4 heads, N <= 64, BatchNorm')
        saved = []
        y_prev, st_prev, prm_prev = x_in, None, None
        attn = None
        out32 = None
        for li, layer in enumerate(layers):
            (w_in, b_in, w_o, b_o, g1, be1, w1, bb1, w2, bb2, g2, be2) = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            ff = w1.shape[0]
            want = need_attn and li == nl - 1
            attn = new(b, heads, n, n) if want else None
            ast = new(b, heads, n, 2)
            qkv = newt(m, 3 * d)
            out = torch.empty((n, b, heads, dh), dtype=dt, device=dev)
            if lowp and li == nl - 1:
                out32 = new(n, b, heads, dh)
            y1 = newt(m, d)
            bn_prev = {}
            if li > 0:
                pl = layers[li - 1].norm2
                prm_prev = new(4, d)
                bn_prev = dict(x_stats=st_prev, Gx=G2_prev, x_gamma=params[(li - 1) * PER_LAYER + 10],
                               x_beta=params[(li - 1) * PER_LAYER + 11], x_bn_out=prm_prev,
                               x_rmean=pl.running_mean, x_rvar=pl.running_var, x_nbt=pl.num_batches_tracked,
                               momentum=float(pl.momentum), eps=float(pl.eps))
                saved[li - 1]['prm2'] = prm_prev
            if block:
                # F1 + F2 + F3 in one launch, one or two workgroups per graph (csrc/block.hip)
                G1 = abi.attn_block_stat_rows(b, n)
                st1 = new(G1 + 1, 2, d)      # (+ the shift row: the sums are relative to norm1's running mean)
                abi.attn_block_fwd(b, n, scale, stream, tie_qk=tie, x=y_prev, w_in=w_in, b_in=b_in, w_out=w_o,
                                   b_out=b_o, pe=pe_c, n_real=n_real, rowscale=degree_rows, qkv=qkv, out=out,
                                   attn_stats=ast, attn=attn, y=y1, y_stats=st1, y_shift=layer.norm1.running_mean,
                                   out_f32=(out32 if li == nl - 1 else None),
                                   sums=(pending.take_fwd() if (pending is not None and li == 0) else ()), **bn_prev)
                st1, G1 = _cap_partials(abi, stream, st1, new, shift_row=True)
            else:
                # F1
                dsc = abi.rowlin_ex(m, d, 3 * d, x=y_prev if li else x_in, w=w_in, bias=b_in, y=qkv, **bn_prev)
                abi.rowlin_fwd_ex(dsc, stream)
                if USE_ATTN_OUT and not lowp and abi.attn_out_supported(n, d, heads):
                    # F2 + F3 in one launch, one workgroup per (graph, 32 query rows) (csrc/attnout.hip)
                    G1 = abi.attn_out_stat_rows(b, n)
                    st1 = new(G1 + 1, 2, d)
                    abi.attn_out_fwd(b, n, scale, stream, tie_qk=tie, x=y_prev if li else x_in, x_bn=prm_prev, w_out=w_o,
                                     b_out=b_o, pe=pe_c, n_real=n_real, rowscale=degree_rows, qkv=qkv, out=out,
                                     attn_stats=ast, attn=attn, y=y1, y_stats=st1, y_shift=layer.norm1.running_mean,
                                     sums=(pending.take_fwd() if (pending is not None and li == 0) else ()))
                    st1, G1 = _cap_partials(abi, stream, st1, new, shift_row=True)
                else:
                    # F2
                    q, k, v = _views(qkv, n, b, heads, dh)
                    if tie:
                        k = q
                    abi.attn_fwd(q, k, v, pe_c, n_real, out.permute(1, 0, 2, 3), attn, ast, scale, stream)
                    # F3
                    G1 = G
                    st1 = new(G1 + 1, 2, d)
                    dsc = abi.rowlin_ex(m, d, d, x=out.view(m, d), w=w_o, bias=b_o, rowscale=degree_rows,
                                        residual=y_prev, res_bn=prm_prev, y=y1, stats=st1,
                                        stats_shift=layer.norm1.running_mean)
                    abi.rowlin_fwd_ex(dsc, stream)
            h, prm1 = newt(m, ff), new(4, d)
            n1 = layer.norm1
            bn1 = dict(x_stats=st1, Gx=G1, x_gamma=g1, x_beta=be1, x_bn_out=prm1, x_rmean=n1.running_mean,
                       x_rvar=n1.running_var, x_nbt=n1.num_batches_tracked, momentum=float(n1.momentum), eps=float(n1.eps))
            y2 = new(m, d) if li == nl - 1 else newt(m, d)    # (the stack's output is fp32 whatever the storage type)
            if USE_FFN_FUSED and abi.ffn_supported(d, ff):
                # F4 + F5 in one launch: the hidden activations stay in registers (csrc/ffn.hip)
                G2 = abi.ffn_blocks(m)
                st2 = new(G2 + 1, 2, d)
                abi.ffn_fwd(m, ff, stream, x=y1, w1=w1, b1=bb1, w2=w2, b2=bb2, h=h, y=y2, y_stats=st2,
                            y_shift=layer.norm2.running_mean,
                            coeff=_coeff_fwd_role(pending, li, nl, attn, n_real), **bn1)
                st2, G2 = _cap_partials(abi, stream, st2, new, shift_row=True)
            else:
                # F4
                dsc = abi.rowlin_ex(m, d, ff, relu=True, x=y1, w=w1, bias=bb1, y=h, **bn1)
                abi.rowlin_fwd_ex(dsc, stream)
                # F5
                G2 = G
                st2 = new(G2 + 1, 2, d)
                dsc = abi.rowlin_ex(m, ff, d, x=h, w=w2, bias=bb2, residual=y1, res_bn=prm1, y=y2, stats=st2,
                                    stats_shift=layer.norm2.running_mean)
                abi.rowlin_fwd_ex(dsc, stream)
            saved.append(dict(x0=y_prev, prm0=prm_prev, qkv=qkv, out=out, ast=ast, y1=y1, prm1=prm1, h=h, y2=y2))
            y_prev, st_prev, G2_prev = y2, st2, G2
        last = layers[-1].norm2
        prm2 = new(4, d)
        if tail is not None:
            # the consumer (linear_cat) finalizes and applies BN2 of the last layer: nothing is materialised
            tail.y2, tail.st2, tail.G2, tail.norm, tail.prm2 = y_prev, st_prev, G2_prev, last, prm2
            tail.gamma, tail.beta = params[(nl - 1) * PER_LAYER + 10], params[(nl - 1) * PER_LAYER + 11]
            tail.gs = None
            final = y_prev.view(m, d)
        else:
            # end of the stack: materialise BN2(y2) of the last layer
            final = new(m, d)
            abi.bn_apply_fwd_prm(y_prev, st_prev, params[(nl - 1) * PER_LAYER + 10], params[(nl - 1) * PER_LAYER + 11],
                                 final, prm2, last.running_mean, last.running_var, float(last.momentum),
                                 float(last.eps), stream, nbt=last.num_batches_tracked)
        ctx.tail = tail
        ctx.pending = pending
        saved[-1]['prm2'] = prm2
        ctx.saved_state = saved
        if CAPTURE_SAVED is not None:
            CAPTURE_SAVED.append(saved)
        ctx.meta = (n, b, d, heads, dh, tie, scale, G, nl)
        ctx.aux = (pe_c, degree_rows, n_real)
        ctx.params = params
        ctx.owner = layers[0] if len(layers) else None
        if attn is not None:
            ctx.mark_non_differentiable(attn)
        concat_last = (out32 if lowp else saved[-1]['out']).view(n, b, d)
        return final.view(n, b, d), concat_last, attn

    @staticmethod
    def backward(ctx, d_final, d_concat_last, _d_attn):
        saved, params = ctx.saved_state, ctx.params
        if d_final is None and d_concat_last is None:
            # nothing downstream used the stack (e.g. a loss on the filter coefficients alone: they derive from the
            # DETACHED attention matrix, transformer/models.py:282) - autograd still visits the node, with no gradient
            return (None,) * (8 + len(params))
        n, b, d, heads, dh, tie, scale, G, nl = ctx.meta
        pe_c, degree_rows, n_real = ctx.aux
        abi, stream = _lib.backend(saved[0]['qkv'])
        m = n * b
        dev = saved[0]['qkv'].device
        dt = saved[0]['qkv'].dtype      # storage type of the stack's token tensors (forward)
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        newt = lambda *s: torch.empty(s, dtype=dt, device=dev)
        grads = [None] * len(params)
        ff0 = params[6].shape[0]
        fused_attn = _fused_attn_bwd(abi, b, n, d, heads, tie, dt)
        # every weight/bias gradient of the stack goes through the split-K partial buffers and ONE deterministic
        # reduction at the end (instead of one reduction launch per linear)
        part_f, part_a, tf, ta, wslot = _partial_buffers(abi, new, m, b, n, d, heads, ff0, nl, fused_attn)
        coeff_req = _coeff_bwd_request(ctx, abi, stream, d, params[(nl - 1) * PER_LAYER + 6].shape[0])
        total = tf + ta
        # ONE flat gradient buffer for the whole stack: [weights and biases (reduced partials) | dgamma,
        # dbeta of norm1 / norm2 of every layer]; every parameter gradient returned below is a view of it,
        # so a data-parallel trainer all-reduces it in place (parallel.FlatBufferAllReduce)
        dwdb_all = new(total + nl * 4 * d)
        bn_tail = dwdb_all[total:].view(nl, 4, d)

        slots = {}
        early_done = None     # 'a' columns already reduced by the last launch (USE_EARLY_COLSUM)
        if d_final is None:   # only the per-head output of the last layer was used
            d_final = torch.zeros(n, b, d, dtype=torch.float32, device=dev)
        if dt != torch.float32 and d_final.dtype != torch.float32:
            d_final = d_final.float()      # (the stack's output is fp32: so is its gradient)
        dcur, dcur_b = d_final.contiguous().view(m, d), None
        if ctx.tail is not None:
            # (StackTail contract) d_final is the gradient w.r.t. BN2(y2); its partial sums came with it
            gs = ctx.tail.gs
            Gs_cur = G if gs is None else gs.shape[0]     # (row blocks of linear_cat's backward, or one row per graph: CatFold)
            ctx.tail.gs = None
            if gs is None:
                raise RuntimeError('fused stack: the consumer of the un-normalised output did not leave the '
                                   'BatchNorm backward sums (StackTail contract)')
        else:
            gs, Gs_cur = new(G, 2, d), G
            abi.bn_bwd_reduce(saved[-1]['y2'], dcur, saved[-1]['prm2'], gs, stream)
        for li in range(nl - 1, -1, -1):
            s = saved[li]
            (w_in, b_in, w_o, b_o, g1, be1, w1, bb1, w2, bb2, g2, be2) = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            ff = w1.shape[0]
            base = li * PER_LAYER
            fin2, dg2, db2 = new(2, d), bn_tail[li, 2], bn_tail[li, 3]
            pp, off = wslot('f', d, ff)
            slots[base + 8] = (off, d, ff)
            pp1, off1 = wslot('f', ff, d)
            slots[base + 6] = (off1, ff, d)
            dx1 = newt(m, d)
            if USE_FFN_BWD and abi.ffn_bwd_supported(d, ff):
                # B1 + B2 in one launch (csrc/ffn_bwd.hip): the hidden gradient never leaves the chip
                G1s = abi.ffn_bwd_blocks(m)
                gs1 = new(G1s, 2, d)
                abi.ffn_bwd(m, ff, stream, coeff=(coeff_req if li == nl - 1 else None), Gs=Gs_cur, partial_ptr=pp,
                            partial_ld=tf, dy=dcur, dy_b=dcur_b, g_y=s['y2'],
                            g_bn=s['prm2'], g_sum=gs, g_fin_out=fin2, dgamma=dg2, dbeta=db2, h=s['h'], w2=w2, w1=w1,
                            x=s['y1'], x_bn=s['prm1'], dx=dx1, sum_out=gs1)
                gs1, G1s = _cap_partials(abi, stream, gs1, new)
            else:
                assert dcur_b is None
                # B1: linear2 backward, gradient = BN2 backward of dcur
                dh_ = new(m, ff)
                dsc = abi.rowlin_ex(m, ff, d, x=s['h'], w=w2, dy=dcur, dx=dh_, partial_ptr=pp, partial_ld=tf,
                                    g_y=s['y2'], g_bn=s['prm2'], g_sum=gs, Gs=Gs_cur, g_fin_out=fin2, dgamma=dg2,
                                    dbeta=db2)
                abi.rowlin_bwd_ex(dsc, None, stream)
                # B2: linear1 backward (+ residual BN2 backward, + sums for BN1 backward)
                gs1, G1s = new(G, 2, d), G
                dsc = abi.rowlin_ex(m, d, ff, x=s['y1'], x_bn=s['prm1'], w=w1, dy=dh_, relu_y=s['h'], dx=dx1,
                                    partial_ptr=pp1, partial_ld=tf, add_dout=dcur, add_y=s['y2'],
                                    add_bn=s['prm2'], add_fin=fin2, sum_y=s['y1'], sum_bn=s['prm1'], sum_out=gs1)
                abi.rowlin_bwd_ex(dsc, None, stream)
            grads[base + 10], grads[base + 11] = dg2, db2
            fin1, dg1, db1 = new(2, d), bn_tail[li, 0], bn_tail[li, 1]
            grads[base + 4], grads[base + 5] = dg1, db1
            a_done = wslot.cur['a']      # 'a' columns of the layers behind this one: complete
            ppo, offo = wslot('a', d, d)
            slots[base + 2] = (offo, d, d)
            ppi, offi = wslot('a', 3 * d, d)
            slots[base + 0] = (offi, 3 * d, d)
            d2 = d_concat_last if (li == nl - 1 and d_concat_last is not None) else None
            if fused_attn:
                # B3 + B4 + B5 in one launch, one workgroup per graph (csrc/block_bwd.hip): dconcat and dqkv stay on chip
                dx0 = newt(m, d)
                early = ()
                if li == 0 and USE_EARLY_COLSUM and b <= 160:   # (measured: a gain only while this launch leaves CUs idle)
                    # the last launch of this backward: everything but its own partial columns is reduced beside it
                    early = [(part_f, dwdb_all[:tf])]
                    if a_done > 0:
                        early.append((part_a[:, :a_done], dwdb_all[tf:tf + a_done]))
                    early_done = a_done
                # the layer below takes the gradient in two parts iff its FFN backward is the fused kernel
                split = (USE_ATTN_BLOCK_SPLIT and li > 0 and USE_FFN_BWD and abi.attn_block_bwd_blocks(b) == b
                         and abi.ffn_bwd_supported(d, params[(li - 1) * PER_LAYER + 6].shape[0]))
                dx0b = newt(m, d) if split else None
                GB = abi.attn_block_bwd_blocks(b)
                gs_prev = new(2 * GB, 2, d) if li > 0 else None
                abi.attn_block_bwd(b, n, scale, stream, Gs=G1s, partial_ptr=ppo, partial_ld=ta, dy=dx1, y1=s['y1'],
                                   dx_b=dx0b,
                                   bn1=s['prm1'], g_sum=gs1, fin_out=fin1, dgamma=dg1, dbeta=db1, rowscale=degree_rows,
                                   w_out=w_o, w_in=w_in, qkv=s['qkv'], out=s['out'],
                                   dout2=None if d2 is None else d2.contiguous().view(m, d), pe=pe_c, n_real=n_real,
                                   attn_stats=s['ast'], x0=s['x0'], bn0=s['prm0'] if li > 0 else None, dx=dx0,
                                   sum_out=gs_prev, sums=early)
                Gs_next = 2 * GB
                if gs_prev is not None:
                    gs_prev, Gs_next = _cap_partials(abi, stream, gs_prev, new)
                dcur, dcur_b, gs, Gs_cur = dx0, dx0b, gs_prev, Gs_next
                continue
            # B3: out_proj backward, gradient = degree * BN1 backward of dx1
            dconcat = new(m, d)
            dsc = abi.rowlin_ex(m, d, d, x=s['out'].view(m, d), w=w_o, dy=dx1, rowscale=degree_rows, dx=dconcat,
                                partial_ptr=ppo, partial_ld=ta, g_y=s['y1'], g_bn=s['prm1'], g_sum=gs1, Gs=G1s,
                                g_fin_out=fin1, dgamma=dg1, dbeta=db1)
            abi.rowlin_bwd_ex(dsc, None, stream)
            dout2 = None
            if d2 is not None:
                if abi.attn_bwd_takes_dout2(n, dh, heads=heads):   # added inside the kernel's loads
                    dout2 = d2.contiguous().view(n, b, heads, dh).permute(1, 0, 2, 3)
                else:
                    dconcat = dconcat + d2.contiguous().view(m, d)
            # B4: attention backward
            q, k, v = _views(s['qkv'], n, b, heads, dh)
            if tie:
                k = q
            dqkv = new(m, 3 * d)
            dq, dk, dv = _views(dqkv, n, b, heads, dh)
            delta = new(b, heads, n)
            abi.attn_bwd(q, k, v, pe_c, n_real, s['out'].permute(1, 0, 2, 3),
                         dconcat.view(n, b, heads, dh).permute(1, 0, 2, 3), s['ast'], delta, dq, dk, dv, scale,
                         stream, dout2=dout2)
            if tie:
                dqkv[:, :d] += dqkv[:, d:2 * d]
                dqkv[:, d:2 * d] = 0
            # B5: in_proj backward (+ residual BN1 backward, + sums for the previous layer's BN2)
            dx0 = new(m, d)
            gs_prev = new(G, 2, d) if li > 0 else None
            dsc = abi.rowlin_ex(m, d, 3 * d, x=s['x0'], x_bn=s['prm0'], w=w_in, dy=dqkv, dx=dx0,
                                partial_ptr=ppi, partial_ld=ta, add_dout=dx1, add_y=s['y1'],
                                add_bn=s['prm1'], add_fin=fin1, sum_y=(s['x0'] if li > 0 else None),
                                sum_bn=s['prm0'], sum_out=gs_prev)
            abi.rowlin_bwd_ex(dsc, None, stream)
            Gs_next = G
            dcur, dcur_b, gs, Gs_cur = dx0, None, gs_prev, Gs_next
        assert dcur_b is None
        assert wslot.cur == {'f': tf, 'a': ta}
        # ... and whatever column sums the filter stage left for this launch (functional.PendingSums)
        if early_done is not None:
            last = [(part_a[:, early_done:], dwdb_all[tf + early_done:total])]
        else:
            last = [(part_f, dwdb_all[:tf]), (part_a, dwdb_all[tf:total])]
        abi.colsum_multi(last + _take_pending(ctx), stream)
        if ctx.owner is not None:
            STACK_FLAT_GRAD[ctx.owner] = dwdb_all
        for idx, (off, no, ki) in slots.items():
            grads[idx] = dwdb_all[off:off + no * ki].view(no, ki)
            if params[idx + 1] is not None:
                grads[idx + 1] = dwdb_all[off + no * ki:off + no * ki + no]
        return (dcur.view(n, b, d), None, None, None, None, None, None, None) + tuple(grads)


class FusedLayerNormStackFn(torch.autograd.Function):
    """The LayerNorm variant (batch_norm=False: the default of the reference's TU / molhiv / SBM scripts,
    experiments/run_transformer_gengcn_cv.py:56).  LayerNorm is row-local, so the normalised activations
    ARE materialised (feta_layernorm_fwd after each sub-layer) and every kernel reads plain operands.
    Per layer, forward (4 launches where csrc/block.hip and csrc/ffn.hip take the shape, else 7):
        y1 = x0 + degree * out_proj(attention(in_proj(x0)))     feta_attn_block_fwd
        x1 = LN1(y1)                                             feta_layernorm_fwd
        y2 = x1 + linear2(relu(linear1(x1)))                     feta_ffn_fwd
        x2 = LN2(y2)                                             feta_layernorm_fwd
    backward (7 launches): LN2, linear2, linear1 (+ residual dy2 in its dX epilogue), LN1, out_proj,
    attention, in_proj (+ residual dy1); the weight / bias partials of the whole stack share one
    [chunks, total] buffer and the LayerNorm partials another: two reductions per stack, into the same
    flat gradient buffer layout as the BatchNorm stack (parallel.FlatBufferAllReduce)."""

    @staticmethod
    def forward(ctx, src, pe, degree_rows, n_real, layers, need_attn, tail, pending, *params):
        abi, stream = _lib.backend(src, pe, n_real)
        ctx.set_materialize_grads(False)
        assert tail is None   # (LayerNorm is row-local: its output is materialised by feta_layernorm_fwd)
        ctx.pending = pending
        ctx.on_load = (len(layers) > 0 and ln_on_load_supported(abi, layers, src.shape[0], src.shape[1], src.shape[2],
                                                                 layers[0].self_attn.tie_qk))
        if ctx.on_load:
            return _ln_on_load_forward(ctx, abi, stream, src, pe, degree_rows, n_real, layers, need_attn, pending, params)
        if len(layers):
            STACK_FLAT_GRAD.pop(layers[0], None)
        n, b, d = src.shape
        m = n * b
        nl = len(layers)
        heads = layers[0].self_attn.num_heads
        dh = d // heads
        tie = layers[0].self_attn.tie_qk
        scale = float(dh) ** -0.5
        dev = src.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        # bf16 storage: as in the BatchNorm stack - bf16 token tensors inside, fp32 out of the last layer
        dt = src.dtype
        lowp = dt != torch.float32
        newt = lambda *s: torch.empty(s, dtype=dt, device=dev)
        x_in = src.contiguous().view(m, d)
        pe_c = None if pe is None else pe.contiguous()
        block = USE_ATTN_BLOCK and abi.attn_block_supported(n, d, heads)
        if lowp and not lowp_stack_supported(abi, layers, n, b, d):
            raise NotImplementedError('bf16 storage: the fused stack needs d_model = 64, 4 heads, N <= 64')
        saved = []
        attn = None
        out32 = None
        for li, layer in enumerate(layers):
            (w_in, b_in, w_o, b_o, g1, be1, w1, bb1, w2, bb2, g2, be2) = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            ff = w1.shape[0]
            want = need_attn and li == nl - 1
            attn = new(b, heads, n, n) if want else None
            ast = new(b, heads, n, 2)
            qkv = newt(m, 3 * d)
            out = torch.empty((n, b, heads, dh), dtype=dt, device=dev)
            if lowp and li == nl - 1:
                out32 = new(n, b, heads, dh)
            y1 = newt(m, d)
            if block:
                abi.attn_block_fwd(b, n, scale, stream, tie_qk=tie, x=x_in, w_in=w_in, b_in=b_in, w_out=w_o,
                                   b_out=b_o, pe=pe_c, n_real=n_real, rowscale=degree_rows, qkv=qkv, out=out,
                                   attn_stats=ast, attn=attn, y=y1, y_stats=new(abi.attn_block_stat_rows(b, n) + 1, 2, d),   # (statistics unused)
                                   out_f32=(out32 if li == nl - 1 else None),
                                   sums=(pending.take_fwd() if (pending is not None and li == 0) else ()))
            else:
                dsc = abi.rowlin_ex(m, d, 3 * d, x=x_in, w=w_in, bias=b_in, y=qkv)
                abi.rowlin_fwd_ex(dsc, stream)
                if USE_ATTN_OUT and not lowp and abi.attn_out_supported(n, d, heads):
                    abi.attn_out_fwd(b, n, scale, stream, tie_qk=tie, x=x_in, w_out=w_o, b_out=b_o, pe=pe_c, n_real=n_real,
                                     rowscale=degree_rows, qkv=qkv, out=out, attn_stats=ast, attn=attn, y=y1,
                                     sums=(pending.take_fwd() if (pending is not None and li == 0) else ()))
                else:
                    q, k, v = _views(qkv, n, b, heads, dh)
                    if tie:
                        k = q
                    abi.attn_fwd(q, k, v, pe_c, n_real, out.permute(1, 0, 2, 3), attn, ast, scale, stream)
                    dsc = abi.rowlin_ex(m, d, d, x=out.view(m, d), w=w_o, bias=b_o, rowscale=degree_rows,
                                        residual=x_in, y=y1)
                    abi.rowlin_fwd_ex(dsc, stream)
            h, y2 = newt(m, ff), newt(m, d)
            # norm1 on load (ABI 9) where the feed-forward half runs as its two fused kernels: x1 is never written, the
            # backward recomputes it from y1 and takes the LayerNorm backward of norm2 on its gradient load
            ffn_on_load = (USE_LN_ON_LOAD and USE_FFN_FUSED and USE_FFN_BWD and abi.ffn_supported(d, ff)
                           and abi.ffn_bwd_supported(d, ff))
            x1 = lst1 = None
            if ffn_on_load and li == nl - 1:
                y2 = new(m, d)      # (its gradient arrives as fp32: feta_ffn_bwd wants dy and g_y of one type)
            if not ffn_on_load:
                x1, lst1 = newt(m, d), new(m, 2)
                abi.layernorm_fwd(y1, g1, be1, float(layer.norm1.eps), x1, lst1, stream)
            x2 = lst2 = None
            if ffn_on_load:
                ln_out = {}
                if USE_LN_EPILOGUE:
                    # x2 = LN2(y2) from the same launch (feta_ffn.y_ln_out): its consumers here - in_proj, the residual of
                    # feta_attn_out_fwd - are not on-load kernels, so it is materialised, but not by a launch of its own
                    x2 = new(m, d) if li == nl - 1 else newt(m, d)
                    ln_out = dict(y_ln_out=x2, y_ln_gamma=g2, y_ln_beta=be2, y_ln_eps=float(layer.norm2.eps))
                abi.ffn_fwd(m, ff, stream, eps=float(layer.norm1.eps), x=y1, x_ln_gamma=g1, x_ln_beta=be1, w1=w1, b1=bb1,
                            w2=w2, b2=bb2, h=h, y=y2, coeff=_coeff_fwd_role(pending, li, nl, attn, n_real), **ln_out)
            elif USE_FFN_FUSED and abi.ffn_supported(d, ff):
                abi.ffn_fwd(m, ff, stream, x=x1, w1=w1, b1=bb1, w2=w2, b2=bb2, h=h, y=y2,
                            coeff=_coeff_fwd_role(pending, li, nl, attn, n_real))
            else:
                dsc = abi.rowlin_ex(m, d, ff, relu=True, x=x1, w=w1, bias=bb1, y=h)
                abi.rowlin_fwd_ex(dsc, stream)
                dsc = abi.rowlin_ex(m, ff, d, x=h, w=w2, bias=bb2, residual=x1, y=y2)
                abi.rowlin_fwd_ex(dsc, stream)
            if x2 is None:
                x2, lst2 = (new(m, d) if li == nl - 1 else newt(m, d)), new(m, 2)   # (the stack's output is fp32)
                abi.layernorm_fwd(y2, g2, be2, float(layer.norm2.eps), x2, lst2, stream)
            saved.append(dict(x0=x_in, qkv=qkv, out=out, ast=ast, y1=y1, lst1=lst1, x1=x1, h=h, y2=y2, lst2=lst2,
                              ffn_on_load=ffn_on_load))
            x_in = x2
        ctx.saved_state = saved
        if CAPTURE_SAVED is not None:
            CAPTURE_SAVED.append(saved)
        ctx.eps = [(float(l.norm1.eps), float(l.norm2.eps)) for l in layers]
        ctx.meta = (n, b, d, heads, dh, tie, scale, nl)
        ctx.aux = (pe_c, degree_rows, n_real)
        ctx.params = params
        ctx.owner = layers[0] if len(layers) else None
        if attn is not None:
            ctx.mark_non_differentiable(attn)
        concat_last = (out32 if lowp else saved[-1]['out']).view(n, b, d)
        # (a fresh tensor object for the output: the saved x2 of the last layer is not handed out)
        return x_in.view(n, b, d), concat_last, attn

    @staticmethod
    def backward(ctx, d_final, d_concat_last, _d_attn):
        saved, params = ctx.saved_state, ctx.params
        if d_final is None and d_concat_last is None:
            return (None,) * (8 + len(params))      # (see FusedEncoderStackFn.backward)
        if ctx.on_load:
            return _ln_on_load_backward(ctx, d_final, d_concat_last)
        n, b, d, heads, dh, tie, scale, nl = ctx.meta
        pe_c, degree_rows, n_real = ctx.aux
        abi, stream = _lib.backend(saved[0]['qkv'])
        m = n * b
        dev = saved[0]['qkv'].device
        dt = saved[0]['qkv'].dtype
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        newt = lambda *s: torch.empty(s, dtype=dt, device=dev)
        GL = abi.layernorm_blocks(m)
        grads = [None] * len(params)
        ff0 = params[6].shape[0]
        fused_attn = _fused_attn_bwd(abi, b, n, d, heads, tie, dt)
        # feed-forward half on its fused kernels with the LayerNorms on load (ABI 9): norm2's backward is taken on the
        # gradient load of feta_ffn_bwd ([dgamma2 | dbeta2] in its partial rows, behind db1), x1 = LN1(y1) on its operand
        # load; norm1's backward stays a launch (no saved statistics: recomputed from y1)
        ffn_on_load = all(s_['ffn_on_load'] for s_ in saved)     # (one feed-forward width per stack: all or none)
        part_f, part_a, tf, ta, wslot = _partial_buffers(
            abi, new, m, b, n, d, heads, ff0, nl, fused_attn, ln_f=ffn_on_load,
            ffn_fused=bool(ffn_on_load or (USE_FFN_BWD and abi.ffn_bwd_supported(d, ff0))))
        coeff_req = _coeff_bwd_request(ctx, abi, stream, d, params[(nl - 1) * PER_LAYER + 6].shape[0])
        total = tf + ta
        lnw = 2 if ffn_on_load else 4               # LayerNorm partial columns per layer that come from feta_layernorm_bwd
        ln_part = new(GL, nl * lnw * d)
        dwdb_all = new(total + nl * lnw * d)        # [feed-forward slots | attention slots | LayerNorm tail]
        ln_tail = dwdb_all[total:].view(nl, lnw, d)   # dgamma1, dbeta1 (, dgamma2, dbeta2) per layer

        def ln_bwd(dout, y, stats, gamma, li, which, eps=1e-5):
            dy = newt(m, d)
            abi.layernorm_bwd(dout, y, stats, gamma, dy, None, None, stream, partial_ld=nl * lnw * d,
                              partial_ptr=ln_part.data_ptr() + 4 * (li * lnw + 2 * which) * d, eps=eps)
            return dy

        slots = {}
        if d_final is None:
            d_final = torch.zeros(n, b, d, dtype=torch.float32, device=dev)
        if dt != torch.float32 and d_final.dtype != torch.float32:
            d_final = d_final.float()
        dcur = d_final.contiguous().view(m, d)
        for li in range(nl - 1, -1, -1):
            s = saved[li]
            (w_in, b_in, w_o, b_o, g1, be1, w1, bb1, w2, bb2, g2, be2) = params[li * PER_LAYER:(li + 1) * PER_LAYER]
            ff = w1.shape[0]
            base = li * PER_LAYER
            eps1, eps2 = ctx.eps[li]
            if not ffn_on_load:
                dy2 = ln_bwd(dcur, s['y2'], s['lst2'], g2, li, 1, eps2)
                grads[base + 10], grads[base + 11] = ln_tail[li, 2], ln_tail[li, 3]
            pp, off = wslot('f', d, ff)
            slots[base + 8] = (off, d, ff)
            pp1, off1 = wslot('f', ff, d)
            slots[base + 6] = (off1, ff, d)
            dx1 = newt(m, d)
            if ffn_on_load:
                offl = wslot.raw('f', 2 * d)
                grads[base + 10], grads[base + 11] = dwdb_all[offl:offl + d], dwdb_all[offl + d:offl + 2 * d]
                abi.ffn_bwd(m, ff, stream, coeff=(coeff_req if li == nl - 1 else None), partial_ptr=pp, partial_ld=tf,
                            dy=dcur, g_y=s['y2'], g_ln_gamma=g2, ln_eps=eps2, h=s['h'], w2=w2, w1=w1, x=s['y1'],
                            x_ln_gamma=g1, x_ln_beta=be1, dx=dx1)
            elif USE_FFN_BWD and abi.ffn_bwd_supported(d, ff):
                # linear2 + linear1 backward in one launch (csrc/ffn_bwd.hip), dx1 = dy2 + dh W1
                abi.ffn_bwd(m, ff, stream, coeff=(coeff_req if li == nl - 1 else None), partial_ptr=pp, partial_ld=tf,
                            dy=dy2, h=s['h'], w2=w2, w1=w1, x=s['x1'],
                            dx=dx1)
            else:
                # linear2, then linear1 with the residual gradient dy2 added in its dX epilogue
                dh_ = new(m, ff)
                dsc = abi.rowlin_ex(m, ff, d, x=s['h'], w=w2, dy=dy2, dx=dh_, partial_ptr=pp, partial_ld=tf)
                abi.rowlin_bwd_ex(dsc, None, stream)
                dsc = abi.rowlin_ex(m, d, ff, x=s['x1'], w=w1, dy=dh_, relu_y=s['h'], dx=dx1, partial_ptr=pp1,
                                    partial_ld=tf, add_plain=dy2)
                abi.rowlin_bwd_ex(dsc, None, stream)
            dy1 = ln_bwd(dx1, s['y1'], s['lst1'], g1, li, 0, eps1)     # (lst1 None: LayerNorm on load - recomputed)
            grads[base + 4], grads[base + 5] = ln_tail[li, 0], ln_tail[li, 1]
            ppo, offo = wslot('a', d, d)
            slots[base + 2] = (offo, d, d)
            ppi, offi = wslot('a', 3 * d, d)
            slots[base + 0] = (offi, 3 * d, d)
            d2 = d_concat_last if (li == nl - 1 and d_concat_last is not None) else None
            if fused_attn:
                # out_proj + attention + in_proj backward in one launch, one workgroup per graph (csrc/block_bwd.hip)
                dx0 = newt(m, d)
                abi.attn_block_bwd(b, n, scale, stream, partial_ptr=ppo, partial_ld=ta, dy=dy1, rowscale=degree_rows,
                                   w_out=w_o, w_in=w_in, qkv=s['qkv'], out=s['out'],
                                   dout2=None if d2 is None else d2.contiguous().view(m, d), pe=pe_c, n_real=n_real,
                                   attn_stats=s['ast'], x0=s['x0'], dx=dx0)
                dcur = dx0
                continue
            # out_proj (gradient scaled by degree), attention, in_proj with the residual gradient dy1
            dconcat = new(m, d)
            dsc = abi.rowlin_ex(m, d, d, x=s['out'].view(m, d), w=w_o, dy=dy1, rowscale=degree_rows, dx=dconcat,
                                partial_ptr=ppo, partial_ld=ta)
            abi.rowlin_bwd_ex(dsc, None, stream)
            dout2 = None
            if d2 is not None:
                if abi.attn_bwd_takes_dout2(n, dh, heads=heads):
                    dout2 = d2.contiguous().view(n, b, heads, dh).permute(1, 0, 2, 3)
                else:
                    dconcat = dconcat + d2.contiguous().view(m, d)
            q, k, v = _views(s['qkv'], n, b, heads, dh)
            if tie:
                k = q
            dqkv = new(m, 3 * d)
            dq, dk, dv = _views(dqkv, n, b, heads, dh)
            delta = new(b, heads, n)
            abi.attn_bwd(q, k, v, pe_c, n_real, s['out'].permute(1, 0, 2, 3),
                         dconcat.view(n, b, heads, dh).permute(1, 0, 2, 3), s['ast'], delta, dq, dk, dv, scale,
                         stream, dout2=dout2)
            if tie:
                dqkv[:, :d] += dqkv[:, d:2 * d]
                dqkv[:, d:2 * d] = 0
            dx0 = new(m, d)
            dsc = abi.rowlin_ex(m, d, 3 * d, x=s['x0'], w=w_in, dy=dqkv, dx=dx0, partial_ptr=ppi, partial_ld=ta,
                                add_plain=dy1)
            abi.rowlin_bwd_ex(dsc, None, stream)
            dcur = dx0
        assert wslot.cur == {'f': tf, 'a': ta}
        abi.colsum_multi([(part_f, dwdb_all[:tf]), (part_a, dwdb_all[tf:total]), (ln_part, dwdb_all[total:])]
                         + _take_pending(ctx), stream)
        if ctx.owner is not None:
            STACK_FLAT_GRAD[ctx.owner] = dwdb_all
        for idx, (off, no, ki) in slots.items():
            grads[idx] = dwdb_all[off:off + no * ki].view(no, ki)
            if params[idx + 1] is not None:
                grads[idx + 1] = dwdb_all[off + no * ki:off + no * ki + no]
        return (dcur.view(n, b, d), None, None, None, None, None, None, None) + tuple(grads)


def _ln_on_load_forward(ctx, abi, stream, src, pe, degree_rows, n_real, layers, need_attn, pending, params):
    """LayerNorm stack with the LayerNorm applied by the consumer of each pre-norm tensor (ABI 9).  Per layer, forward
    (2 launches):
        y1 = x0 + degree * out_proj(attention(in_proj(x0))),  x0 = LN2_prev(y2_prev) on load     feta_attn_block_fwd
        y2 = x1 + linear2(relu(linear1(x1))),                 x1 = LN1(y1) on load               feta_ffn_fwd
    and ONE feta_layernorm_fwd at the end of the stack (norm2 of the last layer: the output is handed out).  Backward
    (2 launches per layer + one reduction): feta_ffn_bwd takes the gradient w.r.t. LN2(y2) and applies the LayerNorm
    backward where it loads the gradient rows, feta_attn_block_bwd the same for LN1; [dgamma | dbeta] ride in the
    kernels' split-K partial rows.  No normalised activation is written between the kernels."""
    if len(layers):
        STACK_FLAT_GRAD.pop(layers[0], None)
    n, b, d = src.shape
    m = n * b
    nl = len(layers)
    heads = layers[0].self_attn.num_heads
    dh = d // heads
    scale = float(dh) ** -0.5
    dev = src.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    dt = src.dtype
    lowp = dt != torch.float32
    newt = lambda *s: torch.empty(s, dtype=dt, device=dev)
    if lowp and not lowp_stack_supported(abi, layers, n, b, d):
        raise NotImplementedError('bf16 storage: the fused stack needs d_model = 64, 4 heads, N <= 64')
    x_pre = src.contiguous().view(m, d)      # pre-norm input rows of the layer (layer 0: the stack input itself)
    pe_c = None if pe is None else pe.contiguous()
    saved = []
    attn = out32 = None
    ln_prev = {}
    for li, layer in enumerate(layers):
        (w_in, b_in, w_o, b_o, g1, be1, w1, bb1, w2, bb2, g2, be2) = params[li * PER_LAYER:(li + 1) * PER_LAYER]
        ff = w1.shape[0]
        attn = new(b, heads, n, n) if (need_attn and li == nl - 1) else None
        ast = new(b, heads, n, 2)
        qkv = newt(m, 3 * d)
        out = torch.empty((n, b, heads, dh), dtype=dt, device=dev)
        if lowp and li == nl - 1:
            out32 = new(n, b, heads, dh)
        y1 = newt(m, d)
        abi.attn_block_fwd(b, n, scale, stream, x=x_pre, w_in=w_in, b_in=b_in, w_out=w_o, b_out=b_o, pe=pe_c, n_real=n_real,
                           rowscale=degree_rows, qkv=qkv, out=out, attn_stats=ast, attn=attn, y=y1, y_stats=None,
                           out_f32=(out32 if li == nl - 1 else None),
                           sums=(pending.take_fwd() if (pending is not None and li == 0) else ()), **ln_prev)
        h = newt(m, ff)
        y2 = new(m, d) if li == nl - 1 else newt(m, d)    # (what leaves the stack is fp32 whatever the storage type)
        # norm2 of the LAST layer is what the stack hands out: the feed-forward kernel writes it beside y2 from its epilogue
        # (feta_ffn.y_ln_out) - the stack launches no LayerNorm kernel at all
        ln_out = {}
        if li == nl - 1 and USE_LN_EPILOGUE:
            final = new(m, d)
            ln_out = dict(y_ln_out=final, y_ln_gamma=g2, y_ln_beta=be2, y_ln_eps=float(layer.norm2.eps))
        abi.ffn_fwd(m, ff, stream, eps=float(layer.norm1.eps), x=y1, x_ln_gamma=g1, x_ln_beta=be1, w1=w1, b1=bb1, w2=w2,
                    b2=bb2, h=h, y=y2, y_stats=None, coeff=_coeff_fwd_role(pending, li, nl, attn, n_real), **ln_out)
        saved.append(dict(x0=x_pre, qkv=qkv, out=out, ast=ast, y1=y1, h=h, y2=y2))
        x_pre = y2
        ln_prev = dict(x_ln_gamma=g2, x_ln_beta=be2, eps=float(layer.norm2.eps))
    if not USE_LN_EPILOGUE:
        # norm2 of the last layer as a launch of its own (FETA_LN_EPILOGUE=0: A/B timing)
        last = layers[-1].norm2
        final = new(m, d)
        abi.layernorm_fwd(x_pre, params[(nl - 1) * PER_LAYER + 10], params[(nl - 1) * PER_LAYER + 11], float(last.eps), final,
                          new(m, 2), stream)
    ctx.saved_state = saved
    if CAPTURE_SAVED is not None:
        CAPTURE_SAVED.append(saved)
    ctx.meta = (n, b, d, heads, dh, False, scale, nl)
    ctx.aux = (pe_c, degree_rows, n_real)
    ctx.params = params
    ctx.eps = [(float(l.norm1.eps), float(l.norm2.eps)) for l in layers]
    ctx.owner = layers[0] if len(layers) else None
    if attn is not None:
        ctx.mark_non_differentiable(attn)
    concat_last = (out32 if lowp else saved[-1]['out']).view(n, b, d)
    return final.view(n, b, d), concat_last, attn


def _ln_on_load_backward(ctx, d_final, d_concat_last):
    saved, params = ctx.saved_state, ctx.params
    n, b, d, heads, dh, tie, scale, nl = ctx.meta
    pe_c, degree_rows, n_real = ctx.aux
    abi, stream = _lib.backend(saved[0]['qkv'])
    m = n * b
    dev = saved[0]['qkv'].device
    dt = saved[0]['qkv'].dtype
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    newt = lambda *s: torch.empty(s, dtype=dt, device=dev)
    grads = [None] * len(params)
    ff0 = params[6].shape[0]
    part_f, part_a, tf, ta, wslot = _partial_buffers(abi, new, m, b, n, d, heads, ff0, nl, True, ln_cols=True, ffn_fused=True)
    coeff_req = _coeff_bwd_request(ctx, abi, stream, d, params[(nl - 1) * PER_LAYER + 6].shape[0])
    total = tf + ta
    dwdb_all = new(total)      # [feed-forward slots | attention slots], each followed by its LayerNorm's [dgamma | dbeta]
    slots, ln_slots = {}, {}
    if d_final is None:
        d_final = torch.zeros(n, b, d, dtype=torch.float32, device=dev)
    if d_final.dtype != torch.float32:
        d_final = d_final.float()      # (the stack's output is fp32: so is its gradient)
    dcur, dcur_b = d_final.contiguous().view(m, d), None      # gradient w.r.t. LN2(y2) of the layer at hand
    GB = abi.attn_block_bwd_blocks(b)
    for li in range(nl - 1, -1, -1):
        s = saved[li]
        (w_in, b_in, w_o, b_o, g1, be1, w1, bb1, w2, bb2, g2, be2) = params[li * PER_LAYER:(li + 1) * PER_LAYER]
        ff = w1.shape[0]
        base = li * PER_LAYER
        eps1, eps2 = ctx.eps[li]
        pp, off = wslot('f', d, ff)
        slots[base + 8] = (off, d, ff)
        _, off1 = wslot('f', ff, d)
        slots[base + 6] = (off1, ff, d)
        ln_slots[base + 10] = wslot.raw('f', 2 * d)
        dx1 = newt(m, d)
        # linear2 + linear1 backward; LN2 backward on the gradient load, x1 = LN1(y1) on the operand load
        abi.ffn_bwd(m, ff, stream, coeff=(coeff_req if li == nl - 1 else None), partial_ptr=pp, partial_ld=tf,
                    dy=dcur, dy_b=dcur_b, g_y=s['y2'], g_ln_gamma=g2, ln_eps=eps2, h=s['h'], w2=w2, w1=w1, x=s['y1'],
                    x_ln_gamma=g1, x_ln_beta=be1, dx=dx1)
        ppo, offo = wslot('a', d, d)
        slots[base + 2] = (offo, d, d)
        _, offi = wslot('a', 3 * d, d)
        slots[base + 0] = (offi, 3 * d, d)
        ln_slots[base + 4] = wslot.raw('a', 2 * d)
        d2 = d_concat_last if (li == nl - 1 and d_concat_last is not None) else None
        # out_proj + attention + in_proj backward; LN1 backward on the gradient load, x0 = LN2_prev(y2_prev) on load.  The
        # layer below takes its gradient in two parts (two workgroups per graph, one per pair of heads)
        split = USE_ATTN_BLOCK_SPLIT and li > 0 and GB == b
        dx0 = newt(m, d)
        dx0b = newt(m, d) if split else None
        ln0 = {}
        if li > 0:
            ln0 = dict(x0_ln_gamma=params[(li - 1) * PER_LAYER + 10], x0_ln_beta=params[(li - 1) * PER_LAYER + 11])
        abi.attn_block_bwd(b, n, scale, stream, partial_ptr=ppo, partial_ld=ta, dy=dx1, y1=s['y1'], ln1_gamma=g1,
                           ln_eps=eps1, dx_b=dx0b, rowscale=degree_rows, w_out=w_o, w_in=w_in, qkv=s['qkv'], out=s['out'],
                           dout2=None if d2 is None else d2.contiguous().view(m, d), pe=pe_c, n_real=n_real,
                           attn_stats=s['ast'], x0=s['x0'], dx=dx0, **ln0)
        dcur, dcur_b = dx0, dx0b
    assert dcur_b is None
    assert wslot.cur == {'f': tf, 'a': ta}
    abi.colsum_multi([(part_f, dwdb_all[:tf]), (part_a, dwdb_all[tf:total])] + _take_pending(ctx), stream)
    if ctx.owner is not None:
        STACK_FLAT_GRAD[ctx.owner] = dwdb_all
    for idx, (off, no, ki) in slots.items():
        grads[idx] = dwdb_all[off:off + no * ki].view(no, ki)
        if params[idx + 1] is not None:
            grads[idx + 1] = dwdb_all[off + no * ki:off + no * ki + no]
    for idx, off in ln_slots.items():
        grads[idx], grads[idx + 1] = dwdb_all[off:off + d], dwdb_all[off + d:off + 2 * d]
    return (dcur.view(n, b, d), None, None, None, None, None, None, None) + tuple(grads)


def _coeff_fwd_role(pending, li, nl, attn, n_real):
    """The coefficient generator's forward rides in the launch of the LAST layer's feed-forward half."""
    if pending is None or li != nl - 1:
        return None
    return pending.coeff_fwd_role(attn, n_real)


def _coeff_bwd_request(ctx, abi, stream, d, ff_last):
    """The coefficient generator's backward kernel, left by its autograd node for the first launch of this backward
    (the last layer's fused FFN backward); run here on its own when that launch is not the fused kernel."""
    from . import functional as F
    req = ctx.pending.take_coeff_bwd() if ctx.pending is not None else None
    if req is not None and (not (USE_FFN_BWD and abi.ffn_bwd_supported(d, ff_last))
                            or req[6] * req[8] > F.COEFF_ROLE_MAX_BLOCKS):
        cj, n_real, s, gb, dpooled, partial, b, n, h = req
        abi.coeff_bwd(cj, n_real, s, gb, dpooled, partial, None, None, b, n, h, stream)
        req = None
    return req


def _take_pending(ctx):
    """Column sums the filter stage handed to the stack's reduction launch; from here on its nodes reduce on their
    own (a node of the stage that autograd schedules after this one)."""
    pend = ctx.pending
    if pend is None:
        return []
    pend.stack_done = True
    return pend.take()


def fused_encoder_stack(src, pe, degree_rows, n_real, layers, need_attn=True, tail=None, pending=None):
    """-> (output [N,B,d] of the last layer, concat heads of the last layer [N,B,d], attn or None).
    tail (a StackTail, BatchNorm stacks only): the output is the PRE-norm y2 of the last layer and the tail's
    consumer applies the last BatchNorm (functional.row_linear_cat_bn).
    pending (a functional.PendingSums whose stack_armed the caller set): the stack's one reduction launch also
    carries the column sums the filter stage's backward left in it."""
    params = []
    for l in layers:
        params += layer_params(l)
    bn = layers[0].batch_norm
    fn = FusedEncoderStackFn if bn else FusedLayerNormStackFn
    return fn.apply(src, pe, degree_rows, n_real, list(layers), need_attn, tail if bn else None, pending, *params)
