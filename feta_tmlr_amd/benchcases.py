"""Launch closures of the kernels of one fused-stack layer in exactly the variants the stack issues
(feta_tmlr_amd/fused_stack.py), with their algorithmic HBM bytes - shared by
bench.py (roofline object) and tools/kernel_bench.py (per-kernel timing, PMC traffic passes).

Every entry: (name, launches per layer, fn, algorithmic bytes per launch, kernel symbols of one call).
"""
import torch


def stack_layer_cases(abi, st, dev, b, n, d, heads, ff, pe, n_real, last_layer_attn=True, seed=0, dtype=torch.float32):
    """dtype: storage type of the token tensors (torch.bfloat16: the bf16 instantiations of the four fused kernels;
    the row-wise cases stay fp32 - they have no bf16 form); byte counts: ft bytes per token element, 4 per other"""
    m = b * n
    dh = d // heads
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    new = lambda *s: torch.empty(*s, device=dev)
    rndt = lambda *s: torch.randn(*s, generator=g).to(dev).to(dtype)
    newt = lambda *s: torch.empty(*s, device=dev, dtype=dtype)
    f4 = 4
    ft = 2 if dtype == torch.bfloat16 else 4
    if pe is not None:
        pe = pe.to(dtype)
    G = abi.rowlin_blocks(m)
    RC = abi.rowlin_chunks(m)
    cases = []
    prm = torch.rand(4, max(d, ff, 3 * d), generator=g).to(dev) + 0.5

    def prm_of(c):
        return prm[:, :c].contiguous()

    if abi.attn_block_supported(n, d, heads):
        x, w_in, b_in = rndt(m, d), rnd(3 * d, d) / d ** 0.5, rnd(3 * d)
        w_o, b_o, deg = rnd(d, d) / d ** 0.5, rnd(d), torch.rand(m, generator=g).to(dev)
        qkv, out, y1, st1 = newt(m, 3 * d), newt(m, d), newt(m, d), new(abi.attn_block_stat_rows(b, n) + 1, 2, d)
        ast, attn = new(b, heads, n, 2), new(b, heads, n, n)
        stats_prev = rnd(G + 1, 2, d).abs()      # (partial rows + the shift row)
        common = dict(x=x, w_in=w_in, b_in=b_in, w_out=w_o, b_out=b_o, pe=pe, n_real=n_real, rowscale=deg,
                      qkv=qkv, out=out, attn_stats=ast, y=y1, y_stats=st1, x_stats=stats_prev, Gx=G,
                      x_gamma=prm_of(d)[0], x_beta=prm_of(d)[1], x_bn_out=new(4, d))
        base_t = b * (n * d + (n * n if pe is not None else 0) + 3 * n * d + n * d + n * d)   # token elements
        base_o = b * 2 * heads * n + 4 * d * d                                                    # fp32 elements
        scale = dh ** -0.5
        # descriptors are built once: the eager timing loop must not be bound by Python
        d0 = abi.attn_block_desc(b, n, scale, attn=None, **common)
        d1 = abi.attn_block_desc(b, n, scale, attn=attn, **common)
        cases.append(('attn_block_fwd (no attn write)', 1.0,
                      lambda: (abi.attn_block_launch(d0, st), common)[0], ft * base_t + f4 * base_o,
                      ['attn_block_fwd']))
        if last_layer_attn:
            cases.append(('attn_block_fwd (+attn write)', 0.0,
                          lambda: (abi.attn_block_launch(d1, st), common)[0],
                          ft * base_t + f4 * (base_o + b * heads * n * n), ['attn_block_fwd']))
    if abi.ffn_supported(d, ff):
        x, w1, b1, w2, b2 = rndt(m, d), rnd(ff, d) / d ** 0.5, rnd(ff), rnd(d, ff) / ff ** 0.5, rnd(d)
        hbuf, y2, st2 = newt(m, ff), newt(m, d), new(abi.ffn_blocks(m) + 1, 2, d)
        # partial rows of the statistics this launch finalizes = workgroups of the forward block in front of it (two per
        # graph at the BASELINE batch); the stack caps them (fused_stack.MAX_STAT_ROWS)
        g1 = abi.attn_block_stat_rows(b, n) if abi.attn_block_supported(n, d, heads) else b
        g1 = g1 if g1 <= 384 else 1
        stats1 = rnd(g1 + 1, 2, d).abs()
        fkw = dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, h=hbuf, y=y2, y_stats=st2, x_stats=stats1,
                   x_gamma=prm_of(d)[0], x_beta=prm_of(d)[1], x_bn_out=new(4, d))
        fd = abi.ffn_desc(m, ff, Gx=g1, **fkw)
        cases.append(('ffn_fwd', 1.0, lambda: (abi.ffn_launch(fd, st), fkw)[0],
                      ft * (m * d + m * ff + m * d) + f4 * 2 * d * ff, ['ffn_fwd']))
        # the last layer's launch carries the coefficient generator's forward in trailing workgroups
        # (feta_ffn_fwd_coeff): C = P dh^2 channels at the reference's filter order 4
        cgen = 4 * (d // heads) ** 2
        attn_in = torch.rand(b, heads, n, n, generator=g).to(dev)
        crole = (attn_in, n_real, rnd(cgen), rnd(cgen), new(heads * b, n), new(heads * b, cgen))
        cases.append(('ffn_fwd (+ coefficient generator)', 1.0, lambda: (abi.ffn_launch(fd, st, crole), fkw)[0],
                      ft * (m * d + m * ff + m * d) + f4 * (2 * d * ff + b * (heads * n * n + heads * cgen + heads * n)),
                      ['ffn_fwd']))

    def bwd_case(name, ki, no, extras):
        x, w, dy, dx = rnd(m, ki), rnd(no, ki) / ki ** 0.5, rnd(m, no), new(m, ki)
        total = no * ki + no
        part = new(RC, total)
        kw = dict(x=x, w=w, dy=dy, dx=dx, partial_ptr=part.data_ptr(), partial_ld=total)
        nbytes = m * ki + no * ki + m * no + m * ki + RC * total        # x, W, dy -> dx, partials
        if 'g' in extras:       # BatchNorm backward folded into the gradient loads
            kw.update(g_y=rnd(m, no), g_bn=prm_of(no), g_sum=rnd(G, 2, no), Gs=G, g_fin_out=new(2, no),
                      dgamma=new(no), dbeta=new(no))
            nbytes += m * no
        if 'r' in extras:       # relu mask from the saved activation
            kw.update(relu_y=rnd(m, no))
            nbytes += m * no
        if 'a' in extras:       # residual gradient through a BatchNorm backward, added in the epilogue
            kw.update(add_dout=rnd(m, ki), add_y=rnd(m, ki), add_bn=prm_of(ki), add_fin=rnd(2, ki))
            nbytes += 2 * m * ki
        if 's' in extras:       # partial sums for the previous BatchNorm backward (its y is x itself)
            kw.update(sum_y=x, sum_bn=prm_of(ki), sum_out=new(G, 2, ki))
        dsc = abi.rowlin_ex(m, ki, no, **kw)
        keep = (x, w, dy, dx, part, kw)
        cases.append((name, 1.0, lambda: (abi.rowlin_bwd_ex(dsc, None, st), keep)[0], f4 * nbytes, ['rowlin_bwd']))

    from .fused_stack import _fused_attn_bwd
    fused_f = abi.ffn_bwd_supported(d, ff)
    fused_a = _fused_attn_bwd(abi, b, n, d, heads, False, dtype)     # (what the stack issues at this batch / type)
    if fused_f:
        # backward of the FFN half in one launch (csrc/ffn_bwd.hip), BatchNorm stack variant
        dy, y2, hh, y1, dx = rndt(m, d), rndt(m, d), rndt(m, ff), rndt(m, d), newt(m, d)
        w2, w1 = rnd(d, ff) / ff ** 0.5, rnd(ff, d) / d ** 0.5
        cols = 2 * d * ff + d + ff
        part = new(RC, cols)
        RCF = abi.ffn_bwd_chunks(m, ff)      # partial rows this kernel writes (feta_ffn_bwd_chunks)
        xb = abi.ffn_bwd_blocks(m)
        kw = dict(dy=dy, g_y=y2, g_bn=prm_of(d), g_sum=rnd(G, 2, d), g_fin_out=new(2, d), dgamma=new(d), dbeta=new(d),
                  h=hh, w2=w2, w1=w1, x=y1, x_bn=prm_of(d), dx=dx, sum_out=new(xb, 2, d))
        fdsc = abi.ffn_bwd_desc(m, ff, Gs=G, partial_ld=cols, partial_ptr=part.data_ptr(), **kw)
        keep_f = (kw, part)
        cases.append(('ffn_bwd', 1.0, lambda: (abi.ffn_bwd_launch(fdsc, st), keep_f)[0],
                      ft * (4 * m * d + m * ff) + f4 * (2 * d * ff + RCF * cols), ['ffn_bwd']))
        # the last layer's launch (the first of the stack's backward) carries the coefficient generator's backward
        # kernel in trailing workgroups (feta_ffn_bwd_coeff)
        cgen = 4 * (d // heads) ** 2
        cgrp = abi.coeff_bwd_groups(b, heads)
        brole = (rnd(heads * b, n), n_real, rnd(cgen), rnd(cgen), rnd(heads * b, cgen), new(cgrp, 2, cgen), b, n, heads)
        cases.append(('ffn_bwd (+ coefficient generator)', 1.0, lambda: (abi.ffn_bwd_launch(fdsc, st, brole), keep_f)[0],
                      ft * (4 * m * d + m * ff) + f4 * (2 * d * ff + RCF * cols + b * (heads * cgen + heads * n) + cgrp * 2 * cgen),
                      ['ffn_bwd']))
        if fused_a:
            # ... below a layer whose attention backward ran as two workgroups per graph: the gradient in two parts
            # (its BatchNorm sums come from that launch: two partial rows per graph)
            gs2 = 2 * abi.attn_block_bwd_blocks(b)
            gs2 = gs2 if gs2 <= 384 else 1
            kw2 = dict(kw, dy_b=rndt(m, d), g_sum=rnd(gs2, 2, d))
            fdsc2 = abi.ffn_bwd_desc(m, ff, Gs=gs2, partial_ld=cols, partial_ptr=part.data_ptr(), **kw2)
            keep_f2 = (kw2, part)
            cases.append(('ffn_bwd (gradient in two parts)', 1.0, lambda: (abi.ffn_bwd_launch(fdsc2, st), keep_f2)[0],
                          ft * (5 * m * d + m * ff) + f4 * (2 * d * ff + RCF * cols), ['ffn_bwd']))
    else:
        bwd_case('rowlin_bwd linear2 (stack: BN-backward gradient)', ff, d, 'g')
        bwd_case('rowlin_bwd linear1 (stack: relu, add, sums)', d, ff, 'ras')
    if fused_a:
        # backward of the attention sub-block in one launch (csrc/block_bwd.hip), BatchNorm stack variant
        dy, y1, x0, qkv, out, dx = rndt(m, d), rndt(m, d), rndt(m, d), rndt(m, 3 * d), rndt(m, d), newt(m, d)
        w_o, w_in = rnd(d, d) / d ** 0.5, rnd(3 * d, d) / d ** 0.5
        ast = torch.rand(b, heads, n, 2, generator=g).to(dev) + 1.0
        deg = torch.rand(m, generator=g).to(dev)
        cols = 4 * d * d + 4 * d
        gb = abi.attn_block_bwd_blocks(b)     # partial rows = workgroups (they walk the graphs beyond 256)
        part = new(gb, cols)
        kw = dict(dy=dy, y1=y1, bn1=prm_of(d), g_sum=rnd(G, 2, d), fin_out=new(2, d), dgamma=new(d), dbeta=new(d),
                  rowscale=deg, w_out=w_o, w_in=w_in, qkv=qkv, out=out, pe=pe, n_real=n_real, attn_stats=ast, x0=x0,
                  bn0=prm_of(d), dx=dx, sum_out=new(2 * gb, 2, d))
        keep_a = (kw, part)
        cases.append(('attn_block_bwd', 1.0,
                      lambda: (abi.attn_block_bwd(b, n, dh ** -0.5, st, Gs=G, partial_ptr=part.data_ptr(), partial_ld=cols,
                                                  **kw), keep_a)[0],
                      ft * (8 * m * d + (b * n * n if pe is not None else 0)) + f4 * (2 * b * heads * n + 4 * d * d + gb * cols),
                      ['attn_block_bwd<false>']))
        if fused_f and gb == b:
            # two workgroups per graph (layers whose consumer is the fused FFN backward): dx in two parts
            kws = dict(kw, dx_b=newt(m, d))
            keep_s = (kws, part)
            cases.append(('attn_block_bwd (two workgroups per graph)', 1.0,
                          lambda: (abi.attn_block_bwd(b, n, dh ** -0.5, st, Gs=G, partial_ptr=part.data_ptr(),
                                                      partial_ld=cols, **kws), keep_s)[0],
                          ft * (9 * m * d + (b * n * n if pe is not None else 0)) + f4 * (2 * b * heads * n + 4 * d * d + b * cols),
                          ['attn_block_bwd<true>']))
    else:
        bwd_case('rowlin_bwd out_proj (stack: BN-backward gradient)', d, d, 'g')
        bwd_case('rowlin_bwd in_proj (stack: add, sums)', d, 3 * d, 'as')
    # linear_cat (2d -> d) of the filter stage: forward and backward, once per step
    bwd_case('rowlin_bwd linear_cat', ff if ff == 2 * d else 2 * d, d, '')
    return cases
