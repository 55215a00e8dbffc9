"""Parity of libfeta_hip.so (hand-written HIP, gfx950) against the CPU oracle through the C ABI.
Same checks as test_kernels_emu.py, on the MI355X, plus BASELINE-size cases."""
import pytest
import torch

import kernel_checks as KC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('bsz,n,h,dh,use_pe,seq_first', [
    (3, 20, 2, 16, True, True),
    (2, 37, 4, 16, False, True),
    (2, 37, 4, 16, True, False),
    (2, 64, 4, 16, True, True),        # 4 heads x 16: one workgroup per graph in backward, 4 row tiles
    (3, 12, 4, 16, False, False),
    (2, 50, 2, 8, True, True),
    (2, 33, 1, 32, True, False),
    (1, 70, 1, 64, True, True),
    (2, 9, 2, 4, False, True),
    (128, 37, 4, 16, True, True),      # BASELINE config 2 (ZINC shape)
    (32, 28, 4, 16, True, True),       # config 1 (MUTAG shape)
    (8, 188, 4, 16, True, True),       # config 4 (PATTERN shape, N_pad 188)
    (64, 128, 4, 16, True, True),      # config 4 at N_pad 128: one workgroup per (graph, head) in backward
    (5, 117, 4, 16, True, False),      # ... odd N_pad, batch-first
    (3, 65, 4, 16, False, True),       # ... five row tiles, no positional kernel
    (16, 222, 4, 16, False, True),     # config 5 (molhiv, largest bucket)
    (4, 256, 2, 32, True, True),       # FETA_MAX_NODES
])
def test_attn(hip, bsz, n, h, dh, use_pe, seq_first):
    abi, dev, stream = hip
    KC.check_attn(abi, dev, stream, bsz, n, h, dh, use_pe, seq_first)


@pytest.mark.parametrize('bsz,n,use_pe,tie_qk,with_bn,write_attn,n_min', [
    (3, 65, True, False, False, True, 1),       # five query tiles, three chunks of 32 rows
    (64, 128, True, False, True, True, 44),     # BASELINE config 4 as timed (PATTERN, B = 64, N_pad = 128)
    (64, 128, False, False, False, False, 44),  # ... pe=None, no attn write (every layer but the last)
    (16, 188, True, True, True, True, 44),      # N_pad 188, K tied to Q
    (16, 188, False, False, False, True, 100),
    (4, 256, True, False, True, True, 200),     # the largest graph the kernel takes
    (5, 120, False, True, False, False, 1),
])
def test_attn_out_against_oracle(hip, bsz, n, use_pe, tie_qk, with_bn, write_attn, n_min):
    """feta_attn_out_fwd directly against oracle.attention_core + out_proj + degree + residual (VERDICT round 3, weak #2:
    until now it was only compared with the two launches it replaces)"""
    KC.check_attn_out(hip[0], hip[1], hip[2], bsz, n, use_pe=use_pe, tie_qk=tie_qk, with_bn=with_bn,
                      write_attn=write_attn, n_min=n_min)


def test_attn_no_attn_write(hip):
    abi, dev, stream = hip
    KC.check_attn(abi, dev, stream, 2, 21, 2, 16, True, write_attn=False)


def test_attn_clamped_rows(hip):
    abi, dev, stream = hip
    KC.check_attn(abi, dev, stream, 2, 19, 2, 16, True, clamp_case=True)


@pytest.mark.parametrize('bsz,n,h,c,faithful', [(3, 12, 2, 64, True), (2, 37, 4, 256, True),
                                                (1, 5, 1, 16, True), (16, 37, 4, 1024, False),
                                                (2, 200, 4, 1024, False), (64, 128, 4, 1024, False),
                                                (3, 190, 2, 1100, False)])
def test_coeff(hip, bsz, n, h, c, faithful):
    abi, dev, stream = hip
    KC.check_coeff(abi, dev, stream, bsz, n, h, c, faithful=faithful)


@pytest.mark.parametrize('directed', [False, True])
def test_lhat_from_edges(hip, directed):
    abi, dev, stream = hip
    KC.check_lhat(abi, dev, stream, bsz=16, directed=directed)


@pytest.mark.parametrize('mode', ['cheb', 'spec'])
@pytest.mark.parametrize('share', [0, 1])
@pytest.mark.parametrize('bsz,h,dh,order,shape,n_min,n_max,seq_first', [
    (3, 2, 16, 4, 'zinc', None, None, True),
    (2, 4, 16, 4, 'mutag', None, None, False),
    (2, 2, 8, 3, 'zinc', 2, 20, True),
    (2, 1, 32, 2, 'zinc', 17, 40, True),
    (1, 2, 16, 1, 'zinc', None, None, True),
    (2, 2, 16, 5, 'pattern', 44, 70, True),
    (32, 4, 16, 4, 'zinc', None, None, True),
    (4, 4, 16, 4, 'pattern', 100, 188, True),
    (2, 1, 64, 4, 'zinc', 30, 60, True),
])
def test_filter_exact(hip, mode, share, bsz, h, dh, order, shape, n_min, n_max, seq_first):
    abi, dev, stream = hip
    KC.check_filter(abi, dev, stream, mode, bsz, h, dh, order, share, shape=shape, n_min=n_min,
                    n_max=n_max, seq_first=seq_first)


def test_cheb_directed_graph(hip):
    abi, dev, stream = hip
    KC.check_filter(abi, dev, stream, 'cheb', 2, 2, 16, 4, 1, directed=True)


@pytest.mark.parametrize('k_eig,bsz', [(8, 3), (16, 3), (16, 64), (32, 4)])
def test_spec_truncated(hip, k_eig, bsz):
    abi, dev, stream = hip
    shape = 'pattern' if k_eig == 32 else 'zinc'
    KC.check_filter(abi, dev, stream, 'spec', bsz, 4, 16, 4, 1, k_eig=k_eig, shape=shape)
    KC.check_filter(abi, dev, stream, 'spec', 2, 2, 16, 4, 0, k_eig=k_eig)


@pytest.mark.parametrize('k_eig,shape,n_min,n_max,order,seq_first,bsz', [
    (16, 'zinc', None, None, 4, True, 130),
    (8, 'mutag', None, None, 4, False, 5),
    (32, 'pattern', 44, 64, 3, True, 9),
    (None, 'zinc', 12, 32, 5, True, 4),
    (32, 'pattern', 100, 188, 4, True, 9),
    (16, 'pattern', 70, 120, 4, False, 5),
])
def test_spec_one_workgroup_per_graph(hip, k_eig, shape, n_min, n_max, order, seq_first, bsz):
    abi, dev, stream = hip
    KC.check_filter(abi, dev, stream, 'spec', bsz, 4, 16, order, 1, k_eig=k_eig, shape=shape, n_min=n_min,
                    n_max=n_max, seq_first=seq_first)


def test_bad_arguments_raise(hip):
    abi, dev, stream = hip
    q = torch.zeros(2, 300, 2, 16, device=dev)
    with pytest.raises(ValueError):
        abi.attn_fwd(q, q, q, None, torch.ones(2, dtype=torch.int32, device=dev), torch.zeros_like(q),
                     None, torch.zeros(2, 2, 300, 2, device=dev), 0.25, stream)


@pytest.mark.parametrize('m,ki,no,relu,rowscale,residual,stats', [
    (37 * 128, 64, 192, False, False, False, False),   # in_proj at BASELINE config 2
    (37 * 128, 64, 64, False, True, True, True),       # out_proj + degree + residual + BN stats
    (37 * 128, 64, 128, True, False, False, False),    # linear1 + relu
    (37 * 128, 128, 64, False, False, True, True),     # linear2 + residual + BN stats
    (33, 32, 32, False, False, False, False),
    (200, 16, 16, True, True, False, True),
    (100000, 64, 64, False, True, True, True),         # more row blocks than partial slots
    (1000, 256, 256, False, False, False, False),
    (1000, 192, 64, False, False, False, False),
])
def test_rowlin(hip, m, ki, no, relu, rowscale, residual, stats):
    abi, dev, stream = hip
    KC.check_rowlin(abi, dev, stream, m, ki, no, relu, rowscale, residual, stats)


@pytest.mark.parametrize('m,d', [(37 * 128, 64), (64, 32), (300, 128), (50, 192), (70000, 64)])
def test_batchnorm(hip, m, d):
    abi, dev, stream = hip
    KC.check_bn(abi, dev, stream, m, d)


def test_batchnorm_statistics_far_from_zero(hip):
    """column mean ~ 10^3 standard deviations: output and running_var against F.batch_norm in fp64, 1e-5 relative"""
    abi, dev, stream = hip
    errs = KC.check_bn_far_from_zero(abi, dev, stream)
    print(errs)
    assert errs['naive_rel_var'] > 1e-3      # (what E[y^2] - mean^2 alone would have given on this data)


@pytest.mark.parametrize('r,c', [(19, 4096 + 64), (3, 8192), (64, 4100), (74, 4096), (300, 4096), (200, 48), (1, 16)])
def test_colsum_shapes(hip, r, c):
    KC.check_colsum(*hip, r, c)


def test_colsum_multi_mixed_segments(hip):
    KC.check_colsum_multi_mixed(*hip)


@pytest.mark.parametrize('r,k,n,with_dx', [(512, 512, 512, True), (36, 48, 80, True), (20, 16, 16, False), (8, 100, 36, True),
                                           (4096, 512, 512, True), (520, 256, 256, True)])
def test_lin_gemm(hip, r, k, n, with_dx):
    """the C x C linear of the coefficient generator: forward and the one-launch backward, ragged tiles"""
    KC.check_lin(*hip, r, k, n, with_dx=with_dx)


@pytest.mark.parametrize('bsz,n,use_pe,seq_first,clamp', [
    (3, 37, True, True, False), (2, 64, True, False, False), (4, 9, False, True, False), (2, 19, True, True, True),
])
def test_attn_bwd_one_workgroup_per_graph(hip, monkeypatch, bsz, n, use_pe, seq_first, clamp):
    """4 heads x dh 16: attn_bwd_graph_kernel (chosen by itself from 192 graphs up; forced here)"""
    monkeypatch.setenv('FETA_ATTN_BWD_GRAPH', '1')
    KC.check_attn(*hip, bsz, n, 4, 16, use_pe, seq_first, clamp_case=clamp)


# ---- spectrum producer ----


@pytest.mark.parametrize('shape,bsz,n_min,n_max,n_pad', [
    ('zinc', 128, None, None, None),
    ('mutag', 32, 3, 16, 16),
    ('pattern', 16, 66, 128, None),
    ('pattern', 8, 150, 188, None),     # BASELINE PATTERN shape: three 64-row chunks, 141 KB of LDS
    ('molhiv', 64, None, 64, 64),
])
def test_eigh_sym(hip, shape, bsz, n_min, n_max, n_pad):
    abi, dev, stream = hip
    KC.check_eigh(abi, dev, stream, shape, bsz, 0, n_min, n_max, n_pad)


def test_eigh_sym_max_n(hip):
    abi, dev, stream = hip
    KC.check_eigh(abi, dev, stream, 'pattern', 4, 1, 180, 192, 192)


def test_eigh_sym_truncated_equals_full(hip):
    abi, dev, stream = hip
    KC.check_eigh_truncated_equals_full(abi, dev, stream)


@pytest.mark.parametrize('kind,zero_diag,from_device', [('diffusion', False, True), ('pstep', True, True),
                                                        ('diffusion', True, False), ('pstep', False, False)])
@pytest.mark.parametrize('shape,n_min,n_max', [('zinc', None, None), ('pattern', 100, 188)])
def test_spectral_kernel(hip, kind, zero_diag, from_device, shape, n_min, n_max):
    abi, dev, stream = hip
    KC.check_spectral_kernel(abi, dev, stream, kind, shape=shape, bsz=8, n_min=n_min, n_max=n_max,
                             zero_diag=zero_diag, from_device_eigh=from_device)


@pytest.mark.parametrize('m,d', [(4736, 64), (37, 64), (5, 32), (7000, 128), (33, 200), (100000, 64), (1, 256)])
def test_layernorm(hip, m, d):
    abi, dev, stream = hip
    KC.check_layernorm(abi, dev, stream, m, d)


@pytest.mark.parametrize('bsz,n,h,dh,use_pe,seq_first', [(128, 37, 4, 16, True, True), (5, 64, 2, 32, False, False),
                                                          (3, 222, 4, 16, True, True), (2, 256, 1, 64, True, True)])
def test_attn_bf16(hip, bsz, n, h, dh, use_pe, seq_first):
    """feta_attn_fwd_bf16 / feta_attn_bwd_bf16 (bf16 storage, bf16 MFMA) against the fp64 oracle, KC.BF16_TOL"""
    abi, dev, stream = hip
    KC.check_attn(abi, dev, stream, bsz, n, h, dh, use_pe, seq_first, dtype=KC.BF16)


def test_attn_bf16_clamped_rows(hip):
    abi, dev, stream = hip
    KC.check_attn(abi, dev, stream, 2, 19, 2, 16, True, clamp_case=True, dtype=KC.BF16)


@pytest.mark.parametrize('bsz,k_eig,share,dh,shape,n_max', [(128, 16, 1, 16, 'zinc', 37), (16, 8, 0, 16, 'mutag', 28),
                                                            (4, 32, 1, 16, 'pattern', 120), (6, 20, 1, 32, 'zinc', 30)])
def test_spec_filter_bf16(hip, bsz, k_eig, share, dh, shape, n_max):
    abi, dev, stream = hip
    KC.check_filter(abi, dev, stream, 'spec', bsz, 4 if dh == 16 else 2, dh, 4, share, shape=shape, n_max=n_max,
                    k_eig=k_eig, dtype=KC.BF16)


def test_eigh_sym_workspace_variant(hip):
    """192 < N <= 256 (BASELINE config 5's largest bucket: 222 nodes) stays on the device"""
    abi, dev, stream = hip
    KC.check_eigh(abi, dev, stream, 'molhiv', 5, 0, 150, 222, 222)
    KC.check_eigh(abi, dev, stream, 'pattern', 3, 1, 200, 256, 256)


@pytest.mark.parametrize('bsz,n,h,dh,p,dtype', [(128, 37, 4, 16, 0.1, torch.float32), (3, 222, 4, 16, 0.5, torch.float32),
                                                 (4, 64, 2, 32, 0.25, KC.BF16)])
def test_attn_dropout(hip, bsz, n, h, dh, p, dtype):
    """feta_attn_fwd_drop / feta_attn_bwd_drop: the oracle holds the mask the kernels derive from (seed, offset)"""
    abi, dev, stream = hip
    KC.check_attn(abi, dev, stream, bsz, n, h, dh, True, drop=(p, 1234567891011, 7), dtype=dtype)


# ---- fused layer-stack kernels on bf16 storage (dtype = FETA_BF16: BASELINE configs 3 / 5) --------------------------------
@pytest.mark.parametrize('kw', [dict(), dict(bsz=3, n_pad=37, n_min=9, with_pe=False),
                                dict(bsz=2, n_pad=16, n_min=1, need_attn=False), dict(bsz=2, n_pad=64, n_min=40),
                                dict(bsz=128, n_pad=37, n_min=9), dict(bsz=300, n_pad=37, n_min=9, seed=2)])
def test_attn_block_fwd_bf16(hip, kw):
    abi, dev, stream = hip
    KC.check_attn_block_lp(abi, dev, stream, **kw)


@pytest.mark.parametrize('env', [dict(FETA_BLOCK_FWD_WAVES='4'), dict(FETA_BLOCK_FWD_WGS='1'),
                                 dict(FETA_BLOCK_FWD_WGS='2', FETA_BLOCK_MAX_GRID='2')])
@pytest.mark.parametrize('kw', [dict(bsz=40, n_pad=37, n_min=9), dict(bsz=7, n_pad=64, n_min=40, with_pe=False)])
def test_attn_block_fwd_bf16_forms(hip, monkeypatch, env, kw):
    abi, dev, stream = hip
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    KC.check_attn_block_lp(abi, dev, stream, **kw)


@pytest.mark.parametrize('kw', [dict(), dict(split=True), dict(bsz=3, n_pad=37, n_min=9, with_pe=False, with_bn=True),
                                dict(bsz=2, n_pad=16, n_min=1, split=True, with_bn=True), dict(bsz=2, n_pad=64, n_min=40),
                                dict(bsz=128, n_pad=37, n_min=9, split=True, with_bn=True),
                                dict(bsz=128, n_pad=37, n_min=9, split=False, with_bn=True, seed=4)])
def test_attn_block_bwd_bf16(hip, kw):
    abi, dev, stream = hip
    KC.check_attn_block_bwd_lp(abi, dev, stream, **kw)


@pytest.mark.parametrize('kw', [dict(), dict(m=33, ff=64, with_bn=False), dict(m=130, ff=256, seed=3), dict(m=4736, ff=128)])
def test_ffn_fwd_bf16(hip, kw):
    abi, dev, stream = hip
    KC.check_ffn_lp(abi, dev, stream, **kw)


@pytest.mark.parametrize('kw', [dict(), dict(m=70, ff=64, with_bn=False), dict(m=300, ff=128, seed=3), dict(m=4736, ff=128)])
def test_ffn_bwd_bf16(hip, kw):
    abi, dev, stream = hip
    KC.check_ffn_bwd_lp(abi, dev, stream, **kw)


@pytest.mark.parametrize('r,k,n,with_dx,bf16', [(512, 1024, 1024, True, False), (512, 1024, 1024, True, True),
                                                (64, 256, 256, True, False), (1024, 256, 512, False, True),
                                                (4096, 1024, 1024, True, False)])
def test_lin_gemm_tiled(hip, monkeypatch, r, k, n, with_dx, bf16):
    """the LDS-tiled kernels of csrc/lin.hip at the BASELINE shape of ``self.linear`` (transformer/models.py:145,284:
    R = H*B = 512 rows, C = 1024) in both compute types, at the smallest shape they take and at a large batch"""
    monkeypatch.setenv('FETA_LIN_TILED', '2')
    abi, dev, stream = hip
    KC.check_lin(abi, dev, stream, r, k, n, with_dx=with_dx, bf16=bf16)


@pytest.mark.parametrize('bsz,n,h,dh,use_pe,dtype', [(3, 20, 2, 16, True, torch.float32), (2, 37, 4, 16, False, torch.float32),
                                                     (2, 70, 1, 64, True, torch.float32), (2, 33, 2, 32, True, torch.bfloat16)])
def test_attn_stab_clamp5(hip, bsz, n, h, dh, use_pe, dtype):
    """stab = clamp5 (witnesses LSPE/layers/graphit_gt_layer.py:39-43): exp(clamp(s, -5, 5)), forward and
    backward with zero gradient through clamped scores, fp32 and bf16 storage"""
    KC.check_attn(*hip, bsz, n, h, dh, use_pe, dtype=dtype, clamp5=True)


# ---- LayerNorm on load in the fused stack kernels (ABI 9) --------------------------------------------------------------
F32, B16 = torch.float32, torch.bfloat16


@pytest.mark.parametrize('kw', [dict(m=4736, ff=128, dtype=F32), dict(m=4736, ff=128, dtype=B16), dict(m=75, ff=64, dtype=F32),
                                dict(m=33, ff=256, dtype=B16), dict(m=65536, ff=128, dtype=F32)])
def test_ffn_fwd_layernorm_on_load(hip, kw):
    KC.check_ffn_ln(hip[0], hip[1], hip[2], **kw)


@pytest.mark.parametrize('kw', [dict(bsz=128, n_pad=37, n_min=9, dtype=F32), dict(bsz=128, n_pad=37, n_min=9, dtype=B16),
                                dict(bsz=32, n_pad=28, n_min=10, dtype=F32, with_pe=False, need_attn=False),
                                dict(bsz=300, n_pad=64, n_min=2, dtype=F32, with_pe=False),     # workgroups walk the batch
                                dict(bsz=300, n_pad=64, n_min=2, dtype=B16), dict(bsz=3, n_pad=21, dtype=F32)])
def test_attn_block_fwd_layernorm_on_load(hip, kw):
    KC.check_attn_block_ln(hip[0], hip[1], hip[2], **kw)


@pytest.mark.parametrize('env', [{'FETA_BLOCK_FWD_WAVES': '4'}, {'FETA_BLOCK_FWD_WGS': '1'}, {'FETA_BLOCK_FWD_WGS': '2'}])
def test_attn_block_fwd_layernorm_on_load_forms(hip, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    KC.check_attn_block_ln(hip[0], hip[1], hip[2], bsz=64, n_pad=37, n_min=5, dtype=F32)


@pytest.mark.parametrize('kw', [dict(m=4736, ff=128, dtype=F32), dict(m=4736, ff=128, dtype=B16, two_parts=True),
                                dict(m=150, ff=64, dtype=F32, two_parts=True), dict(m=65536, ff=128, dtype=F32),
                                dict(m=19200, ff=128, dtype=B16)])
def test_ffn_bwd_layernorm_on_load(hip, kw):
    KC.check_ffn_bwd_ln(hip[0], hip[1], hip[2], **kw)


@pytest.mark.parametrize('kw', [dict(bsz=128, n_pad=37, n_min=9, dtype=F32), dict(bsz=128, n_pad=37, n_min=9, dtype=F32, split=True),
                                dict(bsz=128, n_pad=37, n_min=9, dtype=B16, split=True),
                                dict(bsz=32, n_pad=28, n_min=10, dtype=F32, with_pe=False, first_layer=True),
                                dict(bsz=300, n_pad=64, n_min=2, dtype=F32, with_pe=False),     # the graph-walking form
                                dict(bsz=300, n_pad=64, n_min=2, dtype=B16), dict(bsz=3, n_pad=21, dtype=F32)])
def test_attn_block_bwd_layernorm_on_load(hip, kw):
    KC.check_attn_block_bwd_ln(hip[0], hip[1], hip[2], **kw)


@pytest.mark.parametrize('kw', [dict(norm='bn_fresh', bsz=128), dict(norm='bn_block', k_eig=8, bsz=32), dict(norm='plain', shape='mutag', k_eig=8, bsz=32),
                                dict(norm='bn_fresh', shape='molhiv', n_max=64, k_eig=16, bsz=300),
                                dict(norm='bn_fresh', shape='pattern', n_min=44, n_max=128, k_eig=32, bsz=64),
                                dict(norm='plain', shape='pattern', n_min=70, n_max=100, k_eig=32, bsz=8)])
def test_spec_filter_with_linear_cat(hip, kw):
    """feta_spec_filter_cat_fwd (linear_cat folded into the per-graph eigenbasis filter) against the oracle"""
    KC.check_spec_cat(hip[0], hip[1], hip[2], **kw)


@pytest.mark.parametrize('kw', [dict(norm='bn_block', bsz=128), dict(norm='plain', k_eig=8, bsz=32), dict(norm='bn_block', shape='mutag', k_eig=8, bsz=32),
                                dict(norm='bn_block', shape='molhiv', n_max=64, k_eig=16, bsz=300),
                                dict(norm='bn_block', shape='pattern', n_min=44, n_max=64, k_eig=32, bsz=16)])
def test_spec_filter_with_linear_cat_backward(hip, kw):
    """feta_spec_filter_cat_bwd (the filter's backward with linear_cat's backward inside) against fp64 autograd"""
    KC.check_spec_cat_bwd(hip[0], hip[1], hip[2], **kw)
