"""The kernel sources (feta_tmlr_amd/csrc/*.hip) compiled for the host by tools/simt and
checked against the oracle through the C ABI - no GPU needed.  The same checks run on the
MI355X in test_kernels_gpu.py."""
import os

import pytest
import torch

import kernel_checks as KC

CPU = torch.device('cpu')


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
    (2, 100, 4, 16, True, True),       # 64 < N <= 128, 4 heads x 16: one workgroup per (graph, head) in backward
    (1, 65, 4, 16, False, False),
    (1, 150, 4, 16, True, True),       # 128 < N <= 256: the same kernel with pe read from global memory
])
def test_attn(emu, bsz, n, h, dh, use_pe, seq_first):
    KC.check_attn(emu, CPU, None, bsz, n, h, dh, use_pe, seq_first)


@pytest.mark.parametrize('bsz,n,h,dh,use_pe,seq_first', [(3, 21, 4, 16, True, True), (2, 37, 2, 32, False, False),
                                                          (2, 70, 1, 64, True, True)])
def test_attn_bf16(emu, bsz, n, h, dh, use_pe, seq_first):
    """bf16 storage entry points (feta_attn_fwd_bf16 / feta_attn_bwd_bf16) against the fp64 oracle"""
    KC.check_attn(emu, CPU, None, bsz, n, h, dh, use_pe, seq_first, dtype=KC.BF16)
    KC.check_attn(emu, CPU, None, 2, 19, 2, 16, True, clamp_case=True, dtype=KC.BF16)


@pytest.mark.parametrize('k_eig,share,dh', [(8, 1, 16), (16, 0, 16), (20, 1, 32)])
def test_spec_filter_bf16(emu, k_eig, share, dh):
    KC.check_filter(emu, CPU, None, 'spec', 3, 2, dh, 4, share, shape='zinc', n_min=5, n_max=30, k_eig=k_eig,
                    dtype=KC.BF16)


@pytest.mark.parametrize('bsz,n,h,dh,p,dtype', [(3, 21, 4, 16, 0.1, torch.float32), (2, 37, 2, 32, 0.5, torch.float32),
                                                 (2, 70, 1, 64, 0.25, torch.float32), (3, 21, 4, 16, 0.2, KC.BF16)])
def test_attn_dropout(emu, bsz, n, h, dh, p, dtype):
    """attention-probability dropout (feta_attn_fwd_drop / _bwd_drop): forward, the written (dropped) attn and the
    backward against the oracle holding the SAME mask, rebuilt on the host from (seed, offset)"""
    KC.check_attn(emu, CPU, None, bsz, n, h, dh, True, drop=(p, 1234567891011, 7), dtype=dtype)
    KC.check_attn(emu, CPU, None, bsz, n, h, dh, False, seq_first=False, drop=(p, 5, 2 ** 33 + 1), dtype=dtype)


def test_attn_no_attn_write(emu):
    KC.check_attn(emu, CPU, None, 2, 21, 2, 16, True, write_attn=False)


def test_attn_clamped_rows(emu):
    KC.check_attn(emu, CPU, None, 2, 19, 2, 16, True, clamp_case=True)
    KC.check_attn(emu, CPU, None, 2, 19, 4, 16, True, clamp_case=True)


@pytest.mark.parametrize('bsz,n,h,c', [(3, 12, 2, 64), (2, 37, 4, 256), (1, 5, 1, 16), (2, 100, 2, 64), (1, 64, 4, 32), (2, 128, 4, 1024), (1, 190, 2, 1100)])
def test_coeff(emu, bsz, n, h, c):
    KC.check_coeff(emu, CPU, None, bsz, n, h, c)


@pytest.mark.parametrize('directed', [False, True])
def test_lhat_from_edges(emu, directed):
    KC.check_lhat(emu, CPU, None, bsz=5, directed=directed)


@pytest.mark.parametrize('mode', ['cheb', 'spec'])
@pytest.mark.parametrize('share', [0, 1])
@pytest.mark.parametrize('bsz,h,dh,order,shape,n_min,n_max,seq_first', [
    (3, 2, 16, 4, 'zinc', None, None, True),
    (2, 4, 16, 4, 'mutag', None, None, False),
    (2, 2, 8, 3, 'zinc', 2, 20, True),
    (2, 1, 32, 2, 'zinc', 17, 40, True),
    (1, 2, 16, 1, 'zinc', None, None, True),
    (2, 2, 16, 5, 'pattern', 44, 70, True),
    (1, 1, 64, 2, 'zinc', 30, 40, True),          # wide shape: second node pass in spec_bwd_kernel
])
def test_filter_exact(emu, mode, share, bsz, h, dh, order, shape, n_min, n_max, seq_first):
    KC.check_filter(emu, CPU, None, mode, bsz, h, dh, order, share, shape=shape, n_min=n_min,
                    n_max=n_max, seq_first=seq_first)


def test_cheb_directed_graph(emu):
    """Lhat != Lhat^T: catches a transposed propagate in forward or backward."""
    KC.check_filter(emu, CPU, None, 'cheb', 2, 2, 16, 4, 1, directed=True)


@pytest.mark.parametrize('k_eig', [8, 16])
def test_spec_truncated(emu, k_eig):
    KC.check_filter(emu, CPU, None, 'spec', 3, 2, 16, 4, 1, k_eig=k_eig)
    KC.check_filter(emu, CPU, None, 'spec', 2, 2, 16, 4, 0, k_eig=k_eig)


@pytest.mark.parametrize('k_eig,shape,n_min,n_max,order,seq_first', [
    (16, 'zinc', None, None, 4, True),        # BASELINE config 2: N_pad <= 37, K = 16
    (8, 'mutag', None, None, 4, False),
    (32, 'pattern', 44, 64, 3, True),         # 4 row tiles, 2 eigen tiles
    (None, 'zinc', 12, 32, 5, True),          # exact operator, K = N_pad = 32
    (32, 'pattern', 100, 188, 4, True),       # large graphs: 12 row tiles, LDS-tiled U^T X
    (16, 'pattern', 70, 120, 4, False),       # 8 row tiles
])
def test_spec_one_workgroup_per_graph(emu, k_eig, shape, n_min, n_max, order, seq_first):
    """4 heads x dh 16, all heads on the graph: the LDS-staged kernels (spec_*_graph_kernel)."""
    KC.check_filter(emu, CPU, None, 'spec', 3, 4, 16, order, 1, k_eig=k_eig, shape=shape, n_min=n_min,
                    n_max=n_max, seq_first=seq_first)


@pytest.mark.parametrize('m,ki,no,relu,rowscale,residual,stats', [
    (37 * 3, 64, 192, False, False, False, False),     # in_proj
    (100, 64, 64, False, True, True, True),            # out_proj + degree + residual + BN stats
    (70, 64, 128, True, False, False, False),          # linear1 + relu
    (70, 128, 64, False, False, True, True),           # linear2 + residual + BN stats
    (33, 32, 32, False, False, False, False),
    (200, 16, 16, True, True, False, True),
])
def test_rowlin(emu, m, ki, no, relu, rowscale, residual, stats):
    KC.check_rowlin(emu, CPU, None, m, ki, no, relu, rowscale, residual, stats)


@pytest.mark.parametrize('m,d', [(111, 64), (64, 32), (300, 128), (50, 192)])
def test_batchnorm(emu, m, d):
    KC.check_bn(emu, CPU, None, m, d)


def test_batchnorm_statistics_far_from_zero(emu):
    errs = KC.check_bn_far_from_zero(emu, CPU, None, m=600, d=64)
    print(errs)


@pytest.mark.parametrize('r,c', [(19, 4096 + 64), (3, 8192), (64, 4100), (74, 4096), (300, 4096), (200, 48), (1, 16)])
def test_colsum_shapes(emu, r, c):
    KC.check_colsum(emu, CPU, None, r, c)


def test_colsum_multi_mixed_segments(emu):
    KC.check_colsum_multi_mixed(emu, CPU, None)


@pytest.mark.parametrize('r,k,n,with_dx', [(64, 64, 64, True), (36, 48, 80, True), (20, 16, 16, False), (8, 100, 36, True)])
def test_lin_gemm(emu, r, k, n, with_dx):
    """the C x C linear of the coefficient generator: forward and the one-launch backward, ragged tiles"""
    KC.check_lin(emu, CPU, None, r, k, n, with_dx=with_dx)


@pytest.mark.parametrize('bsz,n,use_pe,seq_first,clamp', [
    (3, 37, True, True, False), (2, 64, True, False, False), (4, 9, False, True, False), (2, 19, True, True, True),
])
def test_attn_bwd_one_workgroup_per_graph(emu, monkeypatch, bsz, n, use_pe, seq_first, clamp):
    """4 heads x dh 16: attn_bwd_graph_kernel (chosen by itself from 192 graphs up; forced here)"""
    monkeypatch.setenv('FETA_ATTN_BWD_GRAPH', '1')
    KC.check_attn(emu, CPU, None, bsz, n, 4, 16, use_pe, seq_first, clamp_case=clamp)


# ---- spectrum producer ----


@pytest.mark.parametrize('shape,bsz,n_min,n_max,n_pad', [
    ('zinc', 6, None, None, None),     # N <= 37, one 64-row chunk, edgeless / 1-node / 2-node graphs mixed in
    ('mutag', 5, 3, 16, 16),           # 256-thread launch (N <= 32), odd and even n
    ('pattern', 2, 66, 70, None),      # two 64-row chunks
])
def test_eigh_sym(emu, shape, bsz, n_min, n_max, n_pad):
    KC.check_eigh(emu, CPU, None, shape, bsz, 0, n_min, n_max, n_pad)


def test_eigh_sym_three_chunks(emu):
    KC.check_eigh(emu, CPU, None, 'pattern', 1, 0, 130, 130)


def test_eigh_sym_truncated_equals_full(emu):
    KC.check_eigh_truncated_equals_full(emu, CPU, None)


@pytest.mark.parametrize('bsz,n,use_pe,tie_qk,with_bn,write_attn', [
    (2, 65, True, False, False, True),
    (2, 100, False, True, True, True),
    (1, 130, True, False, True, False),
])
def test_attn_out_against_oracle(emu, bsz, n, use_pe, tie_qk, with_bn, write_attn):
    KC.check_attn_out(emu, CPU, None, bsz, n, use_pe=use_pe, tie_qk=tie_qk, with_bn=with_bn, write_attn=write_attn)


def test_attn_out_rejects_what_it_does_not_take(emu):
    """feta_attn_out_fwd: N <= 256, the residual seen through a PUBLISHED parameter block only (fresh statistics are the
    in_proj launch's to finalize), fp32 token tensors; feta_attn_block_stat_rows / feta_attn_out_stat_rows report 0 for
    shapes their kernels do not take"""
    assert emu.attn_out_supported(256, 64, 4) and not emu.attn_out_supported(257, 64, 4)
    assert not emu.attn_out_supported(100, 32, 4) and not emu.attn_out_supported(100, 64, 2)
    assert emu.attn_out_stat_rows(64, 128) == 256 and emu.attn_out_stat_rows(3, 33) == 6 and emu.attn_out_stat_rows(3, 300) == 0
    assert emu.attn_block_stat_rows(3, 65) == 0 and emu.attn_block_stat_rows(5, 20) == 5
    b, n, d, h = 2, 70, 64, 4
    m = b * n
    z = lambda *s: torch.zeros(*s)
    kw = dict(x=z(m, d), w_out=z(d, d), b_out=z(d), pe=None, n_real=torch.full((b,), n, dtype=torch.int32),
              qkv=z(m, 3 * d), out=z(m, d), attn_stats=z(b, h, n, 2), attn=None, y=z(m, d))
    emu.attn_out_fwd(b, n, 0.25, None, **kw)          # (the plain call is fine)
    with pytest.raises(ValueError, match='x_bn'):
        emu.attn_out_fwd(b, n, 0.25, None, x_stats=z(3, 2, d), **kw)
    with pytest.raises(ValueError, match='fp32'):
        emu.attn_out_fwd(b, n, 0.25, None, **dict(kw, x=z(m, d).bfloat16(), qkv=z(m, 3 * d).bfloat16(),
                                                  out=z(m, d).bfloat16(), y=z(m, d).bfloat16()))


def test_eigh_sym_rejects_large_n(emu):
    assert emu.eigh_sym_supported(256) and not emu.eigh_sym_supported(257)
    assert emu.eigh_sym_workspace_bytes(3, 192) == 0 and emu.eigh_sym_workspace_bytes(3, 222) == 4 * 3 * 222 * 260
    a = torch.zeros(1, 300, 300)
    with pytest.raises(ValueError, match='256'):
        emu.eigh_sym(a, torch.tensor([300], dtype=torch.int32), 2.0, torch.zeros(1, 300, 300), torch.zeros(1, 300),
                     None, 0, 0.0, None)


@pytest.mark.skipif(not os.environ.get('FETA_SLOW_TESTS'), reason='~1 min in the host emulation (512 fibers x ~10^3 '
                    'rotation rounds); the GPU suite runs this variant on the 222- and 256-node buckets')
def test_eigh_sym_workspace_variant(emu):
    """192 < N <= 256: the matrix lives in the caller's workspace instead of LDS (molhiv's 222-node bucket)"""
    KC.check_eigh(emu, CPU, None, 'molhiv', 1, 0, 200, 200, 200)


@pytest.mark.parametrize('kind,zero_diag,from_device', [('diffusion', False, True), ('pstep', True, True),
                                                        ('pstep', False, False)])
def test_spectral_kernel(emu, kind, zero_diag, from_device):
    KC.check_spectral_kernel(emu, CPU, None, kind, zero_diag=zero_diag, from_device_eigh=from_device)


def test_spectral_kernel_pstep_exponents(emu):
    KC.check_spectral_kernel(emu, CPU, None, 'pstep', p=1, bsz=4, from_device_eigh=False)
    KC.check_spectral_kernel(emu, CPU, None, 'pstep', p=0, bsz=4, from_device_eigh=False)   # first power as well


@pytest.mark.parametrize('m,d', [(37, 64), (5, 32), (100, 128), (16, 4), (33, 200), (1, 256)])
def test_layernorm(emu, m, d):
    KC.check_layernorm(emu, CPU, None, m, d)


def test_layernorm_rejects_unsupported_width(emu):
    y = torch.zeros(4, 6)
    with pytest.raises(ValueError):
        emu.layernorm_fwd(y, torch.ones(6), torch.zeros(6), 1e-5, torch.empty_like(y), torch.empty(4, 2), None)


# ---- fused layer-stack kernels on bf16 storage (dtype = FETA_BF16: BASELINE configs 3 / 5) --------------------------------
@pytest.mark.parametrize('kw', [dict(), dict(bsz=3, n_pad=37, n_min=9, with_pe=False),
                                dict(bsz=2, n_pad=16, n_min=1, need_attn=False), dict(bsz=2, n_pad=64, n_min=40)])
def test_attn_block_fwd_bf16(emu, kw):
    KC.check_attn_block_lp(emu, CPU, None, **kw)


@pytest.mark.parametrize('env,kw', [
    (dict(FETA_BLOCK_FWD_WAVES='4'), dict(bsz=3, n_pad=37, n_min=9)),
    (dict(FETA_BLOCK_FWD_WGS='1'), dict(bsz=2, n_pad=64, n_min=40, with_pe=False)),
    (dict(FETA_BLOCK_FWD_WGS='2', FETA_BLOCK_MAX_GRID='2'), dict(bsz=3, n_pad=37, n_min=9))])
def test_attn_block_fwd_bf16_forms(emu, monkeypatch, env, kw):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    KC.check_attn_block_lp(emu, CPU, None, **kw)


@pytest.mark.parametrize('kw', [dict(), dict(split=True), dict(bsz=3, n_pad=37, n_min=9, with_pe=False, with_bn=True),
                                dict(bsz=2, n_pad=16, n_min=1, split=True, with_bn=True), dict(bsz=2, n_pad=64, n_min=40)])
def test_attn_block_bwd_bf16(emu, kw):
    KC.check_attn_block_bwd_lp(emu, CPU, None, **kw)


@pytest.mark.parametrize('kw', [dict(), dict(m=33, ff=64, with_bn=False), dict(m=130, ff=256, seed=3)])
def test_ffn_fwd_bf16(emu, kw):
    KC.check_ffn_lp(emu, CPU, None, **kw)


@pytest.mark.parametrize('kw', [dict(), dict(m=70, ff=64, with_bn=False), dict(m=300, ff=128, seed=3)])
def test_ffn_bwd_bf16(emu, kw):
    KC.check_ffn_bwd_lp(emu, CPU, None, **kw)


@pytest.mark.parametrize('dtype_check', ['bf16'])
def test_attn_block_bwd_walks_several_graphs(emu, monkeypatch, dtype_check):
    """more graphs than workgroups (FETA_BLOCK_BWD_MAX_GRID): a workgroup adds every graph it walks to its own partial
    row and keeps the BatchNorm partial sums in registers"""
    monkeypatch.setenv('FETA_BLOCK_BWD_MAX_GRID', '2')
    KC.check_attn_block_bwd_lp(emu, CPU, None, bsz=5, n_pad=21, n_min=3, with_bn=True)
    KC.check_attn_block_bwd_lp(emu, CPU, None, bsz=4, n_pad=37, n_min=9, with_pe=False)


@pytest.mark.parametrize('r,k,n,with_dx,bf16', [(64, 256, 256, True, False), (128, 256, 512, True, True),
                                                (64, 512, 256, False, False)])
def test_lin_gemm_tiled(emu, monkeypatch, r, k, n, with_dx, bf16):
    """the LDS-tiled kernels of csrc/lin.hip (the form the BASELINE shape R = 512, K = N = 1024 takes), forced at small
    shapes: forward, dX and dW roles interleaved in one launch, fp32 row sums for db, pending column sums"""
    monkeypatch.setenv('FETA_LIN_TILED', '2')
    KC.check_lin(emu, CPU, None, r, k, n, with_dx=with_dx, bf16=bf16)


@pytest.mark.parametrize('bsz,n,h,dh,use_pe,dtype', [(3, 20, 2, 16, True, torch.float32), (2, 37, 4, 16, False, torch.float32),
                                                     (2, 70, 1, 64, True, torch.float32), (2, 33, 2, 32, True, torch.bfloat16)])
def test_attn_stab_clamp5(emu, bsz, n, h, dh, use_pe, dtype):
    """stab = clamp5 (witnesses LSPE/layers/graphit_gt_layer.py:39-43): exp(clamp(s, -5, 5)), forward and
    backward with zero gradient through clamped scores, fp32 and bf16 storage"""
    KC.check_attn(emu, CPU, None, bsz, n, h, dh, use_pe, dtype=dtype, clamp5=True)


# ---- LayerNorm on load in the fused stack kernels (ABI 9) --------------------------------------------------------------
F32, B16 = torch.float32, torch.bfloat16


@pytest.mark.parametrize('kw', [dict(m=75, ff=128, dtype=F32), dict(m=40, ff=64, dtype=B16), dict(m=33, ff=256, dtype=F32)])
def test_ffn_fwd_layernorm_on_load(emu, kw):
    KC.check_ffn_ln(emu, CPU, None, **kw)


@pytest.mark.parametrize('kw', [dict(bsz=3, n_pad=21, dtype=F32), dict(bsz=2, n_pad=37, n_min=9, dtype=B16),
                                dict(bsz=2, n_pad=50, n_min=20, dtype=F32, with_pe=False, need_attn=False)])
def test_attn_block_fwd_layernorm_on_load(emu, kw):
    KC.check_attn_block_ln(emu, CPU, None, **kw)


@pytest.mark.parametrize('env', [{}, {'FETA_BLOCK_FWD_WAVES': '4'}, {'FETA_BLOCK_MAX_GRID': '2'}])
def test_attn_block_fwd_layernorm_on_load_forms(emu, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    KC.check_attn_block_ln(emu, CPU, None, bsz=5, n_pad=33, n_min=5, dtype=F32)


@pytest.mark.parametrize('kw', [dict(m=150, ff=128, dtype=F32), dict(m=70, ff=64, dtype=B16, two_parts=True),
                                dict(m=100, ff=128, dtype=F32, two_parts=True)])
def test_ffn_bwd_layernorm_on_load(emu, kw):
    KC.check_ffn_bwd_ln(emu, CPU, None, **kw)


@pytest.mark.parametrize('kw', [dict(bsz=3, n_pad=21, dtype=F32), dict(bsz=2, n_pad=37, n_min=9, dtype=B16, split=True),
                                dict(bsz=2, n_pad=20, dtype=F32, split=True, first_layer=True),
                                dict(bsz=2, n_pad=50, n_min=20, dtype=F32, with_pe=False)])
def test_attn_block_bwd_layernorm_on_load(emu, kw):
    KC.check_attn_block_bwd_ln(emu, CPU, None, **kw)


def test_attn_block_bwd_layernorm_walks_several_graphs(emu, monkeypatch):
    monkeypatch.setenv('FETA_BLOCK_BWD_MAX_GRID', '2')
    KC.check_attn_block_bwd_ln(emu, CPU, None, bsz=5, n_pad=21, dtype=F32)


@pytest.mark.parametrize('kw', [dict(norm='bn_fresh'), dict(norm='bn_block', k_eig=8, bsz=3), dict(norm='plain', shape='mutag', k_eig=8),
                                dict(norm='bn_fresh', shape='pattern', n_min=44, n_max=64, k_eig=32, bsz=2),
                                dict(norm='plain', shape='pattern', n_min=70, n_max=100, k_eig=32, bsz=2)])
def test_spec_filter_with_linear_cat(emu, kw):
    KC.check_spec_cat(emu, CPU, None, **kw)


@pytest.mark.parametrize('kw', [dict(norm='bn_block'), dict(norm='plain', k_eig=8, bsz=3), dict(norm='bn_block', shape='mutag', k_eig=8),
                                dict(norm='bn_block', shape='pattern', n_min=44, n_max=64, k_eig=32, bsz=2),
                                dict(norm='plain', shape='molhiv', n_min=2, n_max=50, k_eig=16, bsz=4)])
def test_spec_filter_with_linear_cat_backward(emu, kw):
    KC.check_spec_cat_bwd(emu, CPU, None, **kw)
