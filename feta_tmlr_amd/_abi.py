"""ctypes binding of the C ABI declared in include/feta_hip.h.

``bind(cdll)`` attaches argtypes/restype for every exported symbol and returns a
thin ``Abi`` wrapper whose methods take torch tensors (device pointers are taken
with ``data_ptr()``; nothing is copied).  ``feta_tmlr_amd._lib`` opens
``libfeta_hip.so`` with it; the test-suite also binds the host SIMT-emulation build
of the same sources (tools/simt) to run the kernel code without a GPU.
"""
import ctypes as C
import os

import torch

_F = C.c_void_p  # const float* / float*
_I = C.c_void_p  # const int32_t* / int64_t*
_S = C.c_void_p  # feta_stream_t

SIGNATURES = {
    'feta_version': ([], C.c_int),
    'feta_last_error': ([], C.c_char_p),
    'feta_attn_fwd': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, C.c_int64, C.c_int64,
                       _F, _F, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_attn_bwd': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, _F, _F, C.c_int64, C.c_int64,
                       _F, _F, _F, _F, _F, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _S],
                      C.c_int),
    'feta_coeff_fwd': ([_F, _I, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_coeff_bwd_groups': ([C.c_int, C.c_int], C.c_int),
    'feta_coeff_bwd': ([_F, _I, _F, _F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S],
                       C.c_int),
    'feta_colsum': ([_F, _F, C.c_int, C.c_int, _S], C.c_int),
    'feta_cheb_filter_fwd': ([_F, C.c_int64, C.c_int64, _F, _F, _F, _I, _F, C.c_int64, C.c_int64,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_cheb_filter_bwd': ([_F, C.c_int64, C.c_int64, _F, _F, _I, _F, C.c_int64, C.c_int64,
                              _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S],
                             C.c_int),
    'feta_spec_filter_fwd': ([_F, C.c_int64, C.c_int64, _F, _F, _F, _F, _I, _F, C.c_int64, C.c_int64,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S],
                             C.c_int),
    'feta_spec_filter_bwd': ([_F, C.c_int64, C.c_int64, _F, _F, _F, _I, _F, C.c_int64, C.c_int64,
                              _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, _S], C.c_int),
    'feta_rowlin_blocks': ([C.c_int], C.c_int),
    'feta_rowlin_chunks': ([C.c_int], C.c_int),
    'feta_rowlin_fwd': ([_F, _F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_rowlin_bwd': ([_F, _F, _F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_bn_stats': ([_F, _F, C.c_int, C.c_int, _S], C.c_int),
    'feta_bn_stats_shift': ([_F, _F, _F, C.c_int, C.c_int, _S], C.c_int),
    'feta_bn_apply_fwd': ([_F, _F, _F, _F, _F, _F, _F, _F, _I, C.c_float, C.c_float, C.c_int, C.c_int, _S],
                          C.c_int),
    'feta_bn_bwd': ([_F, _F, _F, _F, _F, _F, _F, _F, C.c_int, C.c_int, _S], C.c_int),
    'feta_lhat_from_edges': ([_I, C.c_int64, _I, _I, _F, _F, C.c_int, C.c_int, C.c_int64, _S],
                             C.c_int),
    'feta_layernorm_blocks': ([C.c_int], C.c_int),
    'feta_layernorm_fwd': ([_F, _F, _F, C.c_float, _F, _F, C.c_int, C.c_int, _S], C.c_int),
    'feta_layernorm_bwd': ([_F, _F, _F, _F, _F, _F, C.c_int, _F, C.c_int, C.c_int, _S], C.c_int),
    'feta_layernorm_fwd_ex': ([_F, _F, _F, C.c_float, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_layernorm_bwd_eps': ([_F, _F, _F, C.c_float, _F, _F, _F, C.c_int, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S],
                               C.c_int),
    'feta_layernorm_bwd_ex': ([_F, _F, _F, _F, _F, _F, C.c_int, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S],
                              C.c_int),
    'feta_eigh_sym_supported': ([C.c_int], C.c_int),
    'feta_eigh_sym_workspace_bytes': ([C.c_int, C.c_int], C.c_int64),
    'feta_eigh_sym': ([_F, _I, C.c_float, _F, _F, _I, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _S],
                      C.c_int),
    'feta_spectral_kernel': ([_F, _F, _I, C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, _F,
                              C.c_int, C.c_int, C.c_int, _S], C.c_int),
}

SIGNATURES.update({
    'feta_attn_fwd_bf16': SIGNATURES['feta_attn_fwd'],
    'feta_attn_bwd_bf16': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, _F, C.c_int64, C.c_int64,
                            _F, _F, _F, _F, _F, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_attn_fwd_drop': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, C.c_int64, C.c_int64, _F, _F, C.c_float,
                            C.c_float, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_attn_bwd_drop': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, _F, C.c_int64, C.c_int64, _F, _F, _F, _F, _F,
                            C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            _S], C.c_int),
    'feta_attn_fwd_drop_dev': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, C.c_int64, C.c_int64, _F, _F, C.c_float,
                                C.c_float, _I, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_attn_bwd_drop_dev': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, _F, C.c_int64, C.c_int64, _F, _F, _F, _F, _F,
                                C.c_float, C.c_float, _I, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                _S], C.c_int),
    'feta_attn_fwd_stab': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, C.c_int64, C.c_int64, _F, _F, C.c_float,
                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_attn_bwd_stab': ([_F, _F, _F, C.c_int64, C.c_int64, _F, _I, _F, _F, C.c_int64, C.c_int64, _F, _F, _F, _F, _F,
                            C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int),
    'feta_spec_filter_fwd_bf16': SIGNATURES['feta_spec_filter_fwd'],
    'feta_spec_filter_bwd_bf16': SIGNATURES['feta_spec_filter_bwd'],
})


class ColsumSeg(C.Structure):
    """struct feta_colsum_seg (include/feta_hip.h)."""
    _fields_ = [('in_', _F), ('out', _F), ('R', C.c_int), ('C', C.c_int), ('ld', C.c_int),
                ('bcast_out', _F), ('bcast_rows', C.c_int)]


SIGNATURES['feta_colsum_multi'] = ([C.POINTER(ColsumSeg), C.c_int, _S], C.c_int)
SIGNATURES['feta_lin_supported'] = ([C.c_int, C.c_int, C.c_int], C.c_int)
SIGNATURES['feta_lin_fwd'] = ([_F, _F, _F, _F, C.c_int, C.c_int, C.c_int, _S], C.c_int)
SIGNATURES['feta_lin_bwd'] = ([_F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.POINTER(ColsumSeg), C.c_int, _S],
                              C.c_int)
SIGNATURES['feta_lin_fwd_ex'] = ([_F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, _S], C.c_int)
SIGNATURES['feta_lin_bwd_ex'] = ([_F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.POINTER(ColsumSeg), C.c_int, C.c_int,
                                  _S], C.c_int)


class RowLinEx(C.Structure):
    """struct feta_rowlin_ex (include/feta_hip.h) - field order must match the header."""
    _fields_ = [
        ('x', _F), ('w', _F), ('bias', _F), ('rowscale', _F),
        ('M', C.c_int), ('KI', C.c_int), ('NO', C.c_int), ('relu', C.c_int),
        ('residual', _F), ('res_bn', _F), ('y', _F), ('stats', _F),
        ('x_bn', _F), ('x_stats', _F), ('Gx', C.c_int),
        ('x_gamma', _F), ('x_beta', _F), ('x_bn_out', _F), ('x_rmean', _F), ('x_rvar', _F), ('x_nbt', _I),
        ('momentum', C.c_float), ('eps', C.c_float),
        ('dy', _F), ('relu_y', _F), ('dx', _F), ('partial', _F), ('partial_ld', C.c_int),
        ('g_y', _F), ('g_bn', _F), ('g_sum', _F), ('Gs', C.c_int),
        ('g_fin', _F), ('g_fin_out', _F), ('dgamma', _F), ('dbeta', _F),
        ('add_plain', _F), ('add_dout', _F), ('add_y', _F), ('add_bn', _F), ('add_fin', _F),
        ('x2', _F), ('x_split', C.c_int), ('dx2', _F),
        ('sum_y', _F), ('sum_bn', _F), ('sum_out', _F), ('stats_shift', _F),
    ]


SIGNATURES.update({
    'feta_rowlin_fwd_ex': ([C.POINTER(RowLinEx), _S], C.c_int),
    'feta_rowlin_bwd_ex': ([C.POINTER(RowLinEx), _F, _S], C.c_int),
    'feta_bn_apply_fwd_prm': ([_F, _F, _F, _F, _F, _F, _F, _F, _I, C.c_float, C.c_float, C.c_int, C.c_int,
                               C.c_int, _S], C.c_int),
    'feta_bn_bwd_reduce': ([_F, _F, _F, _F, C.c_int, C.c_int, _S], C.c_int),
})



class SpecCat(C.Structure):
    """struct feta_spec_cat (include/feta_hip.h) - field order must match the header."""
    _fields_ = [
        ('y2', _F), ('y2_sb', C.c_int64), ('y2_sn', C.c_int64), ('y2_bn', _F), ('y2_stats', _F), ('Gx', C.c_int),
        ('gamma', _F), ('beta', _F), ('bn_out', _F), ('rmean', _F), ('rvar', _F), ('nbt', _I),
        ('momentum', C.c_float), ('eps', C.c_float), ('M', C.c_int), ('w_cat', _F), ('b_cat', _F), ('out', _F),
    ]


class SpecCatGrad(C.Structure):
    """struct feta_spec_cat_grad (include/feta_hip.h) - field order must match the header."""
    _fields_ = [
        ('dout', _F), ('y2', _F), ('y2_sb', C.c_int64), ('y2_sn', C.c_int64), ('y2_bn', _F), ('filt', _F), ('w_cat', _F),
        ('dxn', _F), ('gs', _F), ('partial', _F), ('partial_ld', C.c_int64),
    ]


SIGNATURES.update({
    'feta_spec_cat_bwd_supported': ([C.c_int] * 6, C.c_int),
    'feta_spec_cat_bwd_rows': ([C.c_int], C.c_int),
    'feta_spec_filter_cat_bwd': ([_F, C.c_int64, C.c_int64, _F, _F, _F, _I, C.c_int64, C.c_int64, _F, _F, _F,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SpecCatGrad), _S],
                                 C.c_int),
    'feta_spec_cat_supported': ([C.c_int] * 6, C.c_int),
    'feta_spec_filter_cat_fwd': ([_F, C.c_int64, C.c_int64, _F, _F, _F, _F, _I, _F, C.c_int64, C.c_int64,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SpecCat), _S], C.c_int),
})


class AttnBlock(C.Structure):
    """struct feta_attn_block (include/feta_hip.h) - field order must match the header."""
    _fields_ = [
        ('x', _F), ('x_bn', _F), ('x_stats', _F), ('Gx', C.c_int),
        ('x_gamma', _F), ('x_beta', _F), ('x_bn_out', _F), ('x_rmean', _F), ('x_rvar', _F), ('x_nbt', _I),
        ('momentum', C.c_float), ('eps', C.c_float),
        ('w_in', _F), ('b_in', _F), ('w_out', _F), ('b_out', _F), ('pe', _F), ('n_real', _I),
        ('rowscale', _F), ('qkv', _F), ('out', _F), ('attn_stats', _F), ('attn', _F), ('y', _F),
        ('y_stats', _F), ('scale', C.c_float), ('B', C.c_int), ('N', C.c_int), ('M', C.c_int),
        ('row_sb', C.c_int64), ('row_sn', C.c_int64), ('tie_qk', C.c_int), ('dtype', C.c_int), ('y_shift', _F),
        ('out_f32', _F), ('x_ln_gamma', _F), ('x_ln_beta', _F),
    ]


SIGNATURES.update({
    'feta_attn_block_supported': ([C.c_int, C.c_int, C.c_int], C.c_int),
    'feta_attn_block_stat_rows': ([C.c_int, C.c_int], C.c_int),
    'feta_attn_out_supported': ([C.c_int, C.c_int, C.c_int], C.c_int),
    'feta_attn_out_stat_rows': ([C.c_int, C.c_int], C.c_int),
    'feta_attn_out_fwd': ([C.POINTER(AttnBlock), _S], C.c_int),
    'feta_attn_out_fwd_sums': ([C.POINTER(AttnBlock), C.POINTER(ColsumSeg), C.c_int, _S], C.c_int),
    'feta_attn_block_fwd': ([C.POINTER(AttnBlock), _S], C.c_int),
    'feta_attn_block_fwd_sums': ([C.POINTER(AttnBlock), C.POINTER(ColsumSeg), C.c_int, _S], C.c_int),
})



class CoeffFwdRole(C.Structure):
    """struct feta_coeff_fwd_role (include/feta_hip.h)."""
    _fields_ = [('attn', _F), ('n_real', _I), ('s', _F), ('gcn_bias', _F), ('cj', _F), ('pooled', _F),
                ('B', C.c_int), ('N', C.c_int), ('H', C.c_int), ('C', C.c_int)]


class CoeffBwdRole(C.Structure):
    """struct feta_coeff_bwd_role (include/feta_hip.h)."""
    _fields_ = [('cj', _F), ('n_real', _I), ('s', _F), ('gcn_bias', _F), ('dpooled', _F), ('partial', _F),
                ('B', C.c_int), ('N', C.c_int), ('H', C.c_int), ('C', C.c_int)]


class Ffn(C.Structure):
    """struct feta_ffn (include/feta_hip.h) - field order must match the header."""
    _fields_ = [
        ('x', _F), ('x_bn', _F), ('x_stats', _F), ('Gx', C.c_int),
        ('x_gamma', _F), ('x_beta', _F), ('x_bn_out', _F), ('x_rmean', _F), ('x_rvar', _F), ('x_nbt', _I),
        ('momentum', C.c_float), ('eps', C.c_float),
        ('w1', _F), ('b1', _F), ('w2', _F), ('b2', _F), ('h', _F), ('y', _F), ('y_stats', _F),
        ('M', C.c_int), ('FF', C.c_int), ('dtype', C.c_int), ('y_shift', _F), ('y_f32', C.c_int),
        ('x_ln_gamma', _F), ('x_ln_beta', _F),
        ('y_ln_out', _F), ('y_ln_gamma', _F), ('y_ln_beta', _F), ('y_ln_eps', C.c_float), ('y_ln_f32', C.c_int),
    ]


SIGNATURES.update({
    'feta_ffn_supported': ([C.c_int, C.c_int], C.c_int),
    'feta_ffn_blocks': ([C.c_int], C.c_int),
    'feta_ffn_fwd': ([C.POINTER(Ffn), _S], C.c_int),
    'feta_ffn_fwd_coeff': ([C.POINTER(Ffn), C.POINTER(CoeffFwdRole), _S], C.c_int),
})

class AttnBlockGrad(C.Structure):
    """struct feta_attn_block_grad (include/feta_hip.h) - field order must match the header."""
    _fields_ = [
        ('dy', _F), ('y1', _F), ('bn1', _F), ('g_sum', _F), ('Gs', C.c_int), ('fin_out', _F), ('dgamma', _F),
        ('dbeta', _F), ('rowscale', _F), ('w_out', _F), ('w_in', _F), ('qkv', _F), ('out', _F), ('dout2', _F),
        ('pe', _F), ('n_real', _I), ('attn_stats', _F), ('x0', _F), ('bn0', _F), ('dx', _F), ('dx_b', _F), ('sum_out', _F),
        ('partial', _F), ('partial_ld', C.c_int), ('scale', C.c_float), ('B', C.c_int), ('N', C.c_int), ('M', C.c_int),
        ('row_sb', C.c_int64), ('row_sn', C.c_int64), ('dtype', C.c_int), ('dout2_f32', C.c_int),
        ('ln1_gamma', _F), ('x0_ln_gamma', _F), ('x0_ln_beta', _F), ('ln_eps', C.c_float),
    ]


SIGNATURES.update({
    'feta_attn_block_bwd_supported': ([C.c_int, C.c_int, C.c_int], C.c_int),
    'feta_attn_block_bwd_blocks': ([C.c_int], C.c_int),
    'feta_attn_block_bwd': ([C.POINTER(AttnBlockGrad), _S], C.c_int),
    'feta_attn_block_bwd_sums': ([C.POINTER(AttnBlockGrad), C.POINTER(ColsumSeg), C.c_int, _S], C.c_int),
})


class FfnGrad(C.Structure):
    """struct feta_ffn_grad (include/feta_hip.h) - field order must match the header."""
    _fields_ = [
        ('dy', _F), ('dy_b', _F), ('g_y', _F), ('g_bn', _F), ('g_sum', _F), ('Gs', C.c_int), ('g_fin', _F), ('g_fin_out', _F),
        ('dgamma', _F), ('dbeta', _F), ('h', _F), ('w2', _F), ('w1', _F), ('x', _F), ('x_bn', _F), ('dx', _F),
        ('sum_out', _F), ('partial', _F), ('partial_ld', C.c_int), ('M', C.c_int), ('FF', C.c_int), ('dtype', C.c_int), ('g_f32', C.c_int),
        ('g_ln_gamma', _F), ('x_ln_gamma', _F), ('x_ln_beta', _F), ('ln_eps', C.c_float),
    ]


SIGNATURES.update({
    'feta_ffn_bwd_supported': ([C.c_int, C.c_int], C.c_int),
    'feta_ffn_bwd_blocks': ([C.c_int], C.c_int),
    'feta_ffn_bwd_chunks': ([C.c_int, C.c_int], C.c_int),
    'feta_ffn_bwd': ([C.POINTER(FfnGrad), _S], C.c_int),
    'feta_ffn_bwd_coeff': ([C.POINTER(FfnGrad), C.POINTER(CoeffBwdRole), _S], C.c_int),
})

ABI_VERSION = 11


class FetaError(RuntimeError):
    pass


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _same_dtype(dtype, *tensors):
    for t in tensors:
        if t is not None and t.dtype != dtype:
            raise TypeError('expected %s operands, got %s' % (dtype, t.dtype))


def _stack_dtype(ptrs, names, f32_ok=()):
    """FETA_F32 | FETA_BF16 of a fused-stack descriptor: the token tensors `names` ([T] in include/feta_hip.h) share
    one storage type; everything else (weights, statistics, partial sums) is fp32."""
    dt = None
    for k in names:
        t = ptrs.get(k)
        if t is None or k in f32_ok and t.dtype == torch.float32:
            continue
        if dt is None:
            dt = t.dtype
        elif t.dtype != dt:
            raise TypeError('%s is %s, the other token tensors are %s' % (k, t.dtype, dt))
    if dt is None:
        dt = torch.float32
    if dt not in (torch.float32, torch.bfloat16):
        raise TypeError('token tensors must be float32 or bfloat16, got %s' % dt)
    for k, t in ptrs.items():
        if t is not None and k not in names and t.dtype == torch.bfloat16:
            raise TypeError('%s must not be bfloat16' % k)
    return 1 if dt == torch.bfloat16 else 0


def tok_strides(t):
    """(sb, sn) element strides of a token tensor given as a [B, N, H, dh] view whose
    last two dims are dense (stride dh, 1)."""
    assert t.dim() == 4 and t.stride(3) == 1 and t.stride(2) == t.shape[3], \
        'token tensor must be a [B,N,H,dh] view with dense (H,dh)'
    # the stride of a size-1 dim is arbitrary: canonicalise it (the kernels multiply it by 0)
    return (t.stride(0) if t.shape[0] > 1 else 0), (t.stride(1) if t.shape[1] > 1 else 0)


class Abi:
    def __init__(self, cdll):
        self.lib = cdll
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError if a declared symbol is missing
            fn.argtypes = argtypes
            fn.restype = restype
        v = cdll.feta_version()
        if v != ABI_VERSION:
            raise FetaError('libfeta ABI version %d, binding expects %d' % (v, ABI_VERSION))

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.feta_last_error().decode('utf-8', 'replace')
            if rc == -1:
                raise ValueError('%s: %s' % (what, msg))
            raise FetaError('%s failed (%d): %s' % (what, rc, msg))

    # q,k,v,out,...: [B,N,H,dh] views (any B/N strides)
    def attn_fwd(self, q, k, v, pe, n_real, out, attn, stats, scale, stream, drop=None, clamp5=False):
        """drop = (p, seed, offset): attention-probability dropout (feta_attn_fwd_drop); clamp5: exp(clamp(s, -5, 5))
        instead of exp(s - rowmax) (feta_attn_fwd_stab)"""
        b, n, h, dh = q.shape
        sb, sn = tok_strides(q)
        assert tok_strides(k) == (sb, sn) and tok_strides(v) == (sb, sn)
        osb, osn = tok_strides(out)
        if clamp5:
            assert drop is None or drop[0] == 0.0, 'clamp5 with attention dropout is not instantiated'
            _same_dtype(q.dtype, k, v, pe, out, attn)
            self._check(self.lib.feta_attn_fwd_stab(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real), _p(out), osb, osn,
                                                    _p(attn), _p(stats), scale, 1, 1 if q.dtype == torch.bfloat16 else 0,
                                                    b, n, h, dh, stream), 'feta_attn_fwd_stab')
            return
        if drop is not None and drop[0] > 0.0 and torch.is_tensor(drop[1]):
            # device-resident key (functional.DropoutState in device mode): drop = (p, state int64[2], offset_add)
            _same_dtype(q.dtype, k, v, pe, out, attn)
            assert drop[1].dtype == torch.int64 and drop[1].numel() == 2 and drop[1].device == q.device
            self._check(self.lib.feta_attn_fwd_drop_dev(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real), _p(out), osb, osn,
                                                        _p(attn), _p(stats), scale, float(drop[0]), _p(drop[1]),
                                                        int(drop[2]), 1 if q.dtype == torch.bfloat16 else 0, b, n, h, dh,
                                                        stream), 'feta_attn_fwd_drop_dev')
            return
        if drop is not None and drop[0] > 0.0:
            _same_dtype(q.dtype, k, v, pe, out, attn)
            self._check(self.lib.feta_attn_fwd_drop(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real), _p(out), osb, osn,
                                                    _p(attn), _p(stats), scale, float(drop[0]), int(drop[1]), int(drop[2]),
                                                    1 if q.dtype == torch.bfloat16 else 0, b, n, h, dh, stream),
                        'feta_attn_fwd_drop')
            return
        if q.dtype == torch.bfloat16:
            _same_dtype(torch.bfloat16, k, v, pe, out, attn)
            self._check(self.lib.feta_attn_fwd_bf16(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real),
                                                    _p(out), osb, osn, _p(attn), _p(stats), scale,
                                                    b, n, h, dh, stream), 'feta_attn_fwd_bf16')
            return
        self._check(self.lib.feta_attn_fwd(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real),
                                           _p(out), osb, osn, _p(attn), _p(stats), scale,
                                           b, n, h, dh, stream), 'feta_attn_fwd')

    @staticmethod
    def attn_bwd_takes_dout2(n, dh, dtype=torch.float32, heads=None):
        """a second gradient into out_each_head added inside the kernel's loads: the batched-load kernels (N <= 64,
        dh <= 16) and the per-(graph, head) kernel (4 heads x 16, N <= 256)"""
        if dtype != torch.float32:
            return False
        head_kernel = os.environ.get('FETA_ATTN_BWD_HEAD', '1') != '0'     # (csrc/attn.hip: try_bwd_head)
        return (n <= 64 and dh <= 16) or (head_kernel and n <= 256 and dh == 16 and heads == 4)

    def attn_bwd(self, q, k, v, pe, n_real, out, dout, stats, delta, dq, dk, dv, scale, stream, dout2=None,
                 drop=None, clamp5=False):
        b, n, h, dh = q.shape
        sb, sn = tok_strides(q)
        for t in (k, v, dq, dk, dv):
            assert tok_strides(t) == (sb, sn)
        osb, osn = tok_strides(out)
        assert tok_strides(dout) == (osb, osn)
        if clamp5:
            assert dout2 is None and (drop is None or drop[0] == 0.0)
            _same_dtype(q.dtype, k, v, pe, out, dout, dq, dk, dv)
            self._check(self.lib.feta_attn_bwd_stab(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real), _p(out), _p(dout),
                                                    osb, osn, _p(stats), _p(delta), _p(dq), _p(dk), _p(dv), scale, 1,
                                                    1 if q.dtype == torch.bfloat16 else 0, b, n, h, dh, stream),
                        'feta_attn_bwd_stab')
            return
        if drop is not None and drop[0] > 0.0 and torch.is_tensor(drop[1]):
            _same_dtype(q.dtype, k, v, pe, out, dout, dq, dk, dv)
            assert dout2 is None
            self._check(self.lib.feta_attn_bwd_drop_dev(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real), _p(out), _p(dout),
                                                        osb, osn, _p(stats), _p(delta), _p(dq), _p(dk), _p(dv), scale,
                                                        float(drop[0]), _p(drop[1]), int(drop[2]),
                                                        1 if q.dtype == torch.bfloat16 else 0, b, n, h, dh, stream),
                        'feta_attn_bwd_drop_dev')
            return
        if drop is not None and drop[0] > 0.0:
            _same_dtype(q.dtype, k, v, pe, out, dout, dq, dk, dv)
            assert dout2 is None
            self._check(self.lib.feta_attn_bwd_drop(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real), _p(out), _p(dout),
                                                    osb, osn, _p(stats), _p(delta), _p(dq), _p(dk), _p(dv), scale,
                                                    float(drop[0]), int(drop[1]), int(drop[2]),
                                                    1 if q.dtype == torch.bfloat16 else 0, b, n, h, dh, stream),
                        'feta_attn_bwd_drop')
            return
        if q.dtype == torch.bfloat16:
            _same_dtype(torch.bfloat16, k, v, pe, out, dout, dq, dk, dv)
            assert dout2 is None
            self._check(self.lib.feta_attn_bwd_bf16(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real),
                                                    _p(out), _p(dout), osb, osn, _p(stats), _p(delta),
                                                    _p(dq), _p(dk), _p(dv), scale, b, n, h, dh, stream),
                        'feta_attn_bwd_bf16')
            return
        self._check(self.lib.feta_attn_bwd(_p(q), _p(k), _p(v), sb, sn, _p(pe), _p(n_real),
                                           _p(out), _p(dout), _p(dout2), osb, osn, _p(stats), _p(delta),
                                           _p(dq), _p(dk), _p(dv), scale, b, n, h, dh, stream),
                    'feta_attn_bwd')

    def coeff_fwd(self, attn, n_real, s, gcn_bias, cj, pooled, stream):
        b, h, n, _ = attn.shape
        c = s.shape[0]
        self._check(self.lib.feta_coeff_fwd(_p(attn), _p(n_real), _p(s), _p(gcn_bias), _p(cj),
                                            _p(pooled), b, n, h, c, stream), 'feta_coeff_fwd')

    def coeff_bwd_groups(self, b, h):
        return self.lib.feta_coeff_bwd_groups(b, h)

    def coeff_bwd(self, cj, n_real, s, gcn_bias, dpooled, partial, ds, dbias, b, n, h, stream, dw_dense=None):
        c = s.shape[0]
        self._check(self.lib.feta_coeff_bwd(_p(cj), _p(n_real), _p(s), _p(gcn_bias), _p(dpooled),
                                            _p(partial), _p(ds), _p(dbias), _p(dw_dense),
                                            0 if dw_dense is None else dw_dense.shape[0], b, n, h, c, stream),
                    'feta_coeff_bwd')

    @staticmethod
    def _colsum_segs(pairs):
        """pairs: [(in [R, C] (row stride >= C), out [C]) or (in, out, bcast [rows, C])] -> feta_colsum_seg array"""
        segs = (ColsumSeg * max(len(pairs), 1))()
        for sg, pr in zip(segs, pairs):
            x, out = pr[0], pr[1]
            assert x.stride(1) == 1
            sg.in_, sg.out, sg.R, sg.C, sg.ld = x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.stride(0)
            if len(pr) > 2 and pr[2] is not None:
                sg.bcast_out, sg.bcast_rows = pr[2].data_ptr(), pr[2].shape[0]
        return segs

    def colsum_multi(self, pairs, stream):
        """pairs: [(in [R, C], out [C]) or (in, out, bcast [rows, C])] - all reduced by one launch."""
        self._check(self.lib.feta_colsum_multi(self._colsum_segs(pairs), len(pairs), stream), 'feta_colsum_multi')

    def lin_supported(self, r, k, n):
        return bool(self.lib.feta_lin_supported(r, k, n))

    def lin_fwd(self, x, w, bias, y, stream, bf16=False):
        """bf16: operands rounded to bf16 when staged, bf16 MFMA, fp32 accumulate (feta_lin_fwd_ex)"""
        r, k = x.shape
        self._check(self.lib.feta_lin_fwd_ex(_p(x), _p(w), _p(bias), _p(y), r, k, w.shape[0], int(bool(bf16)), stream),
                    'feta_lin_fwd')

    def lin_bwd(self, x, w, dy, dx, dw, db, stream, pairs=(), bf16=False):
        """pairs: [(in [R, C], out [C])] pending column sums that ride in trailing workgroups of the launch"""
        r, k = x.shape
        self._check(self.lib.feta_lin_bwd_ex(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), r, k, w.shape[0],
                                             self._colsum_segs(pairs), len(pairs), int(bool(bf16)), stream),
                    'feta_lin_bwd')

    def colsum(self, x, out, stream):
        r, c = x.shape
        self._check(self.lib.feta_colsum(_p(x), _p(out), r, c, stream), 'feta_colsum')

    def cheb_filter_fwd(self, x, lhat, coeff, bias, n_real, y, order, share, stream):
        b, n, h, dh = x.shape
        xsb, xsn = tok_strides(x)
        ysb, ysn = tok_strides(y)
        self._check(self.lib.feta_cheb_filter_fwd(_p(x), xsb, xsn, _p(lhat), _p(coeff), _p(bias),
                                                  _p(n_real), _p(y), ysb, ysn, b, n, h, dh, order,
                                                  int(share), stream), 'feta_cheb_filter_fwd')

    def cheb_filter_bwd(self, x, lhat, coeff, n_real, dy, dx, dcoeff, dbias_part, order, share,
                        stream):
        b, n, h, dh = x.shape
        xsb, xsn = tok_strides(x)
        assert tok_strides(dx) == (xsb, xsn)
        ysb, ysn = tok_strides(dy)
        self._check(self.lib.feta_cheb_filter_bwd(_p(x), xsb, xsn, _p(lhat), _p(coeff), _p(n_real),
                                                  _p(dy), ysb, ysn, _p(dx), _p(dcoeff),
                                                  _p(dbias_part), b, n, h, dh, order, int(share),
                                                  stream), 'feta_cheb_filter_bwd')

    def spec_filter_fwd(self, x, u, lam, coeff, bias, n_real, y, order, share, stream):
        b, n, h, dh = x.shape
        k = u.shape[2]
        xsb, xsn = tok_strides(x)
        ysb, ysn = tok_strides(y)
        fn, nm = self.lib.feta_spec_filter_fwd, 'feta_spec_filter_fwd'
        if x.dtype == torch.bfloat16:
            _same_dtype(torch.bfloat16, u, coeff, y)
            fn, nm = self.lib.feta_spec_filter_fwd_bf16, 'feta_spec_filter_fwd_bf16'
        self._check(fn(_p(x), xsb, xsn, _p(u), _p(lam), _p(coeff), _p(bias), _p(n_real), _p(y), ysb, ysn,
                       b, n, h, dh, order, k, int(share), stream), nm)

    def spec_cat_supported(self, n, h, dh, order, k, share):
        return bool(self.lib.feta_spec_cat_supported(n, h, dh, order, k, int(share)))

    def spec_filter_cat_fwd(self, x, u, lam, coeff, bias, n_real, y, order, share, stream, y2, w_cat, b_cat, out,
                            y2_bn=None, y2_stats=None, Gx=0, gamma=None, beta=None, bn_out=None, rmean=None, rvar=None,
                            nbt=None, momentum=0.1, eps=1e-5):
        """feta_spec_filter_cat_fwd: the eigenbasis filter with linear_cat folded in.  y2 [N, B, 64] (or [B, N, 64]
        batch-first like x) = the stack output, out like y."""
        b, n, h, dh = x.shape
        k = u.shape[2]
        xsb, xsn = tok_strides(x)
        ysb, ysn = tok_strides(y)
        assert tok_strides(out) == (ysb, ysn)
        c = SpecCat()
        c.y2_sb, c.y2_sn = tok_strides(y2)
        c.Gx, c.momentum, c.eps, c.M = Gx, momentum, eps, b * n
        for name, t in (('y2', y2), ('y2_bn', y2_bn), ('y2_stats', y2_stats), ('gamma', gamma), ('beta', beta),
                        ('bn_out', bn_out), ('rmean', rmean), ('rvar', rvar), ('nbt', nbt), ('w_cat', w_cat),
                        ('b_cat', b_cat), ('out', out)):
            if t is not None:
                setattr(c, name, t.data_ptr())
        self._check(self.lib.feta_spec_filter_cat_fwd(_p(x), xsb, xsn, _p(u), _p(lam), _p(coeff), _p(bias), _p(n_real), _p(y),
                                                      ysb, ysn, b, n, h, dh, order, k, int(share), C.byref(c), stream),
                    'feta_spec_filter_cat_fwd')

    def spec_cat_bwd_supported(self, n, h, dh, order, k, share):
        return bool(self.lib.feta_spec_cat_bwd_supported(n, h, dh, order, k, int(share)))

    def spec_cat_bwd_rows(self, b):
        return int(self.lib.feta_spec_cat_bwd_rows(b))

    def spec_filter_cat_bwd(self, x, u, lam, coeff, n_real, filt, dx, dcoeff, dbias_part, order, share, stream, dout, y2,
                            w_cat, dxn, partial, y2_bn=None, gs=None):
        """feta_spec_filter_cat_bwd: the eigenbasis filter's backward with the backward of linear_cat folded in.
        x / filt / dx: [B, N, H, dh] token views; dout / y2 / dxn: [B, N, H, dh] views of [N, B, 64] row tensors (same
        strides); partial [B, >= 64 * 128 + 64]; gs [B, 2, 64] with y2_bn."""
        b, n, h, dh = x.shape
        k = u.shape[2]
        xsb, xsn = tok_strides(x)
        assert tok_strides(dx) == (xsb, xsn)
        ysb, ysn = tok_strides(filt)
        c = SpecCatGrad()
        c.y2_sb, c.y2_sn = tok_strides(y2)
        assert tok_strides(dout) == (c.y2_sb, c.y2_sn) and tok_strides(dxn) == (c.y2_sb, c.y2_sn)
        assert partial.shape[0] == self.spec_cat_bwd_rows(b) and partial.stride(1) == 1
        assert gs is None or gs.shape[0] == partial.shape[0]
        c.partial_ld = partial.stride(0)
        for name, t in (('dout', dout), ('y2', y2), ('y2_bn', y2_bn), ('filt', filt), ('w_cat', w_cat), ('dxn', dxn),
                        ('gs', gs), ('partial', partial)):
            if t is not None:
                setattr(c, name, t.data_ptr())
        self._check(self.lib.feta_spec_filter_cat_bwd(_p(x), xsb, xsn, _p(u), _p(lam), _p(coeff), _p(n_real), ysb, ysn,
                                                      _p(dx), _p(dcoeff), _p(dbias_part), b, n, h, dh, order, k, int(share),
                                                      C.byref(c), stream), 'feta_spec_filter_cat_bwd')

    def spec_filter_bwd(self, x, u, lam, coeff, n_real, dy, dx, dcoeff, dbias_part, order, share,
                        stream):
        b, n, h, dh = x.shape
        k = u.shape[2]
        xsb, xsn = tok_strides(x)
        assert tok_strides(dx) == (xsb, xsn)
        ysb, ysn = tok_strides(dy)
        fn, nm = self.lib.feta_spec_filter_bwd, 'feta_spec_filter_bwd'
        if x.dtype == torch.bfloat16:
            _same_dtype(torch.bfloat16, u, coeff, dy, dx, dcoeff)
            fn, nm = self.lib.feta_spec_filter_bwd_bf16, 'feta_spec_filter_bwd_bf16'
        self._check(fn(_p(x), xsb, xsn, _p(u), _p(lam), _p(coeff), _p(n_real), _p(dy), ysb, ysn, _p(dx),
                       _p(dcoeff), _p(dbias_part), b, n, h, dh, order, k, int(share), stream), nm)

    ROWLIN_DIMS = (16, 32, 64, 128, 192, 256)

    def rowlin_blocks(self, m):
        return self.lib.feta_rowlin_blocks(m)

    def rowlin_chunks(self, m):
        return self.lib.feta_rowlin_chunks(m)

    def rowlin_fwd(self, x, w, bias, rowscale, residual, y, stats, relu, stream):
        m, ki = x.shape
        no = w.shape[0]
        self._check(self.lib.feta_rowlin_fwd(_p(x), _p(w), _p(bias), _p(rowscale), _p(residual), _p(y),
                                             _p(stats), int(relu), m, ki, no, stream), 'feta_rowlin_fwd')

    def rowlin_bwd(self, x, w, dy, rowscale, ysaved, dx, partial, dwdb, stream):
        m, ki = x.shape
        no = w.shape[0]
        self._check(self.lib.feta_rowlin_bwd(_p(x), _p(w), _p(dy), _p(rowscale), _p(ysaved), _p(dx),
                                             _p(partial), _p(dwdb), m, ki, no, stream), 'feta_rowlin_bwd')

    def bn_stats(self, y, stats, stream, shift=None):
        """stats [rowlin_blocks(M) + 1, 2, D]: shifted partial sums + the shift row (include/feta_hip.h)"""
        m, d = y.shape
        assert stats.shape[0] == self.rowlin_blocks(m) + 1, 'stats needs rowlin_blocks(M) + 1 rows (shift row)'
        self._check(self.lib.feta_bn_stats_shift(_p(y), _p(shift), _p(stats), m, d, stream), 'feta_bn_stats')

    def bn_apply_fwd(self, y, stats, gamma, beta, out, mean_rstd, running_mean, running_var, momentum,
                     eps, stream, nbt=None):
        m, d = y.shape
        self._check(self.lib.feta_bn_apply_fwd(_p(y), _p(stats), _p(gamma), _p(beta), _p(out),
                                               _p(mean_rstd), _p(running_mean), _p(running_var), _p(nbt),
                                               momentum, eps, m, d, stream), 'feta_bn_apply_fwd')

    def bn_bwd(self, y, dout, mean_rstd, gamma, partial, dy, dgamma, dbeta, stream):
        m, d = y.shape
        self._check(self.lib.feta_bn_bwd(_p(y), _p(dout), _p(mean_rstd), _p(gamma), _p(partial), _p(dy),
                                         _p(dgamma), _p(dbeta), m, d, stream), 'feta_bn_bwd')

    def rowlin_ex(self, m, ki, no, relu=False, momentum=0.1, eps=1e-5, Gx=0, Gs=0, partial_ld=0,
                  partial_ptr=None, x_split=0, **ptrs):
        """Builds a feta_rowlin_ex descriptor; tensor-valued keyword arguments become pointers."""
        d = RowLinEx()
        d.M, d.KI, d.NO, d.relu = m, ki, no, int(relu)
        d.momentum, d.eps, d.Gx, d.Gs, d.partial_ld = momentum, eps, Gx, Gs, partial_ld
        d.x_split = x_split
        if partial_ptr is not None:
            d.partial = partial_ptr
        for k, t in ptrs.items():
            if t is not None:
                setattr(d, k, t.data_ptr())
        return d

    def rowlin_fwd_ex(self, desc, stream):
        self._check(self.lib.feta_rowlin_fwd_ex(C.byref(desc), stream), 'feta_rowlin_fwd_ex')

    def rowlin_bwd_ex(self, desc, dwdb, stream):
        self._check(self.lib.feta_rowlin_bwd_ex(C.byref(desc), _p(dwdb), stream), 'feta_rowlin_bwd_ex')

    def attn_block_supported(self, n, d_model, heads):
        return bool(self.lib.feta_attn_block_supported(n, d_model, heads))

    def attn_block_stat_rows(self, b, n):
        """partial rows a forward launch writes into y_stats (+ 1 shift row behind them)"""
        return int(self.lib.feta_attn_block_stat_rows(b, n))

    def attn_block_fwd(self, b, n, scale, stream, seq_first=True, momentum=0.1, eps=1e-5, Gx=0, tie_qk=False,
                       sums=(), **ptrs):
        """feta_attn_block_fwd; tensor-valued keyword arguments become the descriptor's pointers."""
        self.attn_block_launch(self.attn_block_desc(b, n, scale, seq_first, momentum, eps, Gx, tie_qk, **ptrs),
                               stream, sums)

    def attn_block_desc(self, b, n, scale, seq_first=True, momentum=0.1, eps=1e-5, Gx=0, tie_qk=False, **ptrs):
        d = AttnBlock()
        d.B, d.N, d.M, d.scale = b, n, b * n, scale
        d.row_sb, d.row_sn = (1, b) if seq_first else (n, 1)
        d.momentum, d.eps, d.Gx, d.tie_qk = momentum, eps, Gx, int(tie_qk)
        d.dtype = _stack_dtype(ptrs, ('x', 'pe', 'qkv', 'out', 'y'))
        for k, t in ptrs.items():
            if t is not None:
                setattr(d, k, t.data_ptr())
        return d

    def attn_block_launch(self, desc, stream, sums=()):
        """sums: [(in [R, C], out [C])] column sums that ride in trailing workgroups of the launch"""
        if sums:
            self._check(self.lib.feta_attn_block_fwd_sums(C.byref(desc), self._colsum_segs(sums), len(sums), stream),
                        'feta_attn_block_fwd_sums')
        else:
            self._check(self.lib.feta_attn_block_fwd(C.byref(desc), stream), 'feta_attn_block_fwd')

    def attn_out_supported(self, n, d_model, heads):
        return bool(self.lib.feta_attn_out_supported(n, d_model, heads))

    def attn_out_stat_rows(self, b, n):
        return int(self.lib.feta_attn_out_stat_rows(b, n))

    def attn_out_fwd(self, b, n, scale, stream, seq_first=True, tie_qk=False, sums=(), **ptrs):
        """feta_attn_out_fwd (attention core + out_proj behind the in_proj launch, N <= 256): the feta_attn_block
        descriptor with qkv as input and x as the residual; sums: [(in [R, C], out [C])] column sums in trailing workgroups"""
        d = self.attn_block_desc(b, n, scale, seq_first, tie_qk=tie_qk, **ptrs)
        if sums:
            self._check(self.lib.feta_attn_out_fwd_sums(C.byref(d), self._colsum_segs(sums), len(sums), stream),
                        'feta_attn_out_fwd_sums')
        else:
            self._check(self.lib.feta_attn_out_fwd(C.byref(d), stream), 'feta_attn_out_fwd')

    def attn_block_bwd_supported(self, n, d_model, heads):
        return bool(self.lib.feta_attn_block_bwd_supported(n, d_model, heads))

    def attn_block_bwd_blocks(self, b):
        return int(self.lib.feta_attn_block_bwd_blocks(b))

    def attn_block_bwd(self, b, n, scale, stream, seq_first=True, Gs=0, partial_ld=0, partial_ptr=None, sums=(), ln_eps=1e-5,
                       **ptrs):
        """feta_attn_block_bwd; tensor-valued keyword arguments become the descriptor's pointers.  sums: [(in [R, C],
        out [C])] column sums that ride in trailing workgroups of the launch (feta_attn_block_bwd_sums)"""
        d = AttnBlockGrad()
        d.B, d.N, d.M, d.scale, d.Gs, d.partial_ld, d.ln_eps = b, n, b * n, scale, Gs, partial_ld, ln_eps
        d.row_sb, d.row_sn = (1, b) if seq_first else (n, 1)
        if partial_ptr is not None:
            d.partial = partial_ptr
        d.dtype = _stack_dtype(ptrs, ('dy', 'y1', 'qkv', 'out', 'dout2', 'pe', 'x0', 'dx', 'dx_b'), f32_ok=('dout2',))
        d.dout2_f32 = int(ptrs.get('dout2') is not None and ptrs['dout2'].dtype == torch.float32)
        for k, t in ptrs.items():
            if t is not None:
                setattr(d, k, t.data_ptr())
        if sums:
            self._check(self.lib.feta_attn_block_bwd_sums(C.byref(d), self._colsum_segs(sums), len(sums), stream),
                        'feta_attn_block_bwd_sums')
        else:
            self._check(self.lib.feta_attn_block_bwd(C.byref(d), stream), 'feta_attn_block_bwd')

    def ffn_supported(self, d_model, ff):
        return bool(self.lib.feta_ffn_supported(d_model, ff))

    def ffn_blocks(self, m):
        return int(self.lib.feta_ffn_blocks(m))

    def ffn_fwd(self, m, ff, stream, momentum=0.1, eps=1e-5, Gx=0, coeff=None, y_ln_eps=1e-5, **ptrs):
        """feta_ffn_fwd; tensor-valued keyword arguments become the descriptor's pointers.  coeff (optional): the
        arguments of coeff_fwd as a tuple - the coefficient generator's forward rides in trailing workgroups."""
        self.ffn_launch(self.ffn_desc(m, ff, momentum, eps, Gx, y_ln_eps=y_ln_eps, **ptrs), stream, coeff)

    def ffn_desc(self, m, ff, momentum=0.1, eps=1e-5, Gx=0, y_ln_eps=1e-5, **ptrs):
        d = Ffn()
        d.M, d.FF, d.momentum, d.eps, d.Gx, d.y_ln_eps = m, ff, momentum, eps, Gx, y_ln_eps
        d.dtype = _stack_dtype(ptrs, ('x', 'h', 'y', 'y_ln_out'), f32_ok=('y', 'y_ln_out'))
        d.y_f32 = int(ptrs['y'].dtype == torch.float32)
        if ptrs.get('y_ln_out') is not None:
            d.y_ln_f32 = int(ptrs['y_ln_out'].dtype == torch.float32)
        for k, t in ptrs.items():
            if t is not None:
                setattr(d, k, t.data_ptr())
        return d

    def ffn_launch(self, desc, stream, coeff=None):
        if coeff is None:
            self._check(self.lib.feta_ffn_fwd(C.byref(desc), stream), 'feta_ffn_fwd')
            return
        attn, n_real, s, gcn_bias, cj, pooled = coeff
        b, h, n, _ = attn.shape
        r = CoeffFwdRole()
        r.attn, r.n_real, r.s, r.gcn_bias, r.cj, r.pooled = (t.data_ptr() for t in (attn, n_real, s, gcn_bias, cj, pooled))
        r.B, r.N, r.H, r.C = b, n, h, s.shape[0]
        self._check(self.lib.feta_ffn_fwd_coeff(C.byref(desc), C.byref(r), stream), 'feta_ffn_fwd_coeff')

    def ffn_bwd_supported(self, d_model, ff):
        return bool(self.lib.feta_ffn_bwd_supported(d_model, ff))

    def ffn_bwd_blocks(self, m):
        return int(self.lib.feta_ffn_bwd_blocks(m))

    def ffn_bwd_chunks(self, m, ff):
        return int(self.lib.feta_ffn_bwd_chunks(m, ff))

    def ffn_bwd_desc(self, m, ff, Gs=0, partial_ld=0, partial_ptr=None, ln_eps=1e-5, **ptrs):
        d = FfnGrad()
        d.M, d.FF, d.Gs, d.partial_ld, d.ln_eps = m, ff, Gs, partial_ld, ln_eps
        if partial_ptr is not None:
            d.partial = partial_ptr
        d.dtype = _stack_dtype(ptrs, ('dy', 'dy_b', 'g_y', 'h', 'x', 'dx'), f32_ok=('dy', 'g_y'))
        d.g_f32 = int(ptrs['dy'].dtype == torch.float32)
        if ptrs.get('g_y') is not None and ptrs['g_y'].dtype != ptrs['dy'].dtype:
            raise TypeError('dy and g_y must share a dtype')
        for k, t in ptrs.items():
            if t is not None:
                setattr(d, k, t.data_ptr())
        return d

    def ffn_bwd_launch(self, desc, stream, coeff=None):
        if coeff is None:
            self._check(self.lib.feta_ffn_bwd(C.byref(desc), stream), 'feta_ffn_bwd')
            return
        cj, n_real, s, gcn_bias, dpooled, partial, b, n, h = coeff
        r = CoeffBwdRole()
        r.cj, r.n_real, r.s, r.gcn_bias, r.dpooled, r.partial = (t.data_ptr() for t in
                                                                (cj, n_real, s, gcn_bias, dpooled, partial))
        r.B, r.N, r.H, r.C = b, n, h, s.shape[0]
        self._check(self.lib.feta_ffn_bwd_coeff(C.byref(desc), C.byref(r), stream), 'feta_ffn_bwd_coeff')

    def ffn_bwd(self, m, ff, stream, coeff=None, **kw):
        """feta_ffn_bwd; tensor-valued keyword arguments become the descriptor's pointers.  coeff (optional): the
        arguments of coeff_bwd as a tuple - the coefficient generator's backward kernel rides in trailing workgroups."""
        self.ffn_bwd_launch(self.ffn_bwd_desc(m, ff, **kw), stream, coeff)

    def bn_apply_fwd_prm(self, y, stats, gamma, beta, out, bn_prm, running_mean, running_var, momentum,
                         eps, stream, nbt=None):
        m, d = y.shape
        self._check(self.lib.feta_bn_apply_fwd_prm(_p(y), _p(stats), _p(gamma), _p(beta), _p(out),
                                                   _p(bn_prm), _p(running_mean), _p(running_var), _p(nbt),
                                                   momentum, eps, m, d, stats.shape[0] - 1, stream),   # (last row: shift)
                    'feta_bn_apply_fwd_prm')

    def bn_bwd_reduce(self, y, dout, bn_prm, partial, stream):
        m, d = y.shape
        self._check(self.lib.feta_bn_bwd_reduce(_p(y), _p(dout), _p(bn_prm), _p(partial), m, d, stream),
                    'feta_bn_bwd_reduce')

    def lhat_from_edges(self, edge_index, node_graph, node_off, deg, lhat, stream):
        b, n, _ = lhat.shape
        self._check(self.lib.feta_lhat_from_edges(_p(edge_index), edge_index.shape[1],
                                                  _p(node_graph), _p(node_off), _p(deg), _p(lhat),
                                                  b, n, node_graph.shape[0], stream),
                    'feta_lhat_from_edges')

    def layernorm_blocks(self, m):
        return self.lib.feta_layernorm_blocks(m)

    @staticmethod
    def _dt(t):
        if t.dtype == torch.float32:
            return 0
        if t.dtype == torch.bfloat16:
            return 1
        raise TypeError('float32 or bfloat16 rows, got %s' % t.dtype)

    def layernorm_fwd(self, y, gamma, beta, eps, out, stats, stream):
        """y, out: float32 or bfloat16 rows (independently: feta_layernorm_fwd_ex)"""
        m, d = y.shape
        self._check(self.lib.feta_layernorm_fwd_ex(_p(y), _p(gamma), _p(beta), eps, _p(out), _p(stats), m, d,
                                                   self._dt(y), self._dt(out), stream), 'feta_layernorm_fwd')

    def layernorm_bwd(self, dout, y, stats, gamma, dy, partial, dgdb, stream, partial_ld=0, partial_ptr=None, eps=1e-5):
        """partial_ptr / partial_ld: this LayerNorm's columns inside a shared [blocks, total] partial buffer
        (dgdb None: the caller reduces it).  dout, y, dy: float32 or bfloat16 rows, independently.  stats None: the
        forward saved none (LayerNorm on load) - recomputed from y with eps."""
        m, d = y.shape
        pp = _p(partial) if partial_ptr is None else C.c_void_p(partial_ptr)
        self._check(self.lib.feta_layernorm_bwd_eps(_p(dout), _p(y), _p(stats), eps, _p(gamma), _p(dy), pp, partial_ld,
                                                    _p(dgdb), m, d, self._dt(dout), self._dt(y), self._dt(dy), stream),
                    'feta_layernorm_bwd')

    def eigh_sym_supported(self, n):
        return bool(self.lib.feta_eigh_sym_supported(n))

    def eigh_sym_workspace_bytes(self, b, n):
        return int(self.lib.feta_eigh_sym_workspace_bytes(b, n))

    def eigh_sym(self, a, n_real, shift, u, lam, sweeps, max_sweeps, tol, stream, workspace=None):
        b, n, k = u.shape
        if workspace is None and self.eigh_sym_workspace_bytes(b, n) > 0:   # convenience for callers / tests
            workspace = torch.empty(self.eigh_sym_workspace_bytes(b, n) // 4, dtype=torch.float32, device=a.device)
        self._check(self.lib.feta_eigh_sym(_p(a), _p(n_real), shift, _p(u), _p(lam), _p(sweeps), _p(workspace),
                                           b, n, k, max_sweeps, tol, stream), 'feta_eigh_sym')

    def spectral_kernel(self, u, lam, n_real, mode, beta, p, lam_offset, zero_diag, out, stream):
        b, n, k = u.shape
        self._check(self.lib.feta_spectral_kernel(_p(u), _p(lam), _p(n_real), mode, beta, p, lam_offset,
                                                  int(zero_diag), _p(out), b, n, k, stream),
                    'feta_spectral_kernel')


def bind(cdll):
    return Abi(cdll)
