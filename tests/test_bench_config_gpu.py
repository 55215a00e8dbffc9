"""The EXACT configuration bench.py times (BASELINE.json configs[1]: B=128, N_pad=37, d=64, 4 heads, K=16,
3 layers, BatchNorm, every head on the graph, eigenbasis filter) through the captured hipGraph, against the
fp64 oracle of the same truncated-K operator; and the reference-literal operator bench.py reports beside it."""
import contextlib

import pytest
import torch

import bench_checks as BC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode,share,two_phase', [
    (None, None, False),           # bench.py defaults: spectral, K = 16, heads_share_graph = True
    (None, None, True),            # the split backward bench.py uses for --gpus > 1
    ('cheb', False, False),        # reference_literal leg
    ('spectral', False, False),
])
def test_timed_configuration_matches_oracle(hip, mode, share, two_phase):
    errs, used_graph = BC.check_bench_step(hip[1], contextlib.nullcontext, [], filter_mode=mode, share=share,
                                           replays=3, two_phase=two_phase)
    assert used_graph
    print('max abs errors:', {k: '%.2e' % v for k, v in errs.items()})


PATTERN = ['--shape', 'pattern', '--batch', '64', '--k-eig', '32']


@pytest.mark.parametrize('argv', [
    PATTERN + ['--n-pad', '128'],                               # BASELINE config 4 as bench.py's extra_configs time it
    PATTERN + ['--n-pad', '120', '--layer-norm'],               # ... with the norm the reference defaults to
    PATTERN + ['--n-pad', '120', '--layer-norm', '--no-pe'],    # ... and pe=None (README.md:71 passes no --pos-enc)
    PATTERN + ['--n-pad', '188', '--batch', '16'],              # beyond 128 nodes: pe read from global memory in backward
    ['--shape', 'mutag', '--batch', '32', '--n-pad', '28', '--k-eig', '8', '--layer-norm', '--no-pe'],   # config 1 (README.md:49)
    ['--shape', 'molhiv', '--batch', '320', '--n-pad', '64', '--layer-norm', '--no-pe'],                # config 5's norm, fp32
    ['--shape', 'molhiv', '--batch', '320', '--n-pad', '64'],
])
def test_timed_extra_configurations_match_oracle(hip, argv):
    """Every further leg bench.py prints (extra_configs): the N > 64 encoder (in_proj -> feta_attn_out_fwd -> ffn -> ...
    -> attn_bwd_head_kernel -> large-graph coefficient generator) and the reference-default LayerNorm / pe=None legs of
    configs 1, 4 and 5, through the captured hipGraph against oracle.encoder_gengcn: output 1e-5, gradients 3e-5 relative."""
    errs, used_graph = BC.check_bench_step(hip[1], contextlib.nullcontext, argv, replays=2)
    assert used_graph
    print('max abs errors:', {k: '%.2e' % v for k, v in errs.items()})


def test_split_backward_with_captured_collectives_matches_oracle(hip):
    """bench.py --two-phase --graph-collectives: the split backward and its (here: empty, world = 1) collectives as ONE
    hipGraph per step - the same gradients as the two-replay form and the single pass (all against the oracle)"""
    errs, used_graph = BC.check_bench_step(hip[1], contextlib.nullcontext, ['--graph-collectives'], replays=3, two_phase=True)
    assert used_graph
    print('max abs errors:', {k: '%.2e' % v for k, v in errs.items()})


@pytest.mark.parametrize('argv', [[], ['--two-phase'],
                                  ['--shape', 'molhiv', '--batch', '96', '--n-pad', '64', '--k-eig', '32', '--layer-norm'],
                                  ['--shape', 'molhiv', '--batch', '300', '--n-pad', '64', '--k-eig', '32']])
def test_bf16_storage_leg_matches_oracle(hip, argv):
    """bench.py --dtype bf16 through the captured hipGraph against the fp64 oracle, bf16 tolerances
    (bench_checks.BF16_MODEL_TOL / bf16_grad_tol): the BASELINE config 3 shape, single-pass and with the split backward
    of --gpus N; a molhiv-shaped LayerNorm bucket of config 5; a batch above 256 graphs (the attention-block backward
    walks several graphs per workgroup).  Every one of them must run the fused stack's bf16 instantiations."""
    from feta_tmlr_amd import fused_stack
    calls = []
    orig_bn, orig_ln = fused_stack.FusedEncoderStackFn.apply, fused_stack.FusedLayerNormStackFn.apply
    fused_stack.FusedEncoderStackFn.apply = staticmethod(lambda *a: (calls.append(a[0].dtype), orig_bn(*a))[1])
    fused_stack.FusedLayerNormStackFn.apply = staticmethod(lambda *a: (calls.append(a[0].dtype), orig_ln(*a))[1])
    try:
        errs, used_graph = BC.check_bench_step(hip[1], contextlib.nullcontext, argv + ['--dtype', 'bf16'], replays=2)
    finally:
        fused_stack.FusedEncoderStackFn.apply, fused_stack.FusedLayerNormStackFn.apply = orig_bn, orig_ln
    assert calls and all(dt == torch.bfloat16 for dt in calls), calls
    assert used_graph
    print('max abs errors:', {k: '%.2e' % v for k, v in errs.items()})
