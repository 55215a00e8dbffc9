"""What bench.py times, at small shapes, on the host emulation of the kernels: truncated-K eigenbasis
operator (the timed default), the reference-literal operator, the split backward; and the self-launcher
(`python bench.py --gpus 2` with no external torchrun) with gloo on the CPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

import bench_checks as BC
from feta_tmlr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = torch.device('cpu')
SMALL = ['--batch', '5', '--n-pad', '14', '--k-eig', '6', '--layers', '2', '--no-graph']


@pytest.mark.parametrize('mode,share,two_phase', [('spectral', True, False), ('cheb', False, False),
                                                  ('spectral', False, True)])
def test_bench_step_matches_oracle(emu, mode, share, two_phase):
    BC.check_bench_step(CPU, lambda: _lib.override_for_tests(emu), SMALL, filter_mode=mode, share=share,
                        replays=1, two_phase=two_phase)


@pytest.mark.parametrize('argv', [
    ['--shape', 'pattern', '--batch', '3', '--n-pad', '70', '--k-eig', '8', '--layers', '2', '--no-graph'],
    ['--shape', 'pattern', '--batch', '2', '--n-pad', '66', '--k-eig', '8', '--layers', '2', '--no-graph', '--layer-norm', '--no-pe'],
    SMALL + ['--layer-norm', '--no-pe'],
    SMALL + ['--shape', 'mutag', '--layer-norm'],
])
def test_bench_extra_configurations_match_oracle(emu, argv):
    """the further legs of the bench line (extra_configs) at small shapes: graphs beyond 64 nodes (in_proj ->
    feta_attn_out_fwd -> ... -> attn_bwd_head_kernel), LayerNorm stacks, pe=None"""
    BC.check_bench_step(CPU, lambda: _lib.override_for_tests(emu), argv, replays=1)


@pytest.mark.parametrize('share,extra', [(True, []), (False, []), (True, ['--layer-norm']), (True, ['--two-phase'])])
def test_bench_step_bf16_matches_oracle(emu, share, extra):
    """--dtype bf16 (BASELINE configs 3 / 5): bf16 storage, bf16 MFMA, fp32 statistics and master weights - the fused
    stack's bf16 instantiations (BatchNorm and LayerNorm stacks, single-pass and split backward)"""
    from feta_tmlr_amd import fused_stack
    calls = []
    orig_bn, orig_ln = fused_stack.FusedEncoderStackFn.apply, fused_stack.FusedLayerNormStackFn.apply
    fused_stack.FusedEncoderStackFn.apply = staticmethod(lambda *a: (calls.append(a[0].dtype), orig_bn(*a))[1])
    fused_stack.FusedLayerNormStackFn.apply = staticmethod(lambda *a: (calls.append(a[0].dtype), orig_ln(*a))[1])
    try:
        errs, _ = BC.check_bench_step(CPU, lambda: _lib.override_for_tests(emu), SMALL + ['--dtype', 'bf16'] + extra,
                                      share=share, replays=1)
    finally:
        fused_stack.FusedEncoderStackFn.apply, fused_stack.FusedLayerNormStackFn.apply = orig_bn, orig_ln
    assert calls and all(dt == torch.bfloat16 for dt in calls), 'the bf16 leg did not take the fused stack: %s' % calls
    print({k: '%.2e' % v for k, v in errs.items()})


def test_bench_starts_its_own_ranks(emu):
    """python bench.py --gpus 2 (no WORLD_SIZE in the environment) spawns two ranks, prints ONE JSON line
    for the whole job and exits 0."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-cpu', '--batch', '3',
                        '--n-pad', '12', '--k-eig', '6', '--layers', '1', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 6 and res['config']['parallelism'] == 'dp2'
    assert 'invalid' in res     # a rehearsal, not a measurement


def test_bench_launcher_propagates_failure():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-cpu', '--shape', 'zinc',
                        '--batch', '0', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
