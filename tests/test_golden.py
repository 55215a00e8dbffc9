"""Golden fixtures: (1) the oracle still reproduces them (pins the restatement against drift),
(2) the kernel sources, run in the host emulation, reproduce them through the product API."""
import numpy as np
import pytest
import torch

import golden_checks as G
import kernel_checks as KC
from feta_tmlr_amd import _lib
from oracle import feta_oracle as O

CPU = torch.device('cpu')


@pytest.mark.parametrize('name', ['model_mutag_b4', 'model_zinc_b8_bn'])
def test_oracle_reproduces_model_fixture(name):
    z = G.load(name)
    d, heads, layers, order, bn, share, _ = (int(v) for v in z['cfg'])
    p = {k[len('param/'):]: torch.from_numpy(v).double() for k, v in z.items() if k.startswith('param/')}
    t = lambda k: torch.from_numpy(z[k])
    out, coeff = O.graph_transformer_gengcn(t('x').double(), t('edge_index'), t('batch'), t('feature_indices'),
                                            t('mask'), t('pe').double(), t('degree').double(), p,
                                            num_layers=layers, num_heads=heads, order=order,
                                            batch_norm=bool(bn), heads_share_graph=bool(share))
    # fixtures are stored in fp32: agreement to fp32 rounding of the stored values
    KC.assert_close('out', out, t('out'), tol=2e-6)
    KC.assert_close('coeff', coeff, t('coeff'), tol=2e-6)


@pytest.mark.parametrize('name', ['model_mutag_b4', 'model_zinc_b8_bn'])
def test_emulated_kernels_reproduce_model_fixture(emu, name):
    with _lib.override_for_tests(emu):
        G.check_model_fixture(name, CPU)


@pytest.mark.parametrize('mode', ['cheb', 'spec'])
@pytest.mark.parametrize('name', ['filter_zinc_b8', 'filter_pattern_n120'])
def test_emulated_kernels_reproduce_filter_fixture(emu, name, mode):
    G.check_filter_fixture(name, emu, CPU, None, mode)


@pytest.mark.parametrize('name', sorted(G.STEP_FIXTURES))
def test_emulated_kernels_reproduce_step_fixture(emu, name):
    G.check_step_fixture(name, CPU, lambda: _lib.override_for_tests(emu))


@pytest.mark.parametrize('name', sorted(G.STEP_FIXTURES))
def test_oracle_reproduces_step_fixture(name):
    """the restatement still yields the stored loss / output for the stored batch and parameters"""
    import train_checks as TC
    task, bn, mode = G.STEP_FIXTURES[name]
    z = G.load(name)
    model, _, _ = TC.build_case(task, CPU, bsz=2, d=16, heads=2, layers=2, order=2, batch_norm=bn, mode=mode)
    p64 = {k[len('param/'):]: torch.from_numpy(v).double() for k, v in z.items() if k.startswith('param/')}
    t = lambda k: torch.from_numpy(z['batch/' + k]) if 'batch/' + k in z else None
    batch9 = tuple(t(k) for k in ('x', 'mask', 'pe', 'lap_pe', 'degree', 'labels', 'edge_index', 'batch',
                                  'feature_indices'))
    with torch.no_grad():
        out, loss, coeff, _ = TC.oracle_forward(task, model, batch9, p64, bn)
    KC.assert_close('loss', loss, torch.from_numpy(z['loss']), tol=2e-6)
    KC.assert_close('out', out, torch.from_numpy(z['out']), tol=2e-6)
    KC.assert_close('coeff', coeff, torch.from_numpy(z['coeff']), tol=2e-6)
