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
