"""Golden fixtures on the MI355X (no oracle call, nothing read outside the repo)."""
import pytest
import torch

import golden_checks as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['model_mutag_b4', 'model_zinc_b8_bn'])
def test_model_fixture(name):
    G.check_model_fixture(name, torch.device('cuda:0'))


@pytest.mark.parametrize('mode', ['cheb', 'spec'])
@pytest.mark.parametrize('name', ['filter_zinc_b8', 'filter_pattern_n120'])
def test_filter_fixture(hip, name, mode):
    abi, dev, stream = hip
    G.check_filter_fixture(name, abi, dev, stream, mode)


@pytest.mark.parametrize('name', sorted(G.STEP_FIXTURES))
def test_step_fixture(name):
    import contextlib
    G.check_step_fixture(name, torch.device('cuda:0'), contextlib.nullcontext)
