"""Task shells (graph regression / classification, ogbg-molhiv, SBM node classification) and one
optimisation step per task against the oracle; kernels run in the host SIMT emulation."""
import contextlib

import pytest
import torch

import train_checks as TC
from feta_tmlr_amd import _lib
from feta_tmlr_amd import train as T
from oracle import feta_oracle as O


def _ctx(emu):
    return lambda: _lib.override_for_tests(emu)


@pytest.mark.parametrize('task,batch_norm,mode', [
    ('zinc', True, 'cheb'),
    ('tu', False, 'cheb'),
    ('molhiv', False, 'spectral'),
    ('sbm', False, 'cheb'),
])
def test_one_optimisation_step_matches_oracle(emu, task, batch_norm, mode):
    TC.check_task_step(task, torch.device('cpu'), _ctx(emu), batch_norm=batch_norm, mode=mode)


@pytest.mark.parametrize('task,batch_norm,mode', [('molhiv', False, 'spectral'), ('zinc', True, 'cheb')])
def test_lap_pos_enc_step_matches_oracle(emu, task, batch_norm, mode):
    """--lappe --lap-dim 8 (BASELINE config 5): the embedding_lap_pos_enc branch of the shells."""
    TC.check_task_step(task, torch.device('cpu'), _ctx(emu), batch_norm=batch_norm, mode=mode, lap_dim=8)


def test_config5_bf16_lappe_bucket_step(emu):
    """BASELINE config 5 in one piece (molhiv shell + lappe lap-dim 8 + bf16 storage + an N_pad <= 64 bucket), emulated at
    5 graphs; the MI355X suite runs 320"""
    TC.check_config5_step(torch.device('cpu'), _ctx(emu), bsz=5, n_min=20, n_max=36)


def test_oracle_lap_encoding_matches_product():
    """LapEncoding (transformer/position_encoding.py:127-161): product and oracle agree up to the sign of
    each column on graphs with simple low eigenvalues; zero-padded columns for graphs smaller than dim."""
    import numpy as np
    from feta_tmlr_amd.transformer import data as D
    from feta_tmlr_amd.transformer.position_encoding import LapEncoding, laplacian_dense
    ds = D.SyntheticGraphDataset('mutag', 6, in_dim=4, seed=3, n_min=3, n_max=15)
    enc = LapEncoding(8, normalization='sym')
    for g in ds.samples:
        got = enc.compute_pe(g)
        ref = O.lap_encoding(g.edge_index, g.num_nodes, 8).numpy()
        assert got.shape == ref.shape == (g.num_nodes, 8)
        lam = np.linalg.eigvalsh(laplacian_dense(g.edge_index, g.num_nodes, 'sym'))
        for c in range(8):
            k = c + 1
            if k >= g.num_nodes:
                assert not got[:, c].any() and not ref[:, c].any()
                continue
            gap = min(lam[k] - lam[k - 1], (lam[k + 1] - lam[k]) if k + 1 < g.num_nodes else 1.0)
            if gap > 1e-6:      # a simple eigenvalue: the column is unique up to its sign
                assert min(np.abs(got[:, c] - ref[:, c]).max(), np.abs(got[:, c] + ref[:, c]).max()) < 1e-5


def test_molhiv_shell_outputs(emu):
    TC.check_molhiv_outputs(torch.device('cpu'), _ctx(emu))


def test_sbm_padded_loss_and_weighted_loss(emu):
    TC.check_sbm_padded_equals_gather(torch.device('cpu'), _ctx(emu))


def test_warmup_schedule_matches_oracle():
    for s in (0, 1, 99, 100, 101, 5000):
        assert abs(T.warmup_lr(s, 1e-3, 100) - O.warmup_lr(s, 1e-3, 100)) < 1e-15
    assert abs(T.warmup_lr(0, 1e-3, 100) - 1e-6) < 1e-15
    assert abs(T.warmup_lr(100, 1e-3, 100) - 1e-3) < 1e-12
    assert abs(T.warmup_lr(400, 1e-3, 100) - 5e-4) < 1e-12


def test_lap_sign_flip_is_a_column_sign():
    g = torch.Generator().manual_seed(3)
    lap = torch.randn(4, 7, 5)
    out = T.lap_sign_flip(lap, g)
    ratio = (out / lap).reshape(-1, 5)
    assert torch.all((ratio.abs() - 1).abs() < 1e-6)
    assert torch.all(ratio == ratio[0:1])           # one sign per column for the whole batch


def test_accuracy_sbm_known_answer():
    scores = torch.tensor([[2., 0, 0], [2., 0, 0], [0, 2., 0], [0, 0, 2.], [2., 0, 0]])
    targets = torch.tensor([0, 0, 1, 1, 2])
    # recall: class0 2/2, class1 1/2, class2 0/1 -> mean 50 %
    assert abs(T.accuracy_SBM(scores, targets) - 50.0) < 1e-9


def test_regularisation_matches_oracle():
    from feta_tmlr_amd.transformer.models import regularisation_max_cos
    c = torch.randn(3, 4, 16, dtype=torch.float64)
    assert abs(float(regularisation_max_cos(c)) - float(O.regularisation_max_cos(c))) < 1e-12


def test_rocauc_matches_sklearn():
    from sklearn.metrics import roc_auc_score
    g = torch.Generator().manual_seed(0)
    s = torch.randn(200, generator=g).round(decimals=1)          # ties on purpose
    y = (torch.rand(200, generator=g) < 0.3).float()
    y[::17] = float('nan')
    keep = ~torch.isnan(y)
    assert abs(T.rocauc(s, y) - roc_auc_score(y[keep].numpy(), s[keep].numpy())) < 1e-12


@pytest.mark.parametrize('task', ['zinc', 'tu', 'molhiv', 'sbm'])
def test_evaluate_metrics(emu, task):
    """eval-mode metrics of one split (running BatchNorm statistics, no gradient) against the same
    quantities computed from the model outputs by hand"""
    model, batch9, cache = TC.build_case(task, torch.device('cpu'), batch_norm=(task == 'zinc'))
    crit = T.make_criterion(task, nb_class=3 if task in ('tu', 'sbm') else 1)
    with _lib.override_for_tests(emu):
        res = T.evaluate(task, model, crit, [(batch9, cache), (batch9, cache)])
        model.eval()
        with torch.no_grad():
            loss, out = T.task_loss(task, model, crit, batch9, cache)
        model.train()
    assert model.training
    assert abs(res['loss'] - float(loss)) < 1e-6
    labels = batch9[5]
    if task == 'zinc':
        assert abs(res['mae'] - float((out - labels.view(out.shape)).abs().mean())) < 1e-6
        assert abs(res['mse'] - float(((out - labels.view(out.shape)) ** 2).mean())) < 1e-6
    elif task == 'tu':
        assert abs(res['acc'] - float((out.argmax(1) == labels).float().mean())) < 1e-9
    elif task == 'sbm':
        assert abs(res['acc'] - T.accuracy_SBM(out, labels)) < 1e-9
    else:
        assert res['rocauc'] == T.rocauc(torch.cat([out.view(-1)] * 2), torch.cat([labels] * 2))
