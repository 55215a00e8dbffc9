"""Task shells and optimisation steps on the MI355X against the oracle, plus the hipGraph-captured
training step (forward + loss + backward + optimiser in one replay)."""
import contextlib

import pytest
import torch

import kernel_checks as KC
import train_checks as TC
from feta_tmlr_amd import train as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('task,batch_norm,mode', [
    ('zinc', True, 'cheb'),
    ('zinc', True, 'spectral'),
    ('tu', False, 'cheb'),
    ('molhiv', False, 'spectral'),
    ('molhiv', True, 'cheb'),
    ('sbm', False, 'cheb'),
    ('sbm', True, 'spectral'),
])
def test_one_optimisation_step_matches_oracle(hip, task, batch_norm, mode):
    TC.check_task_step(task, hip[1], contextlib.nullcontext, batch_norm=batch_norm, mode=mode)


@pytest.mark.parametrize('task,batch_norm,mode', [('molhiv', False, 'spectral'), ('zinc', True, 'cheb'),
                                                  ('sbm', False, 'cheb')])
def test_lap_pos_enc_step_matches_oracle(hip, task, batch_norm, mode):
    """--lappe --lap-dim 8 (BASELINE config 5): the embedding_lap_pos_enc branch of the three shells."""
    TC.check_task_step(task, hip[1], contextlib.nullcontext, batch_norm=batch_norm, mode=mode, lap_dim=8)


def test_config5_bf16_lappe_bucket_step(hip):
    """BASELINE config 5 in one piece: molhiv shell + lappe lap-dim 8 + bf16 storage + an N_pad <= 64 bucket of 320 graphs"""
    errs = TC.check_config5_step(hip[1], contextlib.nullcontext)
    print({k: round(v, 4) for k, v in errs.items()})


def test_molhiv_shell_outputs(hip):
    TC.check_molhiv_outputs(hip[1], contextlib.nullcontext)


def test_sbm_padded_loss_and_weighted_loss(hip):
    TC.check_sbm_padded_equals_gather(hip[1], contextlib.nullcontext)


@pytest.mark.parametrize('task', ['zinc', 'sbm', 'molhiv'])
def test_graphed_train_step_equals_eager_steps(hip, task):
    """three steps through the captured hipGraph == three eager train_step calls (same batches,
    same initial weights), including a learning-rate change between replays."""
    dev = hip[1]
    bn = task == 'zinc'
    model_a, batch9, cache = TC.build_case(task, dev, batch_norm=bn)
    model_b, _, _ = TC.build_case(task, dev, batch_norm=bn)
    model_b.load_state_dict(model_a.state_dict())
    nb = 3 if task == 'sbm' else 1
    crit = T.make_criterion(task, nb_class=nb)
    opt_a = T.make_optimizer(task, model_a.parameters(), lr=1e-3)
    opt_b = T.make_optimizer(task, model_b.parameters(), lr=1e-3, capturable=True)
    start = {k: v.clone() for k, v in model_b.state_dict().items()}
    graphed = T.GraphedTrainStep(task, model_b, crit, opt_b, batch9, cache)
    # construction (warm-up steps + capture) leaves weights, buffers and optimiser state untouched
    for k, v in model_b.state_dict().items():
        assert torch.equal(v, start[k]), k
    for st in opt_b.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                assert float(v.abs().max()) == 0.0
    lrs = [1e-3, 5e-4, 2e-3]
    for lr in lrs:
        la = T.train_step(task, model_a, crit, opt_a, batch9, T.prepare_cache(model_a, batch9, cache), lr=lr)
        graphed.set_lr(lr)
        lb = graphed(batch9, cache)
        KC.assert_close('loss', lb.cpu(), la.cpu().double(), tol=3e-5)
    # elements whose gradient is rounding noise (biases in front of a BatchNorm: exactly zero in
    # exact arithmetic) take Adam steps of arbitrary sign; compare where the first moment is real
    for (k, pa), (_, pb) in zip(model_a.named_parameters(), model_b.named_parameters()):
        if pa not in opt_a.state:
            assert torch.equal(pa, pb), k
            continue
        sig = opt_a.state[pa]['exp_avg'].abs() > 1e-6
        assert float((pa.detach() - pb.detach()).abs().max()) <= 3 * max(lrs) * 1.01, k
        if sig.any():
            KC.assert_close('param ' + k, pb.detach()[sig].cpu(), pa.detach()[sig].cpu().double(), tol=3e-5)


def test_graphed_train_step_with_attention_dropout(hip):
    """--dropout > 0 (experiments/run_transformer_gengcn.py:47) inside the captured step: the attention-probability
    masks are keyed by a DEVICE-resident (seed, offset) which every replay reads and advances, so replay i draws the
    masks of eager step i (same losses) and two replays differ (missing #4)."""
    from feta_tmlr_amd import functional as FF
    dev = hip[1]
    task = 'tu'
    model_a, batch9, cache = TC.build_case(task, dev, batch_norm=False)
    model_b, _, _ = TC.build_case(task, dev, batch_norm=False)
    model_b.load_state_dict(model_a.state_dict())
    for m in (model_a, model_b):
        for layer in m.encoder.layers:
            layer.self_attn.dropout = 0.2       # (the activations' nn.Dropout modules stay at p = 0: torch's generator
        m.train()                               #  is offset differently inside a graph)
    crit = T.make_criterion(task, nb_class=3)
    opt_a = T.make_optimizer(task, model_a.parameters(), lr=1e-3)
    opt_b = T.make_optimizer(task, model_b.parameters(), lr=1e-3, capturable=True)
    try:
        FF.DropoutState.manual_seed(77)
        eager = [float(T.train_step(task, model_a, crit, opt_a, batch9, T.prepare_cache(model_a, batch9, cache), lr=lr))
                 for lr in (1e-3, 0.0, 0.0)]
        assert eager[1] != eager[2]                 # lr = 0: only the masks changed between these two steps
        FF.DropoutState.manual_seed(77)
        graphed = T.GraphedTrainStep(task, model_b, crit, opt_b, batch9, cache)
        assert graphed.drop_calls == len(model_b.encoder.layers)
        assert FF.DropoutState.snapshot() == (77, 0)     # warm-up and capture left the key where it was
        got = []
        for lr in (1e-3, 0.0, 0.0):
            graphed.set_lr(lr)
            got.append(float(graphed(batch9, cache)))
        for i, (a, b) in enumerate(zip(eager, got)):
            assert abs(a - b) <= 3e-5 * max(1.0, abs(a)), (i, eager, got)
        assert FF.DropoutState.snapshot() == (77, 3 * graphed.drop_calls)
        torch.cuda.synchronize()
        assert FF.DropoutState._dev.tolist() == [77, 3 * graphed.drop_calls]
    finally:
        FF.DropoutState.end_device_mode()


def test_two_graphed_steps_share_one_dropout_key(hip):
    """One GraphedTrainStep per padded-size bucket (the class docstring's use): both graphs hold the address of the SAME
    persistent device key, so building the second does not orphan the first one's key (the key tensor
    was replaced, graph 1 then read freed memory), interleaved replays draw the offsets the eager loop draws on the same
    batch sequence, a re-seed reaches both graphs, and an eager step between replays advances the shared offset."""
    from feta_tmlr_amd import functional as FF
    dev = hip[1]
    task = 'tu'
    model_a, batch_x, cache_x = TC.build_case(task, dev, batch_norm=False, seed=0, bsz=4)
    _, batch_y, cache_y = TC.build_case(task, dev, batch_norm=False, seed=5, bsz=6)       # another bucket: other [B, N_pad]
    model_b, _, _ = TC.build_case(task, dev, batch_norm=False, seed=0, bsz=4)
    model_b.load_state_dict(model_a.state_dict())
    for m in (model_a, model_b):
        for layer in m.encoder.layers:
            layer.self_attn.dropout = 0.2
        m.train()
    crit = T.make_criterion(task, nb_class=3)
    opt_a = T.make_optimizer(task, model_a.parameters(), lr=0.0)
    opt_b = T.make_optimizer(task, model_b.parameters(), lr=0.0, capturable=True)
    seq = [(batch_x, cache_x), (batch_y, cache_y), (batch_x, cache_x), (batch_y, cache_y), (batch_x, cache_x)]
    try:
        FF.DropoutState.manual_seed(123)
        eager = [float(T.train_step(task, model_a, crit, opt_a, b9, T.prepare_cache(model_a, b9, c))) for b9, c in seq]
        assert len(set(eager)) == len(eager)            # lr = 0: only masks / batches differ - and they all do
        FF.DropoutState.manual_seed(999)                # (the graphs are built under another seed and re-seeded below)
        gx = T.GraphedTrainStep(task, model_b, crit, opt_b, batch_x, cache_x)
        key_x = FF.DropoutState._dev
        gy = T.GraphedTrainStep(task, model_b, crit, opt_b, batch_y, cache_y)
        assert FF.DropoutState._dev is key_x and FF.DropoutState._dev.data_ptr() == key_x.data_ptr()
        # churn the caching allocator: a key that had been freed would be overwritten here
        junk = [torch.full((2,), 7, dtype=torch.int64, device=dev) for _ in range(64)]
        del junk
        FF.DropoutState.manual_seed(123)
        got = []
        for i, (b9, c) in enumerate(seq):
            if i == 2:      # an eager step in the middle of the replays (device mode stays on)
                got.append(float(T.train_step(task, model_b, crit, opt_b, b9, T.prepare_cache(model_b, b9, c))))
            else:
                got.append(float((gx if b9 is batch_x else gy)(b9, c)))
        for i, (a, b) in enumerate(zip(eager, got)):
            assert abs(a - b) <= 3e-5 * max(1.0, abs(a)), (i, eager, got)
        torch.cuda.synchronize()
        calls = gx.drop_calls
        assert FF.DropoutState._dev.tolist() == [123, len(seq) * calls] and FF.DropoutState.snapshot() == (123, len(seq) * calls)
    finally:
        FF.DropoutState.end_device_mode()
