"""End-to-end parity of the module API on the MI355X: DiffGraphTransformerGenGCN forward and all
parameter gradients against the fp64 CPU oracle."""
import pytest
import torch

import kernel_checks as KC
from feta_tmlr_amd.transformer import data as D
from feta_tmlr_amd.transformer.models import DiffGraphTransformerGenGCN
from oracle import feta_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,bsz,d,heads,layers,order,batch_norm,share,mode', [
    ('mutag', 8, 64, 4, 3, 4, False, 0, 'cheb'),
    ('zinc', 16, 64, 4, 3, 4, True, 0, 'cheb'),
    ('zinc', 8, 64, 4, 2, 4, False, 1, 'spectral'),
    ('zinc', 8, 64, 8, 2, 4, True, 1, 'cheb'),
])
def test_model_matches_oracle(shape, bsz, d, heads, layers, order, batch_norm, share, mode):
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    in_dim = 28
    model = DiffGraphTransformerGenGCN(in_dim, 1, d, heads, dim_feedforward=2 * d, dropout=0.0,
                                       nb_layers=layers, batch_norm=batch_norm, filter_order=order,
                                       heads_share_graph=bool(share), filter_mode=mode)
    with torch.no_grad():
        model.encoder.spectral_gnns.bias.normal_(0, 0.1)
        model.encoder.gcn.bias.normal_(0, 0.1)
    ds = D.SyntheticGraphDataset(shape, bsz, in_dim=in_dim, seed=1)
    n_pad = max(g.num_nodes for g in ds.samples)
    batch9, cache = D.collate(ds.samples, k_eig=n_pad if mode == 'spectral' else None)
    p64 = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()
           if v.dtype.is_floating_point}
    model = model.to(dev)
    x, mask, pe, _, degree, _, edge_index, batch, fi = batch9
    xg = x.to(dev).requires_grad_(True)
    out, _, coeff = model(xg, edge_index.to(dev), batch.to(dev), fi.to(dev), mask.to(dev), pe.to(dev),
                          degree=degree.to(dev), return_filter_coeff=True, graph_cache=cache.to(dev))
    w = torch.linspace(0.5, 1.5, out.numel()).view_as(out)
    ((out * w.to(dev)).sum() + 0.01 * coeff.pow(2).sum()).backward()
    torch.cuda.synchronize()

    x64 = x.double().requires_grad_(True)
    out_ref, coeff_ref = O.graph_transformer_gengcn(
        x64, edge_index, batch, fi, mask, pe.double(), degree.double(), p64, num_layers=layers,
        num_heads=heads, order=order, batch_norm=batch_norm, heads_share_graph=bool(share),
        collapsed=True)
    ((out_ref * w.double()).sum() + 0.01 * coeff_ref.pow(2).sum()).backward()
    KC.assert_close('model output', out, out_ref)
    KC.assert_close('coefficients', coeff, coeff_ref)
    KC.assert_close('dx', xg.grad, x64.grad, tol=3e-5)
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        KC.assert_close('grad ' + name, p.grad, p64[name].grad, tol=3e-5)


@pytest.mark.parametrize('shape,n_min,n_max,tie_qk,pe_on,bsz,split', [
    ('zinc', 9, 37, False, True, 128, True),
    ('zinc', 9, 37, False, True, 128, False),
    ('mutag', 3, 14, True, True, 5, True),
    ('pattern', 44, 64, False, False, 9, True),
    ('zinc', 17, 32, False, False, 33, True),
])
def test_attn_block_equals_three_launches(monkeypatch, shape, n_min, n_max, tie_qk, pe_on, bsz, split):
    import contextlib
    from test_modules_emu import check_attn_block_equals_three_launches
    check_attn_block_equals_three_launches(torch.device('cuda:0'), contextlib.nullcontext, monkeypatch, shape,
                                           n_min, n_max, tie_qk, pe_on, bsz=bsz, split=split)


@pytest.mark.parametrize('shape,n_min,n_max,batch_norm,tie_qk,pe_on,bsz', [
    ('pattern', 70, 120, True, False, True, 64),
    ('pattern', 65, 117, False, False, True, 9),
    ('pattern', 100, 188, True, True, False, 16),
    ('pattern', 200, 256, True, False, True, 3),
    ('zinc', 9, 37, True, False, True, 5),         # (N <= 64 is the block kernel's shape; this one takes any N)
])
def test_attn_out_equals_two_launches(monkeypatch, shape, n_min, n_max, batch_norm, tie_qk, pe_on, bsz):
    import contextlib
    from test_modules_emu import check_attn_out_equals_two_launches
    check_attn_out_equals_two_launches(torch.device('cuda:0'), contextlib.nullcontext, monkeypatch, shape, n_min, n_max,
                                       batch_norm, tie_qk, pe_on, bsz=bsz, use_block=shape != 'zinc')


@pytest.mark.parametrize('form', ['four waves', 'eight waves, one workgroup per graph', 'eight waves, two workgroups per graph',
                                  'two workgroups per graph walking the batch', 'one workgroup walking the batch'])
@pytest.mark.parametrize('shape,n_min,n_max,pe_on,bsz', [('zinc', 9, 37, True, 128), ('pattern', 44, 64, False, 9),
                                                         ('zinc', 9, 37, True, 300)])
def test_attn_block_forward_forms(monkeypatch, form, shape, n_min, n_max, pe_on, bsz):
    import contextlib
    from test_modules_emu import FWD_FORMS, check_attn_block_equals_three_launches
    for k, v in FWD_FORMS[form].items():
        monkeypatch.setenv(k, v)
    check_attn_block_equals_three_launches(torch.device('cuda:0'), contextlib.nullcontext, monkeypatch, shape,
                                           n_min, n_max, False, pe_on, bsz=bsz)


@pytest.mark.parametrize('n_min,n_max,bsz', [(2, 3, 1), (1, 2, 2), (16, 16, 2), (17, 17, 1), (48, 48, 1), (9, 37, 300)])
def test_fused_kernels_edge_shapes(monkeypatch, n_min, n_max, bsz):
    """tiny graphs, one graph, tile-boundary node counts, and a batch above the per-graph attention
    backward threshold and above the resident-workgroup caps (workgroups walk several graphs)"""
    import contextlib
    from test_modules_emu import check_attn_block_equals_three_launches
    check_attn_block_equals_three_launches(torch.device('cuda:0'), contextlib.nullcontext, monkeypatch, 'zinc' if bsz > 100 else 'mutag',
                                           n_min, n_max, False, True, bsz=bsz)


@pytest.mark.parametrize('shape,bsz,n_min,n_max,d,heads', [('mutag', 8, 5, 19, 32, 2), ('zinc', 64, None, None, 64, 4),
                                                            ('pattern', 6, 100, 188, 64, 4)])
def test_device_spectrum_feeds_the_model(shape, bsz, n_min, n_max, d, heads):
    import contextlib
    from test_modules_emu import check_device_spectrum_feeds_the_model
    check_device_spectrum_feeds_the_model(torch.device('cuda:0'), contextlib.nullcontext, shape, bsz, n_min, n_max,
                                          d, heads)


@pytest.mark.parametrize('shape,n_min,n_max,d,heads,tie_qk,use_block,bsz', [
    ('zinc', 9, 37, 64, 4, False, True, 32),
    ('mutag', 5, 19, 64, 4, True, False, 8),
    ('mutag', 5, 19, 64, 2, False, True, 8),
    ('pattern', 70, 120, 64, 4, False, True, 4),    # N > 64: general attention kernels inside the stack
    ('molhiv', None, 64, 64, 4, False, True, 300),  # more graphs than workgroups: persistent loops
])
def test_layernorm_stack_equals_per_op(monkeypatch, shape, n_min, n_max, d, heads, tie_qk, use_block, bsz):
    import contextlib
    from test_modules_emu import check_layernorm_stack_equals_per_op
    check_layernorm_stack_equals_per_op(torch.device('cuda:0'), contextlib.nullcontext, monkeypatch, shape, n_min, n_max,
                                        d, heads, tie_qk, bsz=bsz, use_block=use_block)


@pytest.mark.parametrize('spectral_k', [None, 6])
def test_spectral_mode_without_eigenbasis(spectral_k):
    import contextlib
    from test_modules_emu import check_spectral_mode_without_eigenbasis
    check_spectral_mode_without_eigenbasis(torch.device('cuda:0'), contextlib.nullcontext, spectral_k)


@pytest.mark.parametrize('bf16,device_key', [(False, False), (True, False), (False, True), (True, True)])
def test_layer_attention_dropout(bf16, device_key):
    import contextlib
    from test_modules_emu import check_layer_attention_dropout
    check_layer_attention_dropout(torch.device('cuda:0'), contextlib.nullcontext, bf16, device_key)


@pytest.mark.parametrize('scalar_mode', [False, True])
def test_chebconvdynamic_operator_api(scalar_mode):
    """the reference's PATTERN escape hatch learn_only_filter_order_coeff (run_transformer_gengcn_SBM_cv.py:67) on the
    MI355X, not only in the emulator"""
    import contextlib
    from test_modules_emu import check_chebconvdynamic_operator_api
    check_chebconvdynamic_operator_api(torch.device('cuda:0'), contextlib.nullcontext, scalar_mode)


def test_fused_stack_updates_running_statistics():
    import contextlib
    from test_modules_emu import check_fused_stack_updates_running_statistics
    check_fused_stack_updates_running_statistics(torch.device('cuda:0'), contextlib.nullcontext)


def test_linear_bf16_library_path_equals_tiled_kernels(monkeypatch):
    """bf16 compute of the C x C linear at config 5's row count (H * B = 4096): library bf16 GEMMs with fp32 output on
    bf16 copies of the operands (functional.LIN_LIB_BF16_MIN_ROWS) == the tiled kernels that round the fp32 operands when
    they stage them (csrc/lin.hip) - coefficients, filter output and every gradient of the node"""
    from feta_tmlr_amd import functional as FF
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    b, n, h, dh, k, order = 1024, 6, 4, 16, 4, 4
    c = order * dh * dh
    rnd = lambda *s_: torch.randn(*s_, generator=g).to(dev)
    x = rnd(n, b, h, dh).permute(1, 0, 2, 3)
    n_real = torch.randint(2, n + 1, (b,), generator=g, dtype=torch.int32).to(dev)
    u, lam = rnd(b, n, k) / 2, (torch.rand(b, k, generator=g) * 2 - 1).to(dev)
    res = []
    for min_rows in (1 << 30, 4096):
        monkeypatch.setattr(FF, 'LIN_LIB_BF16_MIN_ROWS', min_rows)
        g.manual_seed(7)          # the same operands for both runs
        pooled = (rnd(h * b, c) / 4).requires_grad_(True)
        lin_w, lin_b = (rnd(c, c) / 32).requires_grad_(True), (rnd(c) / 8).requires_grad_(True)
        bias = (rnd(dh) / 8).requires_grad_(True)
        y, coeff = FF.filter_from_pooled(x, pooled, lin_w, lin_b, bias, n_real, (u, lam), 'spec', order,
                                         heads_share_graph=True, gemm_bf16=True)
        w_out = torch.linspace(0.5, 1.5, y.numel(), device=dev).view_as(y)
        ((y * w_out).sum() + 0.01 * coeff.pow(2).sum()).backward()
        res.append((y.detach(), coeff.detach(), pooled.grad, lin_w.grad, lin_b.grad, bias.grad, pooled.detach()))
    a, bb = res
    assert torch.equal(a[6], bb[6])
    for name, t0, t1 in zip(('y', 'coeff', 'dpooled', 'dW', 'db', 'dbias'), a[:6], bb[:6]):
        scale = float(t0.abs().max())
        assert float((t0 - t1).abs().max()) <= 2e-5 * max(1.0, scale), name


@pytest.mark.parametrize('bsz,layers', [(32, 3), (300, 2)])
def test_layernorm_on_load_launches(hip, monkeypatch, bsz, layers):
    import contextlib
    from test_modules_emu import check_layernorm_on_load_launches
    check_layernorm_on_load_launches(hip[1], contextlib.nullcontext, hip[0], monkeypatch, bsz=bsz, layers=layers)
