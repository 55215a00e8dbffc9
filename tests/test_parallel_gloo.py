"""World-size-2 data parallelism on CPU (gloo): sharded graphs + flat-bucket all-reduce give the
same averaged gradients on every rank as one process holding the whole batch.  The kernels run in
the host SIMT emulation (test hook); the collective logic is what is under test."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(seed=0):
    from feta_tmlr_amd.transformer.models import DiffGraphTransformerGenGCN
    torch.manual_seed(seed)
    return DiffGraphTransformerGenGCN(8, 1, 32, 2, dim_feedforward=64, dropout=0.0, nb_layers=2,
                                      batch_norm=False, filter_order=3)


def _loss_on(model, samples, scale):
    from feta_tmlr_amd.transformer import data as D
    batch9, cache = D.collate(samples)
    x, mask, pe, _, degree, labels, edge_index, batch, fi = batch9
    out, _ = model(x, edge_index, batch, fi, mask, pe, degree=degree, graph_cache=cache)
    return ((out.squeeze(-1) - labels) ** 2).sum() * scale


def _full_grads(model):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in model.parameters()])


def _worker(rank, world, port, ret, views, wire=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ctypes
    from feta_tmlr_amd import _abi, _lib
    from feta_tmlr_amd.parallel import FlatGradAllReduce, shard_indices
    from feta_tmlr_amd.transformer import data as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    emu = _abi.bind(ctypes.CDLL(os.path.join(ROOT, 'tools', 'simt', 'libfeta_emu.so')))
    ds = D.SyntheticGraphDataset('mutag', 6, in_dim=8, seed=5, n_min=4, n_max=12)
    model = _build()
    bucket = FlatGradAllReduce(model.parameters(), world, views=views, bucket_dtype=wire)
    mine = [ds[i] for i in shard_indices(len(ds), rank, world)]
    with _lib.override_for_tests(emu):
        bucket.zero()
        _loss_on(model, mine, 1.0 / len(mine)).backward()
    bucket.all_reduce()
    ret[rank] = _full_grads(model)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('views', [False, True])
def test_flat_bucket_allreduce_world2(emu, views):
    from feta_tmlr_amd import _lib
    from feta_tmlr_amd.parallel import FlatGradAllReduce
    from feta_tmlr_amd.transformer import data as D
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 1000) + int(views)
    mp.spawn(_worker, args=(world, port, ret, views), nprocs=world, join=True)
    assert torch.equal(ret[0], ret[1]), 'ranks disagree after the all-reduce'

    # single-process reference: mean over ranks of the per-rank mean losses
    ds = D.SyntheticGraphDataset('mutag', 6, in_dim=8, seed=5, n_min=4, n_max=12)
    model = _build()
    bucket = FlatGradAllReduce(model.parameters(), 1, views=True)
    with _lib.override_for_tests(emu):
        bucket.zero()
        for r in range(world):
            mine = [ds[i] for i in range(r, len(ds), world)]
            _loss_on(model, mine, 1.0 / len(mine) / world).backward()
    ref = _full_grads(model)
    err = (ref - ret[0]).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err
    # encoder.gcn.weight travels as ONE row (its gradient is row-constant)
    g = model.encoder.gcn.weight.grad
    assert torch.equal(g, g[0:1].expand_as(g))
    # the unused outer GCN (transformer/models.py:508) stays exactly zero in the bucket
    assert float(model.gcn.weight.grad.abs().max()) == 0.0
    n_unused = model.gcn.weight.numel() + model.gcn.bias.numel()
    names = [n for n, _ in model.named_parameters()]
    assert names[-6:-4] == ['gcn.weight', 'gcn.bias'] or 'gcn.weight' in names


@pytest.mark.parametrize('world', [2, 4])
def test_bf16_wire_bucket(emu, world):
    """FlatGradAllReduce(bucket_dtype=bfloat16) (BASELINE config 3): the collective moves bf16, the gradients that
    come back are fp32, identical on every rank and within bf16 rounding (2^-8 relative per addend; the addends are
    pre-scaled by 1 / world and SUMMED) of the fp32 bucket"""
    mgr = mp.Manager()
    ret16, ret32 = mgr.dict(), mgr.dict()
    port = 29500 + (os.getpid() % 1000) + 3
    mp.spawn(_worker, args=(world, port, ret16, False, torch.bfloat16), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, port + 1, ret32, False, None), nprocs=world, join=True)
    assert ret16[0].dtype == torch.float32 and torch.equal(ret16[0], ret16[1])
    err = (ret16[0] - ret32[0]).abs().max().item()
    assert 0.0 < err <= 2.0 ** -7 * max(1.0, ret32[0].abs().max().item()), err


def test_shard_indices_partition():
    from feta_tmlr_amd.parallel import shard_indices
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in shard_indices(37, r, world))
        assert got == list(range(37))


def _build_bn(seed=0, batch_norm=True, filter_mode='cheb'):
    """d_model 64 / 4 heads: takes the fused stack, BatchNorm or LayerNorm (flat stack-gradient buffer)"""
    from feta_tmlr_amd.transformer.models import DiffGraphTransformerGenGCN
    torch.manual_seed(seed)
    return DiffGraphTransformerGenGCN(8, 1, 64, 4, dim_feedforward=128, dropout=0.0, nb_layers=2,
                                      batch_norm=batch_norm, filter_order=2, heads_share_graph=True,
                                      filter_mode=filter_mode)


def _worker_inplace(rank, world, port, ret, batch_norm, lowp=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ctypes
    from feta_tmlr_amd import _abi, _lib
    from feta_tmlr_amd.parallel import FlatBufferAllReduce, FlatGradAllReduce, HybridGradAllReduce, shard_indices
    from feta_tmlr_amd.transformer import data as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    emu = _abi.bind(ctypes.CDLL(os.path.join(ROOT, 'tools', 'simt', 'libfeta_emu.so')))
    ds = D.SyntheticGraphDataset('mutag', 4, in_dim=64, seed=5, n_min=4, n_max=12)
    mine = [ds[i] for i in shard_indices(len(ds), rank, world)]
    batch9, cache = D.collate(mine, k_eig=12) if lowp else D.collate(mine)
    x, mask, pe, _, degree, _, edge_index, batch, fi = batch9
    src = x.permute(1, 0, 2).contiguous()
    g = torch.Generator().manual_seed(100 + rank)
    dout = torch.randn(src.shape, generator=g)
    if lowp:       # bf16 storage (BASELINE config 3): token rows and pe of the resident batch in the storage type
        src, pe = src.to(torch.bfloat16), pe.to(torch.bfloat16)
    out = {}
    for mode in ('packed', 'inplace'):
        enc = _build_bn(batch_norm=batch_norm, filter_mode='spectral' if lowp else 'cheb').encoder
        if lowp:
            from feta_tmlr_amd.transformer.layers import set_storage_dtype
            set_storage_dtype(enc, torch.bfloat16)
        enc.train()
        kw = dict(degree=degree, src_key_padding_mask=mask, graph_cache=cache)
        with _lib.override_for_tests(emu):
            if mode == 'packed':      # one backward, one packed bucket
                o, _, _ = enc(src, pe, edge_index, fi, batch, **kw)
                o.backward(gradient=dout)
                FlatGradAllReduce(enc.parameters(), world).all_reduce()
            else:                     # the flow of bench.py --gpus N
                enc.keep_stack_boundary = True
                r_head = HybridGradAllReduce(enc.head_parameters(), world, big_numel=1 << 12)
                r_stack = FlatBufferAllReduce(enc.stack_flat_grad, world)
                o, _, _ = enc(src, pe, edge_index, fi, batch, **kw)
                enc.backward_head(o, dout)
                gw = enc.gcn.weight.grad      # dense, every row equal: row 0 travels, finish() rewrites the rest
                assert gw.is_contiguous() and torch.equal(gw, gw[0:1].expand_as(gw))
                w1 = r_head.start()
                enc.backward_stack()
                flat = enc.stack_flat_grad()
                for p in enc.stack_parameters():   # every stack gradient is a view of the flat buffer
                    assert p.grad is None or (flat.data_ptr() <= p.grad.data_ptr() <
                                              flat.data_ptr() + 4 * flat.numel())
                w2 = r_stack.start()
                r_head.finish(w1)
                r_stack.finish(w2)
                gw = enc.gcn.weight.grad
                assert torch.equal(gw, gw[0:1].expand_as(gw))
        out[mode] = _full_grads(enc)
    ret[rank] = (out['packed'], out['inplace'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('batch_norm', [True, False])
def test_inplace_reducers_equal_packed_bucket_world2(emu, batch_norm):
    """HybridGradAllReduce (big gradients in place) + FlatBufferAllReduce (the fused stack's flat
    gradient buffer) give the averaged gradients of the packed single bucket"""
    port = 29500 + (os.getpid() % 400) + 7 + (0 if batch_norm else 11)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_inplace, args=(2, port, ret, batch_norm), nprocs=2, join=True)
    for rank in (0, 1):
        packed, inplace = ret[rank]
        assert torch.allclose(packed, inplace, rtol=1e-6, atol=1e-7), float((packed - inplace).abs().max())
    assert torch.allclose(ret[0][1], ret[1][1])


@pytest.mark.parametrize('batch_norm', [True, False])
def test_bf16_stack_split_backward_world2(emu, batch_norm):
    """the bf16 storage stack (fused bf16 kernels, fp32 gradients) under the split backward of bench.py --gpus N:
    in-place reducers == the packed bucket, identical on both ranks; batch_norm=False: the LayerNorm-on-load stack
    (ABI 9), whose flat gradient buffer carries [dgamma | dbeta] inside the kernels' split-K slots"""
    port = 29500 + (os.getpid() % 400) + 23 + (0 if batch_norm else 13)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_inplace, args=(2, port, ret, batch_norm, True), nprocs=2, join=True)
    for rank in (0, 1):
        packed, inplace = ret[rank]
        assert torch.allclose(packed, inplace, rtol=1e-5, atol=1e-6), float((packed - inplace).abs().max())
    assert torch.allclose(ret[0][1], ret[1][1])
