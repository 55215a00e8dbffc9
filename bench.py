"""graphs/s, forward+backward of the FeTA spectral-attention hot path on synthetic ZINC-shaped
padded-graph batches (BASELINE.json configs[1]: B=128, N_pad=37, d=64, 4 heads, K=16 eigenpairs,
fp32), one process per GPU, weak scaling.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = zero grads, forward and backward of DiffTransformerEncoderGenGCN (all layers, attention
-> coefficient generator -> spectral filter on the last layer -> linear_cat) on one batch that is
already resident in HBM, plus, for N > 1, the RCCL all-reduce of the flat gradient bucket.
Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      the dominant hand-written kernel of the step, timed live with HIP events
  cpu_baseline  the reference-faithful CPU restatement (oracle/) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from feta_tmlr_amd import _lib                                            # noqa: E402
from feta_tmlr_amd import functional as FF                                # noqa: E402
from feta_tmlr_amd.parallel import FlatBufferAllReduce, FlatGradAllReduce, HybridGradAllReduce                      # noqa: E402
from feta_tmlr_amd.transformer import data as D                           # noqa: E402
from feta_tmlr_amd.transformer.layers import DiffTransformerEncoderLayer  # noqa: E402
from feta_tmlr_amd.transformer.models import DiffTransformerEncoderGenGCN  # noqa: E402

def log(msg):
    sys.stderr.write('[bench %.1fs] %s\n' % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, capped at the GPU
    box's per-GPU share of 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


_T0 = time.perf_counter()
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=128, help='graphs per GPU per step')
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--heads', type=int, default=4)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--order', type=int, default=4)
    ap.add_argument('--k-eig', type=int, default=16)
    ap.add_argument('--n-pad', type=int, default=37)
    ap.add_argument('--shape', default='zinc', choices=['mutag', 'zinc', 'pattern', 'molhiv'],
                    help='synthetic graph-size distribution (SURVEY 8d); the headline metric is zinc')
    ap.add_argument('--layer-norm', action='store_true', help='LayerNorm instead of the ZINC default BatchNorm')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of one hipGraph per step')
    ap.add_argument('--two-phase', action='store_true',
                    help='force the split backward (default for --gpus > 1: overlaps the all-reduce of the '
                         'filter-stage gradients with the backward of the encoder stack)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=10)
    ap.add_argument('--kernel-iters', type=int, default=200)
    ap.add_argument('--stream-batch', type=int, default=16384,
                    help='graphs in the streaming-batch run of the north-star kernel (roofline_streaming); 0 = skip')
    return ap.parse_args()


def make_batch(args, rank, dev):
    n_max = min(args.n_pad, D.SHAPES[args.shape][1])
    n_min = min(D.SHAPES[args.shape][0], n_max)
    ds = D.SyntheticGraphDataset(args.shape, args.batch, in_dim=args.dim, seed=rank, n_min=n_min, n_max=n_max)
    batch9, cache = D.collate(ds.samples, k_eig=args.k_eig, n_pad=args.n_pad)
    x, mask, pe, _, degree, _, edge_index, batch, fi = batch9
    src = x.permute(1, 0, 2).contiguous()          # [N,B,d] seq-first, embedding skipped
    g = torch.Generator().manual_seed(1000 + rank)
    dout = torch.randn(src.shape, generator=g)
    cpu = dict(src=src, mask=mask, pe=pe, degree=degree, edge_index=edge_index, batch=batch, fi=fi,
               dout=dout, cache=cache)
    gpu = {k: (v.to(dev) if v is not None else None) for k, v in cpu.items()}
    return cpu, gpu


def build_encoder(args):
    torch.manual_seed(0)
    layer = DiffTransformerEncoderLayer(args.dim, args.heads, 2 * args.dim, 0.0,
                                        batch_norm=not args.layer_norm)
    return DiffTransformerEncoderGenGCN(args.dim, args.heads, layer, args.layers,
                                        num_coefficients=args.order, heads_share_graph=True,
                                        filter_mode='spectral')


def time_kernel(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def roofline(args, gpu, dev):
    """Times the hand-written kernels of one step in isolation, in exactly the variants the fused stack
    issues (feta_tmlr_amd/benchcases.py; HIP events on the stream they are launched on - torch's
    current stream), picks the DOMINANT one = largest launches-per-step x launch time, and prices it
    against the HBM roofline with its algorithmic bytes.  `traffic` = HBM bytes
    per launch from the rocprofv3 PMC passes committed in profiles/traffic.json (FETCH_SIZE x2 +
    WRITE_SIZE, tools/pmc_summary.py), null if absent."""
    from feta_tmlr_amd.benchcases import stack_layer_cases
    abi, st = _lib.abi(), _lib.stream_handle()
    b, n, h, d = args.batch, args.n_pad, args.heads, args.dim
    dh, k_eig, p = d // h, args.k_eig, args.order
    c = p * dh * dh
    L = args.layers
    nr = gpu['cache'].n_real
    rnd = lambda *s: torch.randn(*s, device=dev)
    cand = []   # (name, launches per step, fn, algorithmic bytes)
    for name, per_layer, fn, nbytes, _ in stack_layer_cases(abi, st, dev, b, n, d, h, 2 * d, gpu['pe'], nr):
        if name == 'attn_block_fwd (no attn write)':
            cnt = L - 1
        elif name == 'attn_block_fwd (+attn write)':
            cnt = 1
        elif 'linear2' in name:
            cnt = L + 1      # + linear_cat backward (same shape class)
        else:
            cnt = L
        cand.append((name, cnt, fn, nbytes))
    qkv = rnd(n, b, 3 * d)
    v5 = qkv.view(n, b, 3, h, dh)
    q, k, v = (v5[:, :, i].permute(1, 0, 2, 3) for i in range(3))
    tok = lambda: torch.empty(n, b, h, dh, device=dev).permute(1, 0, 2, 3)
    out, dout = tok(), rnd(n, b, h, dh).permute(1, 0, 2, 3)
    stats = torch.rand(b, h, n, 2, device=dev) + 1.0
    delta = torch.empty(b, h, n, device=dev)
    dqkv = torch.empty_like(qkv)
    g5 = dqkv.view(n, b, 3, h, dh)
    dq, dk, dv = (g5[:, :, i].permute(1, 0, 2, 3) for i in range(3))
    sc = dh ** -0.5
    cand.append(('attn_bwd (dq + dkdv)', L,
                 lambda: abi.attn_bwd(q, k, v, gpu['pe'], nr, out, dout, stats, delta, dq, dk, dv, sc, st),
                 4 * b * (3 * n * d + 2 * n * d + n * n + 2 * h * n + 3 * n * d + h * n)))
    xs, dys = rnd(n, b, h, dh).permute(1, 0, 2, 3), rnd(n, b, h, dh).permute(1, 0, 2, 3)
    ys, dxs = tok(), tok()
    coeff, bias = rnd(h * b, c), rnd(dh)
    dcoeff, dbp = torch.empty_like(coeff), torch.empty(b * h, dh, device=dev)
    u, lam = gpu['cache'].u, gpu['cache'].lam
    cand.append(('spec_filter_fwd', 1, lambda: abi.spec_filter_fwd(xs, u, lam, coeff, bias, nr, ys, p, 1, st),
                 4 * b * (n * d + n * k_eig + k_eig + h * c + n * d)))
    cand.append(('spec_filter_bwd', 1,
                 lambda: abi.spec_filter_bwd(xs, u, lam, coeff, nr, dys, dxs, dcoeff, dbp, p, 1, st),
                 4 * b * (2 * n * d + n * k_eig + k_eig + h * c + n * d + h * c)))
    rows = []
    for name, cnt, fn, nbytes in cand:
        t = time_kernel(fn, args.kernel_iters)
        rows.append({'kernel': name, 'launches_per_step': cnt, 'launch_us': round(t * 1e6, 3),
                     'algorithmic_bytes': nbytes, 'achieved': round(nbytes / t / 1e9, 2),
                     'frac': round(nbytes / t / 1e9 / HBM_PEAK_GBS, 5)})
    traffic = {}
    tp = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp))
        except Exception:
            traffic = {}
    for r in rows:
        r['traffic'] = (traffic.get(r['kernel']) or {}).get('hbm_bytes')
    dom = max(rows, key=lambda r: r['launches_per_step'] * r['launch_us'])
    res = {'kernel': dom['kernel'], 'bound': 'hbm', 'achieved': dom['achieved'], 'peak': HBM_PEAK_GBS,
           'unit': 'GB/s', 'frac': dom['frac'], 'traffic': dom['traffic'],
           'algorithmic_bytes': dom['algorithmic_bytes'], 'launch_us': dom['launch_us'],
           'launches_per_step': dom['launches_per_step'], 'other_kernels': [r for r in rows if r is not dom]}
    return res


def roofline_streaming(args, dev):
    """The north-star kernel (eigenbasis filter: U^T X -> g(Lambda) -> U) at a batch whose working
    set (~0.6 GB) does not fit the 256 MB Infinity Cache, so that HBM is what is measured; same
    kernel, same shape per graph as the BASELINE batch.  HIP events on the launch stream."""
    abi, st = _lib.abi(), _lib.stream_handle()
    b, n, h, d = args.stream_batch, args.n_pad, args.heads, args.dim
    dh, k_eig, p = d // h, args.k_eig, args.order
    c = p * dh * dh
    g = torch.Generator(device='cpu').manual_seed(1)
    nr = torch.randint(9, n + 1, (b,), generator=g, dtype=torch.int32).to(dev)
    rnd = lambda *s: torch.randn(*s, device=dev)
    xs, dys = (rnd(n, b, h, dh).permute(1, 0, 2, 3) for _ in range(2))
    ys, dxs = (torch.empty(n, b, h, dh, device=dev).permute(1, 0, 2, 3) for _ in range(2))
    u, lam = rnd(b, n, k_eig), torch.rand(b, k_eig, device=dev) * 2 - 1
    coeff, bias = rnd(h * b, c), rnd(dh)
    dcoeff, dbp = torch.empty_like(coeff), torch.empty(b * h, dh, device=dev)
    out = []
    for name, fn, nbytes in (
            ('spec_filter_fwd', lambda: abi.spec_filter_fwd(xs, u, lam, coeff, bias, nr, ys, p, 1, st),
             4 * b * (n * d + n * k_eig + k_eig + h * c + n * d)),
            ('spec_filter_bwd', lambda: abi.spec_filter_bwd(xs, u, lam, coeff, nr, dys, dxs, dcoeff, dbp, p, 1, st),
             4 * b * (2 * n * d + n * k_eig + k_eig + h * c + n * d + h * c))):
        t = time_kernel(fn, 20)
        out.append({'kernel': name, 'batch': b, 'bound': 'hbm', 'launch_us': round(t * 1e6, 2),
                    'algorithmic_bytes': nbytes, 'achieved': round(nbytes / t / 1e9, 1), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4)})
    return out


def cpu_baseline(args, cpu, enc):
    """Reference-faithful CPU restatement (edge-list recursion, per-node weight copies, the
    un-collapsed GCNConv on ones, Python loop over H*B blocks), PyTorch CPU fp32, all host cores."""
    from oracle import feta_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    p = {k: v.detach().cpu().float() for k, v in enc.state_dict().items() if v.dtype.is_floating_point}
    sub = args.batch                               # the whole per-GPU batch, a few steps
    m = cpu['mask'][:sub]
    nb = (~m).sum(-1)
    n_tot = int(nb.sum())
    ei = cpu['edge_index']
    ei = ei[:, ei[0] < n_tot]

    def step(collapsed=False):
        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        src = cpu['src'][:, :sub].clone().requires_grad_(True)
        out, _, _ = O.encoder_gengcn(src, cpu['pe'][:sub], ei, cpu['fi'][:n_tot], cpu['batch'][:n_tot],
                                     cpu['degree'][:sub], m, leaves, args.layers, args.heads, args.order,
                                     batch_norm=not args.layer_norm, heads_share_graph=True,
                                     collapsed=collapsed)
        (out * cpu['dout'][:, :sub]).sum().backward()

    step()
    log('cpu baseline warm-up step done')
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step()
    dt = (time.perf_counter() - t0) / args.cpu_steps
    # the same restatement with the exact GCNConv(ones) = c_j colsum(W) + b collapse, so that
    # the GPU/CPU ratio is not inflated by the reference's redundant ones @ W product alone
    step(True)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step(True)
    dt_opt = (time.perf_counter() - t0) / args.cpu_steps
    return {'value': round(sub / dt, 2), 'unit': 'graphs/s', 'cores': cores, 'kind': 'port',
            'sample': '%d steps of fwd+bwd on the first %d graphs of the batch (oracle.encoder_gengcn, '
                      'faithful formulation, exact Chebyshev operator, torch CPU fp32, %d threads)'
                      % (args.cpu_steps, sub, cores),
            'optimised_value': round(sub / dt_opt, 2),
            'optimised_sample': 'same, with the collapsed coefficient generator (no ones @ W, no dense edge list)'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    _lib.abi()

    log('library loaded')
    cpu, gpu = make_batch(args, rank, dev)
    log('batch built')
    enc = build_encoder(args).to(dev)
    enc.train()
    params = [p for p in enc.parameters()]
    two_phase = args.two_phase or world > 1
    fwd_args = (gpu['src'], gpu['pe'], gpu['edge_index'], gpu['fi'], gpu['batch'])
    fwd_kw = dict(degree=gpu['degree'], src_key_padding_mask=gpu['mask'], graph_cache=gpu['cache'])
    use_graph = not args.no_graph

    def capture(fn):
        """hipGraph of fn() (after warm-up on a side stream), or fn itself with --no-graph."""
        if not use_graph:
            return fn
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=capture.pool):
            fn()
        capture.pool = g.pool()
        return g.replay
    capture.pool = None

    if not two_phase:
        reducer = FlatGradAllReduce(params, world)   # fresh .grad per step, one flat bucket for RCCL

        def fwd_bwd():
            reducer.zero()
            out, _, _ = enc(*fwd_args, **fwd_kw)
            out.backward(gradient=gpu['dout'])   # upstream gradient dOut ~ N(0,1) injected directly

        fwd_bwd()
        torch.cuda.synchronize()
        log('first eager step done')
        run = capture(fwd_bwd)

        def step():
            run()
            if world > 1:
                reducer.all_reduce()
    else:
        # split backward: the filter-stage gradients (96 % of the bytes) are ready first; their
        # all-reduce runs under the backward of the encoder stack
        enc.keep_stack_boundary = True
        # head: the two 4 MB gradients in place + one small packed bucket, all under the stack backward;
        # stack: its gradients already live in one flat buffer - one in-place collective, no pack / unpack
        r_head = HybridGradAllReduce(enc.head_parameters(), world)
        r_stack = FlatBufferAllReduce(enc.stack_flat_grad, world)

        def phase1():
            r_head.zero()
            for p_ in enc.stack_parameters():
                p_.grad = None
            out, _, _ = enc(*fwd_args, **fwd_kw)
            enc.backward_head(out, gpu['dout'])

        def both():
            phase1()
            enc.backward_stack()

        both()
        torch.cuda.synchronize()
        log('first eager step done')
        if use_graph:   # warm-up runs whole steps; the two graphs share one memory pool
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    both()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                phase1()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool()):
                enc.backward_stack()
            run1, run2 = g1.replay, g2.replay
        else:
            run1, run2 = phase1, enc.backward_stack

        def step():
            run1()
            w1 = r_head.start()
            run2()
            w2 = r_stack.start()
            r_head.finish(w1)
            r_stack.finish(w2)

    graph = use_graph
    log('graph captured' if graph else 'eager mode')
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        total_graphs = args.batch * world * args.steps
        res = {
            'metric': 'graphs/sec fwd+bwd, ZINC batch (N<=37,d=64,K=16)',
            'value': round(total_graphs / dt, 2), 'unit': 'graphs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': args.shape.upper() + '-shaped synthetic padded-graph batch, ChebConvDynamic block '
                                   '(attention + coefficient generator + spectral filter), fwd+bwd',
                       'graphs_per_gpu': args.batch, 'global_batch': args.batch * world,
                       'n_pad': args.n_pad, 'd_model': args.dim, 'heads': args.heads,
                       'filter_order': args.order, 'k_eig': args.k_eig, 'layers': args.layers,
                       'norm': 'layer' if args.layer_norm else 'batch(per-rank stats)',
                       'heads_share_graph': True, 'hip_graph': bool(use_graph),
                       'backward': 'two-phase (head all-reduce under stack backward)' if two_phase else 'single',
                       'parallelism': 'dp%d' % world},
        }
        log('timed region done: %.3f ms/step' % (dt / args.steps * 1e3))
        res['roofline'] = roofline(args, gpu, dev)
        if args.stream_batch > 0:
            res['roofline_streaming'] = roofline_streaming(args, dev)
        log('roofline kernel timed')
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args, cpu, enc)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
