"""graphs/s, forward+backward of the FeTA spectral-attention hot path on synthetic ZINC-shaped
padded-graph batches (BASELINE.json configs[1]: B=128, N_pad=37, d=64, 4 heads, K=16 eigenpairs,
fp32), one process per GPU, weak scaling.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus 4 --steps 50 --warmup 10        # starts its own 4 ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = zero grads, forward and backward of DiffTransformerEncoderGenGCN (all layers, attention
-> coefficient generator -> spectral filter on the last layer -> linear_cat) on one batch that is
already resident in HBM, plus, for N > 1, the RCCL all-reduce of the flat gradient bucket.
Rank 0 prints ONE JSON line (contract in the task description) with these extra objects:
  roofline           the dominant hand-written kernel SYMBOL of the step (largest launches x time), timed
                     live with HIP events in the variants the stack issues
  roofline_streaming the north-star kernel (U^T X -> g(Lambda) -> U) at a batch beyond the Infinity Cache
  cpu_baseline       the CPU restatement (oracle/) of the SAME operator as the timed GPU leg
  reference_literal  the reference-literal operator (filter_mode='cheb': order-P Chebyshev recursion,
                     heads_share_graph=False: transformer/models.py:186) timed the same way, with its own
                     cpu_baseline (the reference-faithful restatement)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from feta_tmlr_amd import _lib                                            # noqa: E402
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)     # (a step is 0.28 ms: 200 steps read the steady state, the first
    ap.add_argument('--warmup', type=int, default=20)     # replays after a capture are ~1.5 % slower)
    ap.add_argument('--batch', type=int, default=128, help='graphs per GPU per step')
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--heads', type=int, default=4)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--order', type=int, default=4)
    ap.add_argument('--k-eig', type=int, default=16)
    ap.add_argument('--n-pad', type=int, default=37)
    ap.add_argument('--shape', default='zinc', choices=['mutag', 'zinc', 'pattern', 'molhiv'],
                    help='synthetic graph-size distribution (SURVEY 8d); the headline metric is zinc')
    ap.add_argument('--filter-mode', default='spectral', choices=['spectral', 'cheb'],
                    help="operator of the timed leg: 'spectral' = K-eigenpair eigenbasis filter (the BASELINE "
                         "metric's K=16), 'cheb' = the reference's order-P Chebyshev recursion")
    ap.add_argument('--no-share-graph', action='store_true',
                    help='heads_share_graph=False: the reference-literal un-replicated edge_index '
                         '(transformer/models.py:186), heads >= 1 filtered with L_hat = 0')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="storage dtype of the timed leg: 'f32' = the reference's arithmetic (the headline), 'bf16' = "
                         'BASELINE configs 3 / 5: bf16 activations / pe / U / attn / per-block filter weights, bf16 MFMA, '
                         'fp32 statistics and master weights, bf16 gradient bucket')
    ap.add_argument('--no-literal', action='store_true', help='skip the reference_literal leg')
    ap.add_argument('--layer-norm', action='store_true', help='LayerNorm instead of the ZINC default BatchNorm')
    ap.add_argument('--no-pe', action='store_true',
                    help='pe=None: no relative positional kernel (what the README commands of the TU / molhiv / SBM '
                         'scripts run: they pass no --pos-enc, README.md:49,65,71)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of one hipGraph per step')
    ap.add_argument('--two-phase', action='store_true',
                    help='force the split backward (default for --gpus > 1: overlaps the all-reduce of the '
                         'filter-stage gradients with the backward of the encoder stack)')
    ap.add_argument('--graph-collectives', action='store_true',
                    help='split backward only: capture the gradient collectives INTO the step (RCCL collectives are '
                         'capturable: the all-reduce of the head gradients becomes a branch of the hipGraph that joins '
                         'behind the stack backward) - a rank\'s step is ONE replay instead of two replays with '
                         'host-issued collectives between them.  Off by default: not exercised on more than one GPU yet')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=8)
    ap.add_argument('--kernel-iters', type=int, default=200)
    ap.add_argument('--stream-batch', type=int, default=16384,
                    help='graphs in the streaming-batch run of the north-star kernel (roofline_streaming); 0 = skip')
    ap.add_argument('--dry-cpu', action='store_true',
                    help='LAUNCHER REHEARSAL, not a measurement: the same step() on CPU tensors through the host '
                         'SIMT emulation of the kernels (tools/simt, test hook) with the gloo backend; the line it '
                         'prints carries "invalid"')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own ranks
# ---------------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """One child process per GPU through torch.distributed.run (which sets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*), started BEFORE this process makes any GPU call; the children are fresh processes, nothing
    is exec'd over a process that has touched the GPU.  Rank 0's JSON line goes to our stdout unchanged;
    the exit code is the launcher's (non-zero if any rank failed)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, host_cores() // max(1, args.gpus))))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    log('starting %d ranks: %s' % (args.gpus, ' '.join(cmd[2:])))
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------

def make_batch(args, rank, dev):
    n_max = min(args.n_pad, D.SHAPES[args.shape][1])
    n_min = min(D.SHAPES[args.shape][0], n_max)
    ds = D.SyntheticGraphDataset(args.shape, args.batch, in_dim=args.dim, seed=rank, n_min=n_min, n_max=n_max)
    batch9, cache = D.collate(ds.samples, k_eig=args.k_eig, n_pad=args.n_pad)
    x, mask, pe, _, degree, _, edge_index, batch, fi = batch9
    if getattr(args, 'no_pe', False):
        pe = None
    src = x.permute(1, 0, 2).contiguous()          # [N,B,d] seq-first, embedding skipped
    g = torch.Generator().manual_seed(1000 + rank)
    dout = torch.randn(src.shape, generator=g)
    cpu = dict(src=src, mask=mask, pe=pe, degree=degree, edge_index=edge_index, batch=batch, fi=fi,
               dout=dout, cache=cache)
    gpu = {k: (v.to(dev) if v is not None else None) for k, v in cpu.items()}
    if args.dtype == 'bf16':
        # bf16 STORAGE: the resident batch holds its token rows and the positional kernel in the storage type (what
        # a collate for this path emits); everything else (degree, U / lambda, node counts, upstream gradient) as before
        gpu['src'] = gpu['src'].to(torch.bfloat16)
        gpu['pe'] = None if gpu['pe'] is None else gpu['pe'].to(torch.bfloat16)
    return cpu, gpu


def build_encoder(args, filter_mode=None, share=None):
    torch.manual_seed(0)
    layer = DiffTransformerEncoderLayer(args.dim, args.heads, 2 * args.dim, 0.0,
                                        batch_norm=not args.layer_norm)
    enc = DiffTransformerEncoderGenGCN(args.dim, args.heads, layer, args.layers,
                                       num_coefficients=args.order,
                                       heads_share_graph=(not args.no_share_graph) if share is None else share,
                                       filter_mode=args.filter_mode if filter_mode is None else filter_mode)
    if args.dtype == 'bf16' and enc.filter_mode == 'spectral':
        from feta_tmlr_amd.transformer.layers import set_storage_dtype
        set_storage_dtype(enc, torch.bfloat16)
    return enc


def _sync(dev):
    if dev.type == 'cuda':
        torch.cuda.synchronize()


def make_step(args, enc, gpu, world, dev):
    """-> step(): one forward + backward (+ gradient all-reduce for world > 1) of `enc` on the resident batch,
    as one hipGraph replay (two for the split backward) unless --no-graph / --dry-cpu."""
    params = [p for p in enc.parameters()]
    lowp = getattr(enc, 'storage_dtype', torch.float32) != torch.float32
    # the split backward needs the fused stack (fp32, or the bf16 instantiations of its kernels where the shape has them)
    fused_lowp = False
    if lowp:
        from feta_tmlr_amd import _lib
        from feta_tmlr_amd.fused_stack import lowp_stack_supported, stack_supported
        n_, b_, d_ = gpu['src'].shape
        fused_lowp = (bool(getattr(enc, 'fused_stack', False)) and stack_supported(enc.layers, d_)
                      and lowp_stack_supported(_lib.backend(gpu['src'])[0], enc.layers, n_, b_, d_))
    two_phase = (args.two_phase or world > 1) and (not lowp or fused_lowp)
    fwd_args = (gpu['src'], gpu['pe'], gpu['edge_index'], gpu['fi'], gpu['batch'])
    fwd_kw = dict(degree=gpu['degree'], src_key_padding_mask=gpu['mask'], graph_cache=gpu['cache'])
    use_graph = not args.no_graph and dev.type == 'cuda'

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
        with torch.cuda.graph(g):
            fn()
        return g.replay

    def settle(run, n=40):
        """Set-up, not measurement: a freshly instantiated graph replays ~3 % slower for its first tens of replays (clocks
        and caches coming out of the capture's idle time) - a 20-step run read 0.273 ms where 200 steps read 0.264.  The
        W warm-up steps of the contract follow these."""
        if use_graph:
            for _ in range(n):
                run()
            _sync(dev)

    if not two_phase:
        # fresh .grad per step, one flat bucket for RCCL (a bf16 wire bucket on the bf16 storage path)
        reducer = FlatGradAllReduce(params, world, bucket_dtype=torch.bfloat16 if lowp else None)

        held = {}

        def fwd_bwd():
            reducer.zero()
            out, _, _ = enc(*fwd_args, **fwd_kw)
            held['out'] = out.detach()           # (static address across replays: read by the parity test)
            out.backward(gradient=gpu['dout'])   # upstream gradient dOut ~ N(0,1) injected directly

        fwd_bwd()
        _sync(dev)
        run = capture(fwd_bwd)
        settle(run)

        def step():
            run()
            if world > 1:
                reducer.all_reduce()
        step.held = held
        return step, two_phase, use_graph

    # split backward: the filter-stage gradients (96 % of the bytes) are ready first; their
    # all-reduce runs under the backward of the encoder stack
    enc.keep_stack_boundary = True
    # head: the two 4 MB gradients in place + one small packed bucket, all under the stack backward;
    # stack: its gradients already live in one flat buffer - one in-place collective, no pack / unpack
    r_head = HybridGradAllReduce(enc.head_parameters(), world)
    r_stack = FlatBufferAllReduce(enc.stack_flat_grad, world, params=enc.stack_parameters())

    held = {}

    def phase1():
        r_head.zero()
        for p_ in enc.stack_parameters():
            p_.grad = None
        out, _, _ = enc(*fwd_args, **fwd_kw)
        held['out'] = out.detach()
        enc.backward_head(out, gpu['dout'])

    def both():
        phase1()
        enc.backward_stack()

    both()
    _sync(dev)
    if use_graph and getattr(args, 'graph_collectives', False):
        # ONE graph per step: phase 1, the head collectives (launched on RCCL's stream: a captured fork), the stack
        # backward under them, the stack collective, and the joins (work.wait() inside the capture = an event wait of the
        # capturing stream).  The communicators exist before the capture: the warm-up steps below run the collectives
        # eagerly.  At world = 1 start() / finish() are no-ops and this is the single-pass step as one replay.
        def whole():
            phase1()
            w1 = r_head.start()
            enc.backward_stack()
            w2 = r_stack.start()
            r_head.finish(w1)
            r_stack.finish(w2)

        run = capture(whole)
        settle(run)

        def step():
            run()
        step.held = held
        return step, two_phase, use_graph
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
        settle(lambda: (run1(), run2()))
    else:
        run1, run2 = phase1, enc.backward_stack

    def step():
        run1()
        w1 = r_head.start()
        run2()
        w2 = r_stack.start()
        r_head.finish(w1)
        r_stack.finish(w2)
    step.held = held
    return step, two_phase, use_graph


def grad_bucket_bytes(enc, two_phase, dtype):
    """Bytes one rank hands to the gradient collectives per step: row-constant gradients (gcn.weight) travel as one row;
    the split backward reduces the two C x C gradients in place and the stack's flat buffer in place, the single-pass
    form packs everything into one bucket (bf16 on the wire for the bf16 storage leg)."""
    head = stack = 0
    stack_ids = {id(p) for p in enc.stack_parameters()}
    for p in enc.parameters():
        if not p.requires_grad:
            continue
        n = p.shape[1] if (getattr(p, '_feta_row_constant', False) and p.dim() == 2) else p.numel()
        if id(p) in stack_ids:
            stack += n
        else:
            head += n
    if two_phase:
        return {'head': 4 * head, 'stack': 4 * stack}
    return {'all': (2 if dtype == 'bf16' else 4) * (head + stack)}


def time_steps(step, args, world, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(args.warmup):
        step()
    _sync(dev)
    if world > 1:
        dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _sync(dev)
    if world > 1:
        dist.barrier()
    _sync(dev)
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item())


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


def spec_bytes(b, n, h, dh, k_eig, p, sum_n, backward):
    """Algorithmic HBM bytes of one launch of the eigenbasis filter kernels: operand rows
    of REAL nodes only (the kernels clamp their loads to n_real; sum_n = sum of node counts), whole padded
    rows for the result (padded rows are written as zeros)."""
    d, c = h * dh, p * dh * dh
    if not backward:     # read x, U, lambda, W; write y
        return 4 * (sum_n * d + sum_n * k_eig + b * k_eig + b * h * c + b * n * d)
    # read x, dy, U, lambda, W; write dx, dW (+ per-block bias partials)
    return 4 * (2 * sum_n * d + sum_n * k_eig + b * k_eig + b * h * c + b * n * d + b * h * c + b * h * dh)


def _load_json(*names):
    """-> (content, 'profiles/<name>') of the first committed profile summary that exists, else ({}, None)"""
    for nm in names:
        p = os.path.join(ROOT, 'profiles', nm)
        if os.path.exists(p):
            try:
                return json.load(open(p)), 'profiles/' + nm
            except Exception:
                pass
    return {}, None


def roofline(args, gpu, dev, lowp=False, pmc_tag=None):
    """Times the hand-written kernels of one step in isolation, in exactly the variants the fused stack
    issues (feta_tmlr_amd/benchcases.py; HIP events on the stream they are launched on - torch's
    current stream), groups the variants by KERNEL SYMBOL (what rocprofv3 --stats reports: e.g. the two
    attention-block variants, with and without the attn write, are one symbol), picks the DOMINANT symbol =
    largest sum over variants of launches-per-step x launch time, and prices its average launch against the
    HBM roofline with its average algorithmic bytes.  `traffic` = HBM bytes per launch
    from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE,
    tools/pmc_summary.py), `mfma_busy` = MFMA-pipe busy share of that symbol (tools/pmc_mfma.py); null if
    absent."""
    from feta_tmlr_amd.benchcases import stack_layer_cases
    abi, st = _lib.abi(), _lib.stream_handle()
    b, n, h, d = args.batch, args.n_pad, args.heads, args.dim
    dh, k_eig, p = d // h, args.k_eig, args.order
    c = p * dh * dh
    L = args.layers
    nr = gpu['cache'].n_real
    sum_n = int(nr.sum().item())
    rnd = lambda *s: torch.randn(*s, device=dev)
    cand = []   # (variant name, symbol, launches per step, fn, algorithmic bytes)
    from feta_tmlr_amd import fused_stack
    split_form = (fused_stack.USE_ATTN_BLOCK_SPLIT and fused_stack.USE_FFN_BWD and fused_stack.USE_ATTN_BLOCK_BWD
                  and abi.ffn_bwd_supported(d, 2 * d) and abi.attn_block_bwd_supported(n, d, h)
                  and abi.attn_block_bwd_blocks(b) > 0)
    coeff_roles = args.filter_mode in ('spectral', 'cheb') and not args.two_phase and args.gpus == 1 and p == 4
    from feta_tmlr_amd import functional as FF0
    cat_fold = (FF0.USE_CAT_FOLD and not lowp and args.filter_mode == 'spectral' and not args.no_share_graph
                and abi.spec_cat_supported(n, h, dh, p, k_eig, True))
    cat_bwd_fold = cat_fold and FF0.USE_CAT_FOLD_BWD and abi.spec_cat_bwd_supported(n, h, dh, p, k_eig, True)
    for name, per_layer, fn, nbytes, syms in stack_layer_cases(abi, st, dev, b, n, d, h, 2 * d, gpu['pe'], nr,
                                                               dtype=torch.bfloat16 if lowp else torch.float32):
        if name == 'attn_block_fwd (no attn write)':
            cnt = L - 1
        elif name == 'attn_block_fwd (+attn write)':
            cnt = 1
        elif 'linear_cat' in name:
            cnt = 0 if cat_bwd_fold else 1      # (ABI 11: linear_cat's backward rides in the filter's backward launch)
        elif name in ('ffn_bwd (gradient in two parts)', 'attn_block_bwd (two workgroups per graph)'):
            cnt = L - 1      # fused_stack.py: every layer but the first hands its input gradient to a fused FFN backward
        elif name == 'attn_block_bwd' and split_form:
            cnt = 1
        elif name in ('ffn_fwd (+ coefficient generator)', 'ffn_bwd (+ coefficient generator)'):
            cnt = 1 if coeff_roles else 0      # the last layer's launches (functional.PendingSums)
        elif name == 'ffn_fwd':
            cnt = L - 1 if coeff_roles else L
        elif name == 'ffn_bwd':
            cnt = (0 if coeff_roles else 1) if split_form else (L - 1 if coeff_roles else L)
        else:
            cnt = L
        if cnt == 0:
            continue
        # candidates are grouped by SOURCE kernel: the one- and two-workgroup instantiations of attn_block_bwd are one
        # kernel, as are the variants of the others
        sym = syms[0].split('<')[0]
        if sym == 'rowlin_bwd':   # one template instantiation (= one symbol) per (KI, NO)
            sym = 'rowlin_bwd<%s>' % name.split()[1]
        cand.append((name, sym, cnt, fn, nbytes))
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
    if not (abi.attn_block_bwd_supported(n, d, h) and abi.attn_block_bwd_blocks(b) > 0):
        # (the fused attention-block backward of benchcases replaces it where the shape allows)
        cand.append(('attn_bwd (dq + dkdv)', 'attn_bwd', L,
                     lambda: abi.attn_bwd(q, k, v, gpu['pe'], nr, out, dout, stats, delta, dq, dk, dv, sc, st),
                     4 * b * (3 * n * d + 2 * n * d + n * n + 2 * h * n + 3 * n * d + h * n)))
    if not abi.attn_block_supported(n, d, h) and abi.attn_out_supported(n, d, h) and not lowp:
        # graphs beyond the one-launch block (config 4): attention core + out_proj + residual + statistics behind in_proj
        xo, wo_, bo_ = rnd(n * b, d), rnd(d, d) / d ** 0.5, rnd(d)
        oo, yo = torch.empty(n * b, d, device=dev), torch.empty(n * b, d, device=dev)
        sto = torch.empty(abi.attn_out_stat_rows(b, n) + 1, 2, d, device=dev)
        deg_o = torch.rand(n * b, device=dev)
        cand.append(('attn_out_fwd', 'attn_out_fwd', L,
                     lambda: abi.attn_out_fwd(b, n, sc, st, x=xo, w_out=wo_, b_out=bo_, pe=gpu['pe'], n_real=nr,
                                              rowscale=deg_o, qkv=qkv, out=oo, attn_stats=stats, attn=None, y=yo, y_stats=sto),
                     4 * b * (3 * n * d + n * n + n * d + 2 * n * d + 2 * h * n) + 4 * d * d))
    if args.filter_mode == 'spectral' and not args.no_share_graph:
        xs, dys = rnd(n, b, h, dh).permute(1, 0, 2, 3), rnd(n, b, h, dh).permute(1, 0, 2, 3)
        ys, dxs = tok(), tok()
        coeff, bias = rnd(h * b, c), rnd(dh)
        dcoeff, dbp = torch.empty_like(coeff), torch.empty(b * h, dh, device=dev)
        u, lam = gpu['cache'].u, gpu['cache'].lam
        if cat_fold:
            # what the step launches since ABI 10: the filter with linear_cat folded in (+ the stack output rows in, the
            # linear_cat output rows out, W_cat from L2)
            y2c, wc, bc_, oc = rnd(n, b, d), rnd(d, 2 * d) / (2 * d) ** 0.5, rnd(d), torch.empty(n, b, d, device=dev)
            cand.append(('spec_filter_cat_fwd', 'spec_cat_fwd', 1,
                         lambda: abi.spec_filter_cat_fwd(xs, u, lam, coeff, bias, nr, ys, p, 1, st,
                                                         y2c.view(n, b, h, dh).permute(1, 0, 2, 3), wc, bc_,
                                                         oc.view(n, b, h, dh).permute(1, 0, 2, 3)),
                         spec_bytes(b, n, h, dh, k_eig, p, sum_n, False) + 4 * (2 * sum_n * d + 2 * d * d + d)))
        else:
            cand.append(('spec_filter_fwd', 'spec_fwd', 1,
                         lambda: abi.spec_filter_fwd(xs, u, lam, coeff, bias, nr, ys, p, 1, st),
                         spec_bytes(b, n, h, dh, k_eig, p, sum_n, False)))
        if cat_bwd_fold:
            # what the step launches since ABI 11: the filter's backward with linear_cat's backward inside (+ dout, y2 and
            # filt rows in, dxn rows out, one [64 x 128 + 64] partial row per workgroup)
            rows_ = abi.spec_cat_bwd_rows(b)
            tv_ = lambda t: t.view(n, b, h, dh).permute(1, 0, 2, 3)
            dob, y2b, dxnb = rnd(n, b, d), rnd(n, b, d), torch.empty(n, b, d, device=dev)
            wcb = rnd(d, 2 * d) / (2 * d) ** 0.5
            partb = torch.empty(rows_, d * 2 * d + d, device=dev)
            cand.append(('spec_filter_cat_bwd', 'spec_cat_bwd', 1,
                         lambda: abi.spec_filter_cat_bwd(xs, u, lam, coeff, nr, ys, dxs, dcoeff, dbp, p, 1, st, dout=tv_(dob),
                                                         y2=tv_(y2b), w_cat=wcb, dxn=tv_(dxnb), partial=partb),
                         spec_bytes(b, n, h, dh, k_eig, p, sum_n, True) + 4 * (4 * sum_n * d + 2 * d * d + rows_ * (2 * d * d + d))))
        else:
            cand.append(('spec_filter_bwd', 'spec_bwd', 1,
                         lambda: abi.spec_filter_bwd(xs, u, lam, coeff, nr, dys, dxs, dcoeff, dbp, p, 1, st),
                         spec_bytes(b, n, h, dh, k_eig, p, sum_n, True)))
    r_ = h * b
    from feta_tmlr_amd import functional as FF
    if (abi.lin_supported(r_, c, c) and (r_ * c * c <= FF.LIN_OWN_GEMM_MAX_MACS or lowp)
            and not (lowp and r_ >= FF.LIN_LIB_BF16_MIN_ROWS)):     # (functional.FilterFromPooledFn's policy: from that
        # many rows the bf16 legs run library GEMMs too - the object lists what the step launches)
        # the C x C linear of the coefficient generator runs as csrc/lin.hip at this size (else: library GEMMs, which
        # are not candidates - the roofline object prices the hand-written kernels)
        lw, lb, lx, ldy = rnd(c, c) / c ** 0.5, rnd(c), rnd(r_, c), rnd(r_, c)
        ly, ldx, ldw, ldb = (torch.empty(r_, c, device=dev), torch.empty(r_, c, device=dev),
                             torch.empty(c, c, device=dev), torch.empty(c, device=dev))
        cat_part = rnd(abi.rowlin_chunks(b * n), d * 2 * d + d)
        pairs = [(rnd(r_, dh), torch.empty(dh, device=dev)), (cat_part, torch.empty(cat_part.shape[1], device=dev))]
        cand.append(('lin_fwd', 'lin_fwd', 1, lambda: abi.lin_fwd(lx, lw, lb, ly, st, bf16=lowp), 4 * (2 * r_ * c + c * c)))
        cand.append(('lin_bwd', 'lin_bwd', 1, lambda: abi.lin_bwd(lx, lw, ldy, ldx, ldw, ldb, st, pairs=pairs, bf16=lowp),
                     4 * (3 * r_ * c + 2 * c * c + cat_part.numel())))
    # NOT measured in this run: committed products of separate rocprofv3 --pmc passes (tools/profile_round.sh), named in
    # the object so that nobody takes them for same-run counters
    tfiles = (('r04_end_traffic_b128_bf16.json', 'r03_end_traffic_b128_bf16.json') if lowp else
              ('r04_end_traffic_b128.json', 'r03_end_traffic_b128.json'))
    mfiles = (('r04_end_pmc_mfma_b128_bf16.json', 'r03_end_pmc_mfma_b128_bf16.json') if lowp else
              ('r04_end_pmc_mfma_b128.json', 'r03_end_pmc_mfma_b128.json'))
    if b != 128 or n != 37:      # (the per-variant PMC passes are of the BASELINE batch)
        tfiles = mfiles = ()
    traffic, traffic_src = _load_json(*tfiles)
    mfma, mfma_src = _load_json(*mfiles)
    # other shapes (extra_configs): counters of the STEP itself, per kernel symbol (tools/profile_config.sh -> tools/pmc_step.py:
    # every launch of a symbol in an eager run of this configuration, FETCH_SIZE x2 + WRITE_SIZE, MFMA-pipe busy share)
    step_pmc, step_src = _load_json('r04_%s_pmc.json' % pmc_tag) if pmc_tag else ({}, None)
    groups = {}
    for name, sym, cnt, fn, nbytes in cand:
        t = time_kernel(fn, args.kernel_iters)
        tr = (traffic.get(name) or {}).get('hbm_bytes')
        row = {'variant': name, 'launches_per_step': cnt, 'launch_us': round(t * 1e6, 3),
               'algorithmic_bytes': nbytes, 'traffic': tr}
        groups.setdefault(sym, []).append(row)
    out_rows = []
    for sym, rows in groups.items():
        launches = sum(r['launches_per_step'] for r in rows)
        t_step = sum(r['launches_per_step'] * r['launch_us'] for r in rows)          # us of this symbol per step
        bytes_step = sum(r['launches_per_step'] * r['algorithmic_bytes'] for r in rows)
        tr = None
        if all(r['traffic'] is not None for r in rows):
            tr = int(sum(r['launches_per_step'] * r['traffic'] for r in rows) / launches)
        busy = None
        base = sym.partition('<')[0]
        hits = [val.get('mfma_busy_pct') for key, val in mfma.items()      # keys: kernel names with template arguments
                if base in key and ('bf16' in key) == lowp and val.get('mfma_busy_pct') is not None]
        if hits:
            busy = round(sum(hits) / len(hits), 2)      # mean over the instantiations of the source kernel
        wait = None
        if step_pmc:
            # (the fused kernels of the stack and the filter stage, grouped by source kernel; row-wise symbols are not matched)
            import re
            pat = re.compile(r'feta::%s(8|_graph|_head|_dense)?_kernel' % re.escape(base))
            ks = [(key, val) for key, val in step_pmc.items()
                  if '<' not in sym and pat.search(key) and ('bf16' in key) == lowp and 'hbm_bytes' in val]
            if ks:
                wsum = sum(v_['launches'] for _, v_ in ks)
                tr = int(sum(v_['launches'] * v_['hbm_bytes'] for _, v_ in ks) / wsum)
                if all('mfma_busy_pct' in v_ for _, v_ in ks):
                    busy = round(sum(v_['launches'] * v_['mfma_busy_pct'] for _, v_ in ks) / wsum, 2)
                    wait = round(sum(v_['launches'] * v_['wait_any_pct_of_wave_cycles'] for _, v_ in ks) / wsum, 1)
        ach = bytes_step / t_step / 1e3       # bytes / us -> GB/s
        out_rows.append({'kernel': sym, 'launches_per_step': launches, 'launch_us': round(t_step / launches, 3),
                         'us_per_step': round(t_step, 2), 'algorithmic_bytes': int(bytes_step / launches),
                         'achieved': round(ach, 2), 'frac': round(ach / HBM_PEAK_GBS, 5), 'traffic': tr,
                         'mfma_busy_pct': busy, 'wave_wait_pct': wait, 'variants': rows})
    dom = max(out_rows, key=lambda r: r['us_per_step'])
    res = {'kernel': dom['kernel'], 'bound': 'hbm', 'achieved': dom['achieved'], 'peak': HBM_PEAK_GBS,
           'unit': 'GB/s', 'frac': dom['frac'], 'traffic': dom['traffic'],
           'algorithmic_bytes': dom['algorithmic_bytes'], 'launch_us': dom['launch_us'],
           'launches_per_step': dom['launches_per_step'], 'us_per_step': dom['us_per_step'],
           'mfma_busy_pct': dom['mfma_busy_pct'], 'wave_wait_pct': dom['wave_wait_pct'],
           'traffic_source': step_src or traffic_src, 'mfma_busy_source': step_src or mfma_src,
           'timing': 'HIP events in this run, launch stream', 'variants': dom['variants'],
           'other_kernels': [{k_: v_ for k_, v_ in r.items() if k_ != 'variants'} for r in out_rows if r is not dom]}
    return res


def roofline_streaming(args, dev):
    """The north-star kernel (eigenbasis filter: U^T X -> g(Lambda) -> U) at a batch whose working
    set (~0.6 GB) does not fit the 256 MB Infinity Cache, so that HBM is what is measured; same
    kernel, same shape per graph as the BASELINE batch.  HIP events on the launch stream; algorithmic bytes
    count operand rows of real nodes only (spec_bytes)."""
    abi, st = _lib.abi(), _lib.stream_handle()
    b, n, h, d = args.stream_batch, args.n_pad, args.heads, args.dim
    dh, k_eig, p = d // h, args.k_eig, args.order
    c = p * dh * dh
    g = torch.Generator(device='cpu').manual_seed(1)
    nr_cpu = torch.randint(9, n + 1, (b,), generator=g, dtype=torch.int32)
    sum_n = int(nr_cpu.sum())
    nr = nr_cpu.to(dev)
    rnd = lambda *s: torch.randn(*s, device=dev)
    xs, dys = (rnd(n, b, h, dh).permute(1, 0, 2, 3) for _ in range(2))
    ys, dxs = (torch.empty(n, b, h, dh, device=dev).permute(1, 0, 2, 3) for _ in range(2))
    u, lam = rnd(b, n, k_eig), torch.rand(b, k_eig, device=dev) * 2 - 1
    coeff, bias = rnd(h * b, c), rnd(dh)
    dcoeff, dbp = torch.empty_like(coeff), torch.empty(b * h, dh, device=dev)
    out = []
    for name, fn, nbytes in (
            ('spec_filter_fwd', lambda: abi.spec_filter_fwd(xs, u, lam, coeff, bias, nr, ys, p, 1, st),
             spec_bytes(b, n, h, dh, k_eig, p, sum_n, False)),
            ('spec_filter_bwd', lambda: abi.spec_filter_bwd(xs, u, lam, coeff, nr, dys, dxs, dcoeff, dbp, p, 1, st),
             spec_bytes(b, n, h, dh, k_eig, p, sum_n, True))):
        t = time_kernel(fn, 20)
        out.append({'kernel': name, 'batch': b, 'bound': 'hbm', 'launch_us': round(t * 1e6, 2),
                    'algorithmic_bytes': nbytes, 'achieved': round(nbytes / t / 1e9, 1), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4)})
    return out


def cpu_baseline(args, cpu, enc, spectral, share, primary=True):
    """CPU restatement of the SAME operator as the GPU leg it is reported beside, PyTorch CPU fp32, all host
    cores: the reference's algorithm for everything the reference has text for (attention layer as
    reconstructed, the un-collapsed GCNConv on ones with its Python loop over the H*B blocks, head stacking,
    linear_cat) and, for the filter stage, either the reference's edge-list recursion with per-node weight
    copies (spectral=False: the reference-literal operator) or the K-eigenpair eigenbasis form the GPU leg
    computes (spectral=True; not a reference operator unless K spans the graph)."""
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
    eig = (cpu['cache'].u[:sub], cpu['cache'].lam[:sub]) if spectral else None

    def step(collapsed=False):
        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        src = cpu['src'][:, :sub].clone().requires_grad_(True)
        out, _, _ = O.encoder_gengcn(src, None if cpu['pe'] is None else cpu['pe'][:sub], ei, cpu['fi'][:n_tot], cpu['batch'][:n_tot],
                                     cpu['degree'][:sub], m, leaves, args.layers, args.heads, args.order,
                                     batch_norm=not args.layer_norm, heads_share_graph=share,
                                     collapsed=collapsed, eig=eig)
        (out * cpu['dout'][:, :sub]).sum().backward()

    def timed(collapsed, warm, steps):
        """warm-up steps, then the MEDIAN of the timed steps"""
        for _ in range(warm):
            step(collapsed)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step(collapsed)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    warm, steps = (5, 20) if primary else (1, max(3, args.cpu_steps // 2))
    if args.cpu_steps != 8:     # an explicit --cpu-steps overrides the protocol (short runs)
        warm, steps = 1, args.cpu_steps
    dt = timed(False, warm, steps)
    log('cpu baseline (%s) timed' % ('eigenbasis K=%d' % args.k_eig if spectral else 'edge-list recursion'))
    # the same restatement with the exact GCNConv(ones) = c_j colsum(W) + b collapse, so that
    # the GPU/CPU ratio is not inflated by the reference's redundant ones @ W product alone
    dt_opt = timed(True, 1, max(3, steps // 4))
    op = ('K=%d eigenbasis filter, every head on the graph' % args.k_eig) if spectral else \
        'order-%d Chebyshev edge-list recursion' % args.order
    if not spectral or not share:
        op += ', heads_share_graph=%s' % share
    return {'value': round(sub / dt, 2), 'unit': 'graphs/s', 'cores': cores, 'kind': 'port',
            'sample': '%d warm-up + %d timed steps (median) of fwd+bwd on the %d graphs of the batch '
                      '(oracle.encoder_gengcn: %s; un-collapsed coefficient generator; torch CPU fp32, %d threads)'
                      % (warm, steps, sub, op, cores),
            'optimised_value': round(sub / dt_opt, 2),
            'optimised_sample': 'same, with the collapsed coefficient generator (no ones @ W, no dense edge list)'}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args, argv))       # nothing above touches a GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    hook = None
    if args.dry_cpu:
        import ctypes
        from feta_tmlr_amd import _abi
        dev = torch.device('cpu')
        hook = _lib.override_for_tests(_abi.bind(ctypes.CDLL(os.path.join(ROOT, 'tools', 'simt', 'libfeta_emu.so'))))
        hook.__enter__()
        if world > 1:
            dist.init_process_group('gloo')
    else:
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
    step, two_phase, use_graph = make_step(args, enc, gpu, world, dev)
    log('graph captured' if use_graph else 'eager mode')
    dt = time_steps(step, args, world, dev)
    log('timed region done: %.3f ms/step' % (dt / args.steps * 1e3))

    # beside the headline leg (N = 1, fp32 run only): the bf16 storage leg of BASELINE configs 3 / 5 on the same batch,
    # and the split backward --gpus N uses, so that the first SCALE run has its single-GPU baseline
    extra = {}
    if world == 1 and not args.dry_cpu and args.dtype == 'f32' and not args.no_literal:
        import copy
        a16 = copy.copy(args)
        a16.dtype = 'bf16'
        _, gpu16 = make_batch(a16, rank, dev)
        enc16 = build_encoder(a16).to(dev)
        enc16.train()
        step16, _, _ = make_step(a16, enc16, gpu16, world, dev)
        dt16 = time_steps(step16, args, world, dev)
        log('bf16 storage leg: %.3f ms/step' % (dt16 / args.steps * 1e3))
        extra['bf16_leg'] = {'value': round(args.batch * args.steps / dt16, 2), 'unit': 'graphs/s',
                             'ms_per_step': round(dt16 / args.steps * 1e3, 4), 'dtype': 'bf16',
                             'what': 'the same step on bf16 storage (python bench.py --dtype bf16): fused stack on '
                                     'v_mfma_f32_16x16x16_bf16, fp32 statistics / parameter gradients / filter stage',
                             'roofline': roofline(a16, gpu16, dev, lowp=True)}
        if not two_phase:
            a2 = copy.copy(args)
            a2.two_phase = True
            enc2 = build_encoder(a2).to(dev)
            enc2.train()
            step2, _, _ = make_step(a2, enc2, gpu, world, dev)
            dt2 = time_steps(step2, args, world, dev)
            log('split backward at N = 1: %.3f ms/step' % (dt2 / args.steps * 1e3))
            extra['two_phase_n1'] = {'value': round(args.batch * args.steps / dt2, 2), 'unit': 'graphs/s',
                                     'ms_per_step': round(dt2 / args.steps * 1e3, 4),
                                     'what': 'python bench.py --two-phase: the backward --gpus N > 1 runs (head '
                                             'gradients first, their all-reduce under the stack backward), on one GPU'}

    # BASELINE configs 4 and 5 as further objects of the same line (parity-test shapes, not the metric): PATTERN
    # (B = 64, N_pad = 128, K = 32; experiments/run_transformer_gengcn_SBM_cv.py) and the molhiv bucket (B = 1024,
    # N_pad = 64, bf16 storage), each timed like the headline leg and with the roofline object of its own dominant kernel
    if (world == 1 and not args.dry_cpu and args.dtype == 'f32' and not args.no_literal and args.shape == 'zinc'
            and args.batch == 128):
        import copy
        extra['extra_configs'] = []
        ref_default = dict(layer_norm=True, no_pe=True)    # what README.md:49,65,71 run: no --batch-norm, no --pos-enc
        for name, kw, with_roofline in (
                ('config 4: PATTERN-shaped, B=64, N_pad=128, K=32, fp32',
                 dict(shape='pattern', batch=64, n_pad=128, k_eig=32, dtype='f32'), 'pattern'),
                ('config 4 (reference defaults: LayerNorm, pe=None): PATTERN-shaped, B=64, N_pad=128, K=32, fp32',
                 dict(shape='pattern', batch=64, n_pad=128, k_eig=32, dtype='f32', **ref_default), None),
                ('config 4 at the largest PATTERN graph: B=64, N_pad=188, K=32, fp32',
                 dict(shape='pattern', batch=64, n_pad=188, k_eig=32, dtype='f32'), None),
                ('config 5: molhiv-shaped, B=1024, N_pad=64 bucket, K=16, bf16 storage',
                 dict(shape='molhiv', batch=1024, n_pad=64, k_eig=16, dtype='bf16'), 'molhiv_bf16'),
                ('config 5 (reference defaults: LayerNorm, pe=None): molhiv-shaped, B=1024, N_pad=64 bucket, K=16, bf16 storage',
                 dict(shape='molhiv', batch=1024, n_pad=64, k_eig=16, dtype='bf16', **ref_default), None),
                ('config 5 (reference defaults: LayerNorm, pe=None), fp32',
                 dict(shape='molhiv', batch=1024, n_pad=64, k_eig=16, dtype='f32', **ref_default), None),
                ('config 1 (reference defaults: LayerNorm, pe=None): MUTAG-shaped, B=32, N_pad=28, K=8, fp32',
                 dict(shape='mutag', batch=32, n_pad=28, k_eig=8, dtype='f32', **ref_default), None)):
            ax = copy.copy(args)
            for k_, v_ in kw.items():
                setattr(ax, k_, v_)
            ax.steps, ax.warmup, ax.kernel_iters = min(args.steps, 100), min(args.warmup, 10), min(args.kernel_iters, 50)
            _, gpux = make_batch(ax, rank, dev)
            encx = build_encoder(ax).to(dev)
            encx.train()
            stepx, _, _ = make_step(ax, encx, gpux, world, dev)
            dtx = time_steps(stepx, ax, world, dev)
            log('%s: %.3f ms/step' % (name, dtx / ax.steps * 1e3))
            row = {
                'config': name, 'value': round(ax.batch * ax.steps / dtx, 2), 'unit': 'graphs/s',
                'ms_per_step': round(dtx / ax.steps * 1e3, 4), 'steps': ax.steps, 'dtype': ax.dtype,
                'norm': 'layer' if ax.layer_norm else 'batch', 'pe': not ax.no_pe,
                'flags': '--shape %s --batch %d --n-pad %d --k-eig %d --dtype %s%s%s' % (
                    ax.shape, ax.batch, ax.n_pad, ax.k_eig, ax.dtype, ' --layer-norm' if ax.layer_norm else '',
                    ' --no-pe' if ax.no_pe else '')}
            if with_roofline:      # (= the tag of this configuration's committed counters, profiles/r04_<tag>_pmc.json)
                row['roofline'] = roofline(ax, gpux, dev, lowp=ax.dtype == 'bf16', pmc_tag=with_roofline)
            extra['extra_configs'].append(row)
            del gpux, encx, stepx

    literal = None
    if (not args.no_literal and not args.dry_cpu and args.dtype == 'f32'
            and not (args.filter_mode == 'cheb' and args.no_share_graph)):
        # the reference-literal operator on the same batch, same parameters, timed the same way
        enc_lit = build_encoder(args, filter_mode='cheb', share=False).to(dev)
        enc_lit.train()
        step_lit, _, _ = make_step(args, enc_lit, gpu, world, dev)
        dt_lit = time_steps(step_lit, args, world, dev)
        log('reference-literal operator: %.3f ms/step' % (dt_lit / args.steps * 1e3))
        literal = {'value': round(args.batch * world * args.steps / dt_lit, 2), 'unit': 'graphs/s',
                   'ms_per_step': round(dt_lit / args.steps * 1e3, 4),
                   'operator': "filter_mode='cheb' (order-%d Chebyshev recursion, transformer/ChebNetDynamic.py:157-187), "
                               'heads_share_graph=False (un-replicated edge_index, transformer/models.py:186)' % args.order}

    if rank == 0:
        total_graphs = args.batch * world * args.steps
        share = not args.no_share_graph
        res = {
            'metric': 'graphs/sec fwd+bwd, ZINC batch (N<=37,d=64,K=16)',
            'value': round(total_graphs / dt, 2), 'unit': 'graphs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': args.shape.upper() + '-shaped synthetic padded-graph batch, ChebConvDynamic block '
                                   '(attention + coefficient generator + spectral filter), fwd+bwd',
                       'graphs_per_gpu': args.batch, 'global_batch': args.batch * world,
                       'n_pad': args.n_pad, 'd_model': args.dim, 'heads': args.heads,
                       'filter_order': args.order, 'k_eig': args.k_eig if args.filter_mode == 'spectral' else None,
                       'filter_mode': args.filter_mode, 'layers': args.layers,
                       'norm': 'layer' if args.layer_norm else 'batch(per-rank stats)',
                       'heads_share_graph': share, 'hip_graph': bool(use_graph),
                       'backward': 'two-phase (head all-reduce under stack backward)' if two_phase else 'single',
                       'two_phase': bool(two_phase), 'graph_collectives': bool(two_phase and args.graph_collectives),
                       'ranks': dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
                       'grad_bucket_bytes': grad_bucket_bytes(enc, two_phase, args.dtype),
                       'parallelism': 'dp%d' % world},
        }
        res.update(extra)
        if args.dry_cpu:
            res['invalid'] = 'dry-cpu launcher rehearsal on the host emulation of the kernels: not a measurement'
            res['data'] = 'synthetic (dry-cpu)'
        elif args.dtype != 'f32':
            res['note'] = ('bf16 storage leg (BASELINE configs 3 / 5): bf16 token tensors and LDS tiles, fused stack on '
                           'v_mfma_f32_16x16x16_bf16, fp32 statistics / parameter gradients / filter stage; the headline '
                           'line is --dtype f32 (the reference\'s arithmetic)')
            res['roofline'] = roofline(args, gpu, dev, lowp=True)
        else:
            res['roofline'] = roofline(args, gpu, dev)
            if args.stream_batch > 0:
                res['roofline_streaming'] = roofline_streaming(args, dev)
            log('roofline kernels timed')
            if world == 1 and not args.no_cpu_baseline and args.dtype == 'f32':
                res['cpu_baseline'] = cpu_baseline(args, cpu, enc, args.filter_mode == 'spectral', share)
            if literal is not None:
                if world == 1 and not args.no_cpu_baseline:
                    literal['cpu_baseline'] = cpu_baseline(args, cpu, enc, False, False, primary=False)
                res['reference_literal'] = literal
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if hook is not None:
        hook.__exit__(None, None, None)


if __name__ == '__main__':
    main()
