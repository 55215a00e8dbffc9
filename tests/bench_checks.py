"""Parity of EXACTLY what bench.py times: the encoder bench.build_encoder builds, on the batch
bench.make_batch builds, through the step bench.make_step captures (one hipGraph replay per step on the
MI355X), against the fp64 oracle of the same operator - output and every parameter gradient.  Shared by
the emulation suite (small shapes, CPU) and the MI355X suite (the BASELINE configuration)."""
import torch

import bench
import kernel_checks as KC
from oracle import feta_oracle as O


# bf16 storage leg (bench.py --dtype bf16): the whole encoder against the fp64 oracle on the SAME fp32 inputs and
# master weights - so, unlike the kernel-level checks, the rounding of inputs and weights to bf16 is part of the
# error: every tensor of the stack is stored with 8 significant bits.  Output: max-abs relative to max(1, max|ref|).
# Parameter gradients: relative FROBENIUS error per parameter.  Round 3 (fused bf16 stack, every bias / affine column sum
# taken from fp32 values before they are rounded into a tile), measured on the MI355X at the BASELINE batch: 0.4-3.5 %
# for every parameter except the four per layer that sit behind the relu mask in backward - linear1.weight / .bias
# (dW1, db1 = sums of dh = (g2 W2) * [h > 0]) and norm1.weight / .bias (sums of dx1 = g2 + dh W1): 4-11 %.  The cause is
# the mask, not a reduction: with 8-bit operands ~400 of the 606 k pre-activations change sign against the fp64 run,
# and each flip moves whole row contributions (tools/relu_flip_probe.py isolates it: 6 / 4 / 6 / 4 % from the flips alone,
# 0.3-0.5 % from all roundings together once the mask is exact).  Any bf16 implementation has this; the bound for those
# four stays loose, everything else is held to 5 %.
BF16_MODEL_TOL = 3e-2
BF16_GRAD_FRO_TOL = 5e-2
BF16_GRAD_FRO_TOL_RELU = 1.5e-1      # linear1.*, norm1.*: behind the relu mask


def bf16_grad_tol(name):
    return BF16_GRAD_FRO_TOL_RELU if ('.linear1.' in name or '.norm1.' in name) else BF16_GRAD_FRO_TOL


def rel_fro(got, ref):
    d = (got.detach().double().cpu() - ref.detach().double().cpu()).norm()
    return float(d / ref.detach().double().norm().clamp(min=1e-30))


def check_bench_step(dev, run_ctx, argv, filter_mode=None, share=None, replays=2, two_phase=False,
                     out_tol=KC.TOL, grad_tol=3e-5):
    args = bench.parse(list(argv) + (['--two-phase'] if two_phase else []))
    lowp = args.dtype == 'bf16'
    if lowp:
        out_tol = BF16_MODEL_TOL
    cpu, gpu = bench.make_batch(args, 0, dev)
    enc = bench.build_encoder(args, filter_mode=filter_mode, share=share).to(dev)
    enc.train()
    with torch.no_grad():   # the zero-initialised biases would hide errors in their paths
        enc.spectral_gnns.bias.normal_(0, 0.1)
        enc.gcn.bias.normal_(0, 0.1)
    p64 = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in enc.state_dict().items()
           if v.dtype.is_floating_point}
    from feta_tmlr_amd import fused_stack
    fused_stack.CAPTURE_SAVED = captured = []
    try:
        with run_ctx():
            step, _, used_graph = bench.make_step(args, enc, gpu, 1, dev)
            for _ in range(replays):
                step()
            if dev.type == 'cuda':
                torch.cuda.synchronize()
    finally:
        fused_stack.CAPTURE_SAVED = None
    out = step.held['out'].detach().cpu()
    mode = enc.filter_mode
    eig = (cpu['cache'].u.double(), cpu['cache'].lam.double()) if mode == 'spectral' else None
    def oracle_pass(relu_force=None, capture=None):
        for v in p64.values():
            v.grad = None
        ref_, _, _ = O.encoder_gengcn(cpu['src'].double(), None if cpu['pe'] is None else cpu['pe'].double(),
                                      cpu['edge_index'], cpu['fi'], cpu['batch'], cpu['degree'].double(), cpu['mask'], p64,
                                      args.layers, args.heads, args.order, batch_norm=not args.layer_norm,
                                      heads_share_graph=enc.heads_share_graph, collapsed=True, eig=eig,
                                      relu_capture=capture, relu_force=relu_force)
        (ref_ * cpu['dout'].double()).sum().backward()
        return ref_.detach()

    zs = []
    ref = oracle_pass(capture=zs)
    # pre-activations of REAL nodes that are zero to fp32 resolution: the relu derivative there is a choice (oracle.
    # encoder_layer, relu_force) - the product may land on either side, and then differs from the reference by that
    # node's whole contribution (seen on the MI355X at PATTERN B = 64: |z| = 8e-8 in the last layer moved dW1 of one
    # hidden unit by 0.3 and every gradient below it by 1e-4).  The forward output and the tolerances are untouched: the
    # gradients must meet `grad_tol` for ONE assignment of those (few) derivatives, each evaluated by the fp64 oracle.
    real = (~cpu['mask']).t().unsqueeze(-1)
    ambiguous = [(li, idx) for li, z in enumerate(zs) for idx in ((z.abs() < RELU_AMBIGUOUS) & real).nonzero().tolist()]
    choices = [None]
    if ambiguous and not lowp:
        import itertools
        nb = cpu['src'].shape[1]
        h_saved = captured[-1] if (captured and len(captured[-1]) == len(zs) and 'h' in captured[-1][0]) else None
        if h_saved is not None:
            # the fused stack ran: its saved relu output says which side the kernels took (the tensors of the LAST forward:
            # a replayed graph writes them in place)
            signs = [1 if float(h_saved[li]['h'][idx[0] * nb + idx[1], idx[2]]) > 0.0 else -1 for li, idx in ambiguous]
            sign_sets = [signs]
        else:
            assert len(ambiguous) <= 4, 'too many ambiguous relu derivatives for an exhaustive check: %d' % len(ambiguous)
            sign_sets = list(itertools.product((1, -1), repeat=len(ambiguous)))
        choices = []
        for signs in sign_sets:
            force = {}
            for (li, idx), sg in zip(ambiguous, signs):
                force.setdefault(li, torch.zeros_like(zs[li]))[tuple(idx)] = sg
            choices.append(force)
    failures = []
    for force in choices:
        if force is not None:
            oracle_pass(relu_force=force)
        try:
            errs = _compare_with_reference(args, enc, p64, cpu, out, ref, lowp, out_tol, grad_tol)
            break
        except AssertionError as e:     # (the next assignment of the ambiguous derivatives)
            failures.append(str(e))
    else:
        raise AssertionError('no assignment of the %d ambiguous relu derivatives %s meets the tolerance: %s'
                             % (len(ambiguous), ambiguous, failures))
    if ambiguous and not lowp:
        errs['relu_ambiguous'] = float(len(ambiguous))
    return errs, used_graph


RELU_AMBIGUOUS = 1e-6     # |z| below this is zero to fp32 resolution (z is a sum of 64 products of O(1) values)


def _compare_with_reference(args, enc, p64, cpu, out, ref, lowp, out_tol, grad_tol):
    errs = {'out': KC.assert_close('encoder output', out, ref, tol=out_tol)}
    m_rows = cpu['src'].shape[0] * cpu['src'].shape[1]
    for name, p in enc.named_parameters():
        g_ref = p64[name].grad
        if p.grad is None:
            assert g_ref is None or float(g_ref.abs().max()) == 0.0, name
            continue
        if lowp:
            if float(g_ref.norm()) < 1e-9:       # exactly-zero gradient (linear2.bias in front of BatchNorm)
                continue
            errs[name] = rel_fro(p.grad, g_ref)
            continue
        if not args.layer_norm and name.endswith('linear2.bias'):
            # linear2.bias sits directly in front of BatchNorm 2 (out_proj.bias does not: it is scaled by the degree
            # first): its gradient is EXACTLY zero (the batch mean is subtracted right
            # after it), computed as a sum of M = N*B O(1) terms that cancel - what any fp32 implementation
            # returns is rounding noise of size ~ eps * sqrt(M) * |terms|, so only that bound is checked
            assert float(g_ref.abs().max()) < 1e-10, name
            noise = float(p.grad.detach().abs().max())
            assert noise <= 64 * 6e-8 * m_rows ** 0.5, '%s: %.3e' % (name, noise)
            errs[name] = noise
            continue
        errs[name] = KC.assert_close('grad ' + name, p.grad.detach().cpu(), g_ref, tol=grad_tol)
    if lowp:
        bad = {k: '%.3e' % v for k, v in errs.items() if k != 'out' and v > bf16_grad_tol(k)}
        assert not bad, 'relative Frobenius error of parameter gradients above %.2g (%.2g behind the relu mask): %s (all: %s)' % (
            BF16_GRAD_FRO_TOL, BF16_GRAD_FRO_TOL_RELU, bad, {k: '%.2e' % v for k, v in errs.items()})
    return errs
