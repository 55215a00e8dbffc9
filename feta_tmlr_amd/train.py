"""One optimisation step of the FeTA models, per task family, as the reference's training scripts
do it - the direct caller of the hot path.

    task      loss                                     optimiser                    reference
    'zinc'    L1                                       Adam(lr), warm-up schedule   experiments/run_transformer_gengcn.py:115-164,301-317
    'tu'      CrossEntropy (BCE-with-logits if 1 out)  AdamW(lr, wd=1e-4), StepLR   experiments/run_transformer_gengcn_cv.py:123-193,356-362
    'molhiv'  BCE-with-logits on non-NaN labels        AdamW                        experiments/run_transformer_gengcn_molhiv.py:136-222
    'sbm'     CrossEntropy over real nodes             AdamW                        experiments/run_transformer_gengcn_SBM_cv.py:145-218,367-373

What is not reproduced: the per-step NaN / |param|>1000 scans that drop into pdb
(run_transformer_gengcn_cv.py:161-179), the host timers and prints, the `.cuda()` copies (the
collate already emits device tensors), dataset download and CSV logging.

``GraphedTrainStep`` is the MI355X form of the loop body: forward, loss, backward and the optimiser
update captured once per padded size into a hipGraph and replayed - the reference pays ~10^3 kernel
launches and a host sync (`.cpu()` at transformer/models.py:246) per step.
"""
import torch
from torch import nn
import torch.nn.functional as F


def warmup_lr(step, lr, warmup):
    """experiments/run_transformer_gengcn.py:310-317."""
    if step < warmup:
        return 1e-6 + step * (lr - 1e-6) / warmup
    return lr * warmup ** 0.5 * step ** -0.5


def lap_sign_flip(lap_pe, generator=None):
    """Random sign per Laplacian-PE column, one draw per batch
    (experiments/run_transformer_gengcn.py:126-131)."""
    flip = torch.rand(lap_pe.shape[-1], generator=generator)
    flip = torch.where(flip >= 0.5, torch.ones_like(flip), -torch.ones_like(flip))
    return lap_pe * flip.to(lap_pe.device).unsqueeze(0)


def make_criterion(task, nb_class=1):
    if task == 'zinc':
        return nn.L1Loss()
    if task in ('tu', 'sbm'):
        return nn.BCEWithLogitsLoss() if nb_class == 1 else nn.CrossEntropyLoss()
    if task == 'molhiv':
        return nn.BCEWithLogitsLoss()
    raise ValueError('unknown task %r' % (task,))


def make_optimizer(task, params, lr, weight_decay=1e-4, capturable=False, fused=None):
    """Adam for ZINC (run_transformer_gengcn.py:302), AdamW elsewhere (..._cv.py:360).
    fused (default: with capturable, on the GPU): PyTorch's single-kernel multi-tensor update instead of
    its ~10 foreach kernels per step - the same arithmetic per element."""
    params = [p for p in params if p.requires_grad]
    if fused is None:
        fused = bool(capturable) and all(p.is_cuda for p in params)
    kw = dict(lr=lr, capturable=capturable)
    if fused:
        kw['fused'] = True
    if task == 'zinc':
        return torch.optim.Adam(params, **kw)
    return torch.optim.AdamW(params, weight_decay=weight_decay, **kw)


def pad_node_labels(labels, feature_indices, bsz, n_pad):
    """[N_tot] node labels (graph-major, transformer/data.py:456) -> [B, N_pad] with -100 on the
    padded positions (the ignore_index of cross_entropy)."""
    out = torch.full((bsz, n_pad), -100, dtype=torch.int64, device=labels.device)
    out[feature_indices[:, 0], feature_indices[:, 1]] = labels
    return out


def task_loss(task, model, criterion, batch9, graph_cache=None, padded_node_labels=False):
    """forward + loss of one collated batch; returns (loss, model output).
    padded_node_labels ('sbm'): labels are [B, N_pad] from pad_node_labels and the model returns the
    logits of every position - the same mean over the real nodes as the reference's boolean gather
    (transformer/models.py:1070-1071) with static shapes and no host sync."""
    x, mask, pe, lap_pe, degree, labels, edge_index, batch, feature_indices = batch9
    extra = {'padded_logits': True} if padded_node_labels else {}
    res = model(x, edge_index, batch, feature_indices, mask, pe, lap_pe, degree,
                graph_cache=graph_cache, **extra)
    output = res[0]
    if padded_node_labels:
        if not isinstance(criterion, nn.CrossEntropyLoss):
            raise ValueError('padded node labels need a CrossEntropyLoss criterion')
        return F.cross_entropy(output.reshape(-1, output.shape[-1]), labels.reshape(-1),
                               weight=criterion.weight, ignore_index=-100), output
    if task == 'zinc':
        labels = labels.view(output.shape)          # default_collate of g.y [1] gives [B, 1]
        return criterion(output, labels), output
    if task == 'molhiv':
        labels = labels.view(-1)
        # BCE over the labeled graphs (run_transformer_gengcn_molhiv.py:177-178) without the
        # boolean gather (a host sync): NaN labels get weight 0, the mean runs over the others
        keep = ~torch.isnan(labels)
        tgt = torch.where(keep, labels, torch.zeros_like(labels)).to(output.dtype)
        w = keep.to(output.dtype)
        per = F.binary_cross_entropy_with_logits(output.view(-1), tgt, reduction='none')
        return (per * w).sum() / w.sum(), output
    labels = labels.view(-1)
    if isinstance(criterion, nn.BCEWithLogitsLoss):
        return criterion(output.view(-1), labels.to(output.dtype)), output
    return criterion(output, labels), output


def train_step(task, model, criterion, optimizer, batch9, graph_cache=None, lr=None):
    """One iteration of the reference's train_epoch loop body.  -> loss (0-d tensor, no sync)."""
    if lr is not None:
        for group in optimizer.param_groups:
            group['lr'] = lr
    optimizer.zero_grad(set_to_none=True)
    loss, _ = task_loss(task, model, criterion, batch9, graph_cache)
    loss.backward()
    optimizer.step()
    from .functional import DropoutState
    if DropoutState.device_mode():
        DropoutState.end_step()      # (an eager step between graph replays: the shared device offset moves past its masks)
    return loss.detach()


def accuracy_SBM(scores, targets, nb_class=None):
    """experiments/run_transformer_gengcn_SBM_cv.py:126-143: mean per-class recall in percent.
    Classes are 0..max(label, prediction) as sklearn's confusion_matrix would order them (or
    nb_class when given); computed on the device, one host read at the end."""
    pred = scores.argmax(dim=1)
    if nb_class is None:
        nb_class = int(max(int(targets.max()), int(pred.max())) + 1)
    size = torch.bincount(targets, minlength=nb_class).to(torch.float64)
    hit = torch.bincount(targets[pred == targets], minlength=nb_class).to(torch.float64)
    recall = torch.where(size > 0, hit / size.clamp(min=1), torch.zeros_like(size))
    return float(100.0 * recall.sum() / nb_class)


def prepare_cache(model, batch9, graph_cache=None):
    """Graph structure in kernel layout, built outside any hipGraph: node counts / offsets from the
    mask and, for filter_mode='cheb', the dense scaled Laplacian from the edge list (the edge list
    has a different length in every batch, so it cannot be an input of a captured graph)."""
    x, mask, edge_index, batch = batch9[0], batch9[1], batch9[6], batch9[7]
    return model.encoder._graph_cache(graph_cache, edge_index, batch, mask, x.shape[1])


class GraphedTrainStep:
    """forward + loss + backward + optimiser update of one padded batch shape, captured into ONE
    hipGraph.  The fixed-shape batch tensors are copied into static buffers, so every batch
    replayed through one instance must have the [B, N_pad] of the example (one instance per bucket
    of ``data.bucket_batches``); the variable-length members of the tuple (edge_index, batch,
    feature_indices) are consumed before the graph by ``prepare_cache`` / ``pad_node_labels``.
    The learning rate lives in a device scalar (capturable optimiser), so the warm-up schedule does
    not re-capture.

    Construction does not advance training: the warm-up iterations and the capture run REAL steps on
    the example batch, so the parameters, the module buffers (BatchNorm running statistics,
    num_batches_tracked) and the optimiser state (moments and step counters) are snapshotted before
    and restored, in place, afterwards - an instance created mid-training (one per padded-size bucket)
    leaves the trajectory exactly where the reference loop would have it."""

    def __init__(self, task, model, criterion, optimizer, batch9, graph_cache=None, warmup_iters=3):
        self.task, self.model, self.criterion, self.optimizer = task, model, criterion, optimizer
        for group in optimizer.param_groups:
            if not group.get('capturable', False):
                raise ValueError('GraphedTrainStep needs make_optimizer(..., capturable=True)')
            if not torch.is_tensor(group['lr']):
                group['lr'] = torch.tensor(float(group['lr']), device=batch9[0].device)
        clone = lambda t: None if t is None else t.clone()
        graph_cache = prepare_cache(model, batch9, graph_cache)
        self.static = tuple(clone(t) for t in self._fixed(batch9))
        self.cache = type(graph_cache)(clone(graph_cache.n_real), clone(graph_cache.node_off),
                                       graph_cache.n_pad, clone(graph_cache.u),
                                       clone(graph_cache.lam), clone(graph_cache.lhat),
                                       {k: (v.clone() if torch.is_tensor(v) else v)
                                        for k, v in graph_cache.extra.items()})
        # attention-probability dropout inside the graph: the (seed, offset) key moves to the device, every replay
        # reads it there and advances it (functional.DropoutState) - the masks of replay i are those of eager step i
        from .functional import DropoutState
        DropoutState.begin_device_mode(batch9[0].device)
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()
        self._restore(snap)

    def _snapshot(self):
        """Copies of everything a step mutates; optimiser state that does not exist yet (lazily created
        by the first step) is recorded as absent and zeroed on restore."""
        tensors = list(self.model.parameters()) + list(self.model.buffers())
        opt = {}
        for p, st in self.optimizer.state.items():
            opt[p] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
        # ... and the random streams the warm-up steps draw from: torch's generators (nn.Dropout) and the
        # (seed, offset) of the attention-probability dropout masks
        from .functional import DropoutState
        dev = tensors[0].device if tensors else None
        rng = (torch.get_rng_state(), torch.cuda.get_rng_state(dev) if (dev is not None and dev.type == 'cuda') else None,
               DropoutState.snapshot())
        return [t.detach().clone() for t in tensors], opt, rng

    @torch.no_grad()
    def _restore(self, snap):
        """In place: the captured graph holds the addresses of these tensors."""
        saved, opt, rng = snap
        tensors = list(self.model.parameters()) + list(self.model.buffers())
        for t, s in zip(tensors, saved):
            t.copy_(s)
        from .functional import DropoutState
        torch.set_rng_state(rng[0])
        if rng[1] is not None:
            torch.cuda.set_rng_state(rng[1], tensors[0].device)
        DropoutState.restore(rng[2])
        for p, st in self.optimizer.state.items():
            before = opt.get(p)
            for k, v in st.items():
                if torch.is_tensor(v):
                    if before is not None and k in before:
                        v.copy_(before[k])
                    else:
                        v.zero_()
                elif before is not None and k in before:
                    st[k] = before[k]

    def _fixed(self, batch9):
        x, mask, pe, lap_pe, degree, labels, edge_index, batch, feature_indices = batch9
        if self.task == 'sbm':
            labels = pad_node_labels(labels, feature_indices, x.shape[0], x.shape[1])
        return (x, mask, pe, lap_pe, degree, labels, None, None, None)

    def _body(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss, _ = task_loss(self.task, self.model, self.criterion, self.static, self.cache,
                            padded_node_labels=self.task == 'sbm')
        loss.backward()
        self.optimizer.step()
        from .functional import DropoutState
        self.drop_calls = DropoutState.end_step()     # (in-graph: the device offset moves past this step's masks)
        return loss.detach()

    def set_lr(self, lr):
        for group in self.optimizer.param_groups:
            group['lr'].fill_(lr)

    def __call__(self, batch9, graph_cache=None):
        graph_cache = prepare_cache(self.model, batch9, graph_cache)
        for dst, src in zip(self.static, self._fixed(batch9)):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        for name in ('n_real', 'node_off', 'u', 'lam', 'lhat'):
            dst, src = getattr(self.cache, name), getattr(graph_cache, name)
            if dst is not None and src is not None:
                dst.copy_(src, non_blocking=True)
        for key, dst in self.cache.extra.items():   # e.g. the per-row degree scale of this batch
            if torch.is_tensor(dst):
                dst.copy_(graph_cache.extra[key], non_blocking=True)
        self.graph.replay()
        if self.drop_calls:
            from .functional import DropoutState
            DropoutState.replayed(self.drop_calls)
        return self.loss


def rocauc(scores, labels):
    """Binary ROC-AUC over the labeled graphs (what ogb's Evaluator('ogbg-molhiv') reports,
    experiments/run_transformer_gengcn_molhiv.py:213-218): rank statistic with ties averaged."""
    keep = ~torch.isnan(labels)
    s, y = scores[keep].double().cpu(), labels[keep].double().cpu()
    pos, neg = int((y > 0.5).sum()), int((y <= 0.5).sum())
    if pos == 0 or neg == 0:
        return float('nan')
    order = torch.argsort(s)
    ranks = torch.empty_like(s)
    ranks[order] = torch.arange(1, len(s) + 1, dtype=torch.float64)
    vals, inv, counts = torch.unique(s, return_inverse=True, return_counts=True)
    sums = torch.zeros_like(vals).scatter_add_(0, inv, ranks)
    ranks = (sums / counts)[inv]                        # average rank of tied scores
    return float((ranks[y > 0.5].sum() - pos * (pos + 1) / 2.0) / (pos * neg))


@torch.no_grad()
def evaluate(task, model, criterion, batches):
    """The reference's eval_epoch for one split: ``batches`` yields (batch9, graph_cache).  Returns a dict:
    'loss' (sample-weighted mean, as running_loss / n_sample) plus 'mae' and 'mse' (zinc,
    run_transformer_gengcn.py:167-209), 'acc' (tu: fraction of graphs; sbm: mean per-class recall in
    percent averaged over batches, ..._SBM_cv.py:221-266) or 'rocauc' (molhiv).  The model is put in eval
    mode (BatchNorm running statistics) and restored."""
    was_training = model.training
    model.eval()
    tot = {'loss': 0.0, 'mae': 0.0, 'mse': 0.0, 'hit': 0.0, 'acc_sum': 0.0}
    n_sample, n_batch = 0, 0
    scores, labels_all = [], []
    for batch9, cache in batches:
        loss, out = task_loss(task, model, criterion, batch9, cache)
        labels = batch9[5]
        bsz = batch9[0].shape[0]
        tot['loss'] += float(loss) * bsz
        n_sample += bsz
        n_batch += 1
        if task == 'zinc':
            lab = labels.view(out.shape)
            tot['mae'] += float(F.l1_loss(out, lab)) * bsz
            tot['mse'] += float(F.mse_loss(out, lab)) * bsz
        elif task == 'tu':
            pred = out.argmax(dim=1) if out.dim() == 2 and out.shape[1] > 1 else (out.view(-1) > 0).long()
            tot['hit'] += float((pred == labels.view(-1)).sum())
        elif task == 'sbm':
            tot['acc_sum'] += accuracy_SBM(out, labels.view(-1))
        elif task == 'molhiv':
            scores.append(out.view(-1).detach())
            labels_all.append(labels.view(-1))
    model.train(was_training)
    res = {'loss': tot['loss'] / max(n_sample, 1)}
    if task == 'zinc':
        res.update(mae=tot['mae'] / n_sample, mse=tot['mse'] / n_sample)
    elif task == 'tu':
        res['acc'] = tot['hit'] / n_sample
    elif task == 'sbm':
        res['acc'] = tot['acc_sum'] / max(n_batch, 1)
    elif task == 'molhiv':
        res['rocauc'] = rocauc(torch.cat(scores), torch.cat(labels_all))
    return res
