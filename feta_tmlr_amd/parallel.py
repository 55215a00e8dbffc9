"""Data parallelism for the FeTA block: one process per GPU, graphs sharded by index, parameters
replicated, gradients summed with ONE collective over a flat bucket.

The reference's only multi-GPU mechanism is single-process nn.DataParallel in three scripts
(experiments/run_transformer_gengcn_molpcba.py:446-452); this is a new design for
8 MI355X on xGMI: the whole model is ~2.2 M parameters, dominated by encoder.gcn.weight and
encoder.linear.weight (C x C = 1 M each).  The all-reduce is latency-bound, so one bucket is the
right size, and its volume is halved by a structural fact: encoder.gcn only ever sees an all-ones
input (transformer/models.py:280-282), so its weight gradient is the SAME row repeated C times
(d colsum(W) / dW) - only one row travels.
Parameters whose gradient is never produced (the reference's unused outer GCNConv,
transformer/models.py:508) contribute zeros on every rank.
"""
import torch
import torch.distributed as dist


def shard_indices(num_items, rank, world_size):
    """Round-robin shard: rank r takes items i with i % world_size == r."""
    return list(range(rank, num_items, world_size))


def mark_row_constant(param):
    """Declare that param.grad always consists of identical rows (DenseGCNParams.weight)."""
    param._feta_row_constant = True
    return param


class FlatGradAllReduce:
    """One flat fp32 bucket for all gradients.

    views=False (default): autograd writes fresh .grad tensors (no accumulate kernels; inside a
        hipGraph their addresses are static); all_reduce() packs them with ONE cat kernel, runs ONE
        all-reduce (RCCL ``nccl`` backend on GPUs, ``gloo`` in the CPU tests), scales by 1/world and
        scatters back with one multi-tensor copy.  Row-constant gradients (mark_row_constant) are
        packed as a single row.
    views=True: every .grad is a view into the bucket and autograd accumulates in place; zero()
        memsets the bucket.  No pack/unpack, one add kernel per parameter in backward.
    """

    def __init__(self, params, world_size=None, process_group=None, views=False, bucket_dtype=None):
        """bucket_dtype=torch.bfloat16 (BASELINE config 3): the collective moves a bf16 copy of the packed fp32
        gradients - half the bytes over xGMI - and the averaged result is widened back into the fp32 bucket before
        it is scattered to the (fp32) .grad tensors; master weights and optimizer never see bf16."""
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev, dt = self.params[0].device, self.params[0].dtype
        self.group = process_group
        self.views = views
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.world_size = world_size
        # RCCL averages inside the collective (ReduceOp.AVG): no scaling kernel after it; gloo (the
        # CPU tests) has no AVG, there the sum is scaled
        self.avg_in_collective = (dist.is_initialized() and world_size > 1
                                  and dist.get_backend(process_group) == 'nccl')
        self.rowconst = [bool(getattr(p, '_feta_row_constant', False)) and not views and p.dim() == 2
                         for p in self.params]
        sizes = [p.shape[1] if rc else p.numel() for p, rc in zip(self.params, self.rowconst)]
        self.numel = sum(sizes)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        self.wire = None
        if bucket_dtype is not None and bucket_dtype != dt:
            assert not views, 'a reduced-precision wire bucket needs the packed (non-view) mode'
            self.wire = torch.zeros(self.numel, device=dev, dtype=bucket_dtype)
        self.slices = []
        off = 0
        for p, sz in zip(self.params, sizes):
            assert p.device == dev and p.dtype == dt
            self.slices.append(self.flat[off:off + sz])
            off += sz
        if views:
            for p, s in zip(self.params, self.slices):
                p.grad = s.view_as(p)

    def zero(self):
        if self.views:
            self.flat.zero_()
        else:
            for p in self.params:
                p.grad = None

    @torch.no_grad()
    def pack(self):
        parts = []
        for p, s, rc in zip(self.params, self.slices, self.rowconst):
            if p.grad is None:
                parts.append(torch.zeros_like(s))
            elif rc:
                parts.append(p.grad[0])
            else:
                parts.append(p.grad.reshape(-1))
        torch.cat(parts, out=self.flat)

    @torch.no_grad()
    def unpack(self):
        dst, src = [], []
        for p, s, rc in zip(self.params, self.slices, self.rowconst):
            if p.grad is None:
                continue
            dst.append(p.grad)
            src.append(s.unsqueeze(0).expand_as(p.grad) if rc else s.view_as(p))
        torch._foreach_copy_(dst, src)

    def start(self):
        """Pack and launch the all-reduce asynchronously (it runs on the collective's own stream,
        under whatever the caller enqueues next); finish(work) completes it."""
        if self.world_size == 1:
            return None
        if not self.views:
            self.pack()
        if self.wire is not None:
            # pre-scaled by 1 / world in fp32, then rounded ONCE to bf16 and SUMMED by the collective (ReduceOp.AVG on
            # bf16 is not there on every backend, and a sum of pre-scaled addends cannot overflow where the mean fits).
            # The running sum travels in bf16: each of the world - 1 additions rounds to 8 bits, so the result is
            # within ~(1 + log2(world)) * 2^-9 relative of the fp32 bucket per element for like-signed addends
            # (tests/test_parallel_gloo.py checks 2^-7 * max|g| at world 2 and 4)
            self.flat.mul_(1.0 / self.world_size)
            self.wire.copy_(self.flat)
            return dist.all_reduce(self.wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(self.flat, op=self._op(), group=self.group, async_op=True)

    def _op(self):
        return dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM

    def finish(self, work):
        if work is None:
            return
        work.wait()
        if self.wire is not None:
            self.flat.copy_(self.wire)     # widened to fp32 before the scatter (already the mean)
        elif not self.avg_in_collective:
            self.flat.mul_(1.0 / self.world_size)
        if not self.views:
            self.unpack()

    def all_reduce(self):
        """Averaged gradients in every p.grad (and in .flat), identical on all ranks."""
        self.finish(self.start())


class HybridGradAllReduce:
    """Large gradients are all-reduced IN PLACE, one collective each (no pack / unpack copies of
    megabytes: ``encoder.linear.weight`` and the dense ``encoder.gcn.weight`` gradient are 4 MB each);
    everything else goes through one packed FlatGradAllReduce bucket.  Meant for gradients that are
    ready early in backward (the filter stage), whose collectives run under the rest of backward."""

    def __init__(self, params, world_size=None, process_group=None, big_numel=1 << 18):
        params = [p for p in params if p.requires_grad]
        self.big = [p for p in params if p.numel() >= big_numel]
        small = [p for p in params if p.numel() < big_numel]
        self.small = FlatGradAllReduce(small, world_size, process_group) if small else None
        self.group = process_group
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.world_size = world_size
        self.avg_in_collective = (dist.is_initialized() and world_size > 1
                                  and dist.get_backend(process_group) == 'nccl')

    def zero(self):
        for p in self.big:
            p.grad = None
        if self.small is not None:
            self.small.zero()

    def start(self):
        if self.world_size == 1:
            return None
        op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
        works = []
        for p in self.big:
            g = p.grad
            if g is None:
                continue
            full = None
            if g.dim() == 2 and g.stride(0) == 0:
                g = g[0]            # broadcast row (d colsum(W) / dW): the one row IS the whole gradient
            elif g.dim() == 2 and getattr(p, '_feta_row_constant', False) and g.shape[0] > 1:
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                full, g = g, g[0]   # dense gradient with identical rows: row 0 travels, finish() rewrites the rest
            elif not g.is_contiguous():
                g = p.grad = g.contiguous()
            works.append((dist.all_reduce(g, op=op, group=self.group, async_op=True), g, full))
        return works, (self.small.start() if self.small is not None else None)

    def finish(self, handle):
        if handle is None:
            return
        works, small = handle
        for w, g, full in works:
            w.wait()
            if not self.avg_in_collective:
                g.mul_(1.0 / self.world_size)
            if full is not None:
                full[1:].copy_(g.unsqueeze(0).expand(full.shape[0] - 1, -1))
        if self.small is not None:
            self.small.finish(small)

    def all_reduce(self):
        self.finish(self.start())


class FlatBufferAllReduce:
    """All-reduce of a gradient buffer that is ALREADY flat: the fused encoder stack writes every weight,
    bias and BatchNorm gradient of its layers into one tensor (fused_stack.py) and hands the parameters
    views of it, so the collective runs in place - no pack, no unpack.  ``getter()`` returns that tensor
    (its address is stable across hipGraph replays)."""

    def __init__(self, getter, world_size=None, process_group=None, params=None):
        """params (optional): the parameters whose gradients must live in the flat buffer; start() then
        verifies that every .grad is a view of it (gradient accumulation or zero_grad(set_to_none=False)
        make AccumulateGrad add into an OLDER buffer, which this collective would silently miss)."""
        self.getter = getter
        self.params = None if params is None else [p for p in params if p.requires_grad]
        self.group = process_group
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.world_size = world_size
        self.avg_in_collective = (dist.is_initialized() and world_size > 1
                                  and dist.get_backend(process_group) == 'nccl')

    def start(self):
        if self.world_size == 1:
            return None
        flat = self.getter()
        if flat is None:
            raise RuntimeError('no flat stack gradient: backward has not run through the fused stack')
        self.check_views(flat)
        op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
        return dist.all_reduce(flat, op=op, group=self.group, async_op=True), flat

    def check_views(self, flat):
        if not self.params:
            return
        lo = flat.data_ptr()
        hi = lo + flat.numel() * flat.element_size()
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            if not (lo <= g.data_ptr() and g.data_ptr() + g.numel() * g.element_size() <= hi):
                raise RuntimeError('a stack gradient does not live in the flat buffer of the last backward (was '
                                   '.grad kept across steps? use set_to_none=True / p.grad = None, or reduce it with '
                                   'FlatGradAllReduce)')

    def finish(self, handle):
        if handle is None:
            return
        work, flat = handle
        work.wait()
        if not self.avg_in_collective:
            flat.mul_(1.0 / self.world_size)

    def all_reduce(self):
        self.finish(self.start())
