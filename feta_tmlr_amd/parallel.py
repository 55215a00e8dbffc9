"""Data parallelism for the FeTA block: one process per GPU, graphs sharded by index, parameters
replicated, gradients summed with ONE collective over a flat bucket.

The reference's only multi-GPU mechanism is single-process nn.DataParallel in three scripts
(experiments/run_transformer_gengcn_molpcba.py:446-452); this is a new design for
8 MI355X on xGMI: the whole model is ~2.2 M parameters (8.8 MB fp32, dominated by encoder.gcn and
encoder.linear), so the all-reduce is latency-bound and a single bucket is the right size.
Parameters whose gradient is never produced (the reference's unused outer GCNConv,
transformer/models.py:508) contribute zeros on every rank.
"""
import torch
import torch.distributed as dist


def shard_indices(num_items, rank, world_size):
    """Round-robin shard: rank r takes items i with i % world_size == r."""
    return list(range(rank, num_items, world_size))


class FlatGradAllReduce:
    """One flat fp32 bucket for all gradients.

    views=False (default): autograd writes fresh .grad tensors (no accumulate kernels; inside a
        hipGraph their addresses are static); all_reduce() packs them with ONE cat kernel, runs ONE
        all-reduce (RCCL ``nccl`` backend on GPUs, ``gloo`` in the CPU tests), scales by 1/world and
        scatters back with one multi-tensor copy.
    views=True: every .grad is a view into the bucket and autograd accumulates in place; zero()
        memsets the bucket.  No pack/unpack, one add kernel per parameter in backward.
    """

    def __init__(self, params, world_size=None, process_group=None, views=False):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        self.group = process_group
        self.views = views
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.world_size = world_size
        self.slices = []
        off = 0
        for p in self.params:
            assert p.device == dev and p.dtype == dt
            self.slices.append(self.flat[off:off + p.numel()])
            off += p.numel()
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
        parts = [p.grad.reshape(-1) if p.grad is not None else torch.zeros_like(s)
                 for p, s in zip(self.params, self.slices)]
        torch.cat(parts, out=self.flat)

    @torch.no_grad()
    def unpack(self):
        dst = [p.grad for p in self.params if p.grad is not None]
        src = [s.view_as(p) for p, s in zip(self.params, self.slices) if p.grad is not None]
        torch._foreach_copy_(dst, src)

    def all_reduce(self):
        """Averaged gradients in every p.grad (and in .flat), identical on all ranks."""
        if self.world_size == 1:
            return
        if not self.views:
            self.pack()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / self.world_size)
        if not self.views:
            self.unpack()
