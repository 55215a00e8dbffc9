"""Data parallelism for the FeTA block: one process per GPU, graphs sharded by index, parameters
replicated, gradients summed with ONE collective over a flat bucket.

The reference's only multi-GPU mechanism is single-process nn.DataParallel in three scripts
(experiments/run_transformer_gengcn_molpcba.py:446-452); this is a new design for
8 MI355X on xGMI: the whole model is ~2.2 M parameters (8.8 MB fp32, dominated by encoder.gcn and
encoder.linear), so the all-reduce is latency-bound and a single bucket is the right size.
Parameters whose gradient is never produced (the reference's unused outer GCNConv,
transformer/models.py:508) simply stay zero in the bucket on every rank.
"""
import torch
import torch.distributed as dist


def shard_indices(num_items, rank, world_size):
    """Round-robin shard: rank r takes items i with i % world_size == r."""
    return list(range(rank, num_items, world_size))


class FlatGradAllReduce:
    """Keeps every parameter's .grad as a view into one flat buffer.

    zero()        memset of the bucket (replaces zero_grad; autograd then accumulates in place,
                  so the addresses are stable across steps and hipGraph replays)
    all_reduce()  one all-reduce (RCCL ``nccl`` backend on GPUs, ``gloo`` in the CPU tests) of the
                  bucket, then 1/world scaling -> averaged gradients, identical on all ranks
    """

    def __init__(self, params, world_size=None, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        self.group = process_group
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.world_size = world_size
        off = 0
        for p in self.params:
            assert p.device == dev and p.dtype == dt
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, async_op=False):
        if self.world_size == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if async_op:
            return work
        work.wait()
        self.flat.mul_(1.0 / self.world_size)
        return None

    def finish(self, work):
        work.wait()
        self.flat.mul_(1.0 / self.world_size)
