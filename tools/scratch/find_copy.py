"""Which op of the timed step issues the device-to-device copy (__amd_rocclr_copyBuffer)?  One eager step under
torch.profiler with Python stacks; prints every Memcpy DtoD event with the stack of its launching op."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

args = bench.parse(['--no-graph'] + sys.argv[1:])
dev = torch.device('cuda:0')
cpu, gpu = bench.make_batch(args, 0, dev)
enc = bench.build_encoder(args).to(dev)
enc.train()
step, _, _ = bench.make_step(args, enc, gpu, 1, dev)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
for e in prof.events():
    n = e.name.lower()
    if 'memcpy' in n or 'copybuffer' in n or n in ('aten::copy_', 'aten::clone', 'aten::contiguous'):
        print(e.name, [tuple(s) for s in (e.input_shapes or [])], 'cuda_time', getattr(e, 'device_time', None))
        for s in (e.stack or [])[:8]:
            print('     ', s)
