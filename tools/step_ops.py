"""Which PyTorch ops (not hand-written kernels) still run inside one fwd+bwd step of the bench
configuration: torch.profiler over a few eager steps, ops with device time, per step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402

sys.argv = ['bench.py'] + sys.argv[1:]   # bench.py flags select the configuration (--shape, --layer-norm, ...)
args = bench.parse()
dev = torch.device('cuda:0')
cpu, gpu = bench.make_batch(args, 0, dev)
enc = bench.build_encoder(args).to(dev)
enc.train()
fwd_args = (gpu['src'], gpu['pe'], gpu['edge_index'], gpu['fi'], gpu['batch'])
fwd_kw = dict(degree=gpu['degree'], src_key_padding_mask=gpu['mask'], graph_cache=gpu['cache'])


def step():
    for p in enc.parameters():
        p.grad = None
    out, _, _ = enc(*fwd_args, **fwd_kw)
    out.backward(gradient=gpu['dout'])


for _ in range(3):
    step()
torch.cuda.synchronize()
STEPS = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(e, 'self_cuda_time_total', 0)
    if dt > 0:
        rows.append((dt / STEPS, e.count / STEPS, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
for dt, cnt, key, shp in rows:
    print('%8.1f us/step  x%5.1f  %-40s %s' % (dt, cnt, key[:40], shp))
