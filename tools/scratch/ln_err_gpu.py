import sys, os, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from feta_tmlr_amd import fused_stack
dev = torch.device('cuda:0')
argv = ['--shape', 'pattern', '--k-eig', '32', '--layer-norm', '--no-pe', '--batch', '64', '--n-pad', '120', '--no-graph']
args = bench.parse(argv)
cpu, gpu = bench.make_batch(args, 0, dev)
res = {}
for flag in (True, False):
    fused_stack.USE_LN_ON_LOAD = flag
    enc = bench.build_encoder(args).to(dev)
    enc.train()
    step, _, _ = bench.make_step(args, enc, gpu, 1, dev)
    step()
    torch.cuda.synchronize()
    res[flag] = {k: p.grad.detach().double().cpu().clone() for k, p in enc.named_parameters() if p.grad is not None}
    res[flag]['out'] = step.held['out'].double().cpu().clone()
for k in res[True]:
    a, b = res[True][k], res[False][k]
    d = (a - b).abs()
    print('%-40s max|ref| %9.3e  max diff %9.3e  rel %8.2e' % (k, float(b.abs().max()), float(d.max()), float(d.max() / b.abs().max().clamp(min=1e-30))))

for k in ('layers.2.linear1.weight', 'layers.1.linear1.weight'):
    d = (res[True][k] - res[False][k]).abs()
    print(k, 'diff by hidden unit (row max):', ' '.join('%.0e' % float(x) for x in d.max(1).values))
d = (res[True]['layers.2.linear1.bias'] - res[False]['layers.2.linear1.bias'])
print('db1 diff:', ' '.join('%.0e' % float(x) for x in d))
print('db1 ref :', ' '.join('%.0e' % float(x) for x in res[False]['layers.2.linear1.bias']))
d = (res[True]['layers.2.norm1.weight'] - res[False]['layers.2.norm1.weight'])
print('dgamma1 l2 diff:', ' '.join('%.0e' % float(x) for x in d))
