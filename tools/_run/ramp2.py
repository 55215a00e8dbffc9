import sys, time, torch
sys.path.insert(0, '.')
import bench
from feta_tmlr_amd import _lib
args = bench.parse(['--no-cpu-baseline', '--no-literal'])
torch.cuda.set_device(0); dev = torch.device('cuda', 0); _lib.abi()
cpu, gpu = bench.make_batch(args, 0, dev)
enc = bench.build_encoder(args).to(dev); enc.train()
step, _, _ = bench.make_step(args, enc, gpu, 1, dev)
args.steps, args.warmup = 20, 5
for rep in range(12):
    dt = bench.time_steps(step, args, 1, dev)
    print('time_steps(20) #%d: %.4f ms/step' % (rep, dt / 20 * 1e3), flush=True)
