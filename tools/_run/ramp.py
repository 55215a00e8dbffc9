"""Per-replay durations of the first replays after capture (is there a ramp, and how long?)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from feta_tmlr_amd import _lib
args = bench.parse(['--no-cpu-baseline', '--no-literal'])
torch.cuda.set_device(0); dev = torch.device('cuda', 0); _lib.abi()
cpu, gpu = bench.make_batch(args, 0, dev)
enc = bench.build_encoder(args).to(dev); enc.train()
step, _, _ = bench.make_step(args, enc, gpu, 1, dev)
torch.cuda.synchronize()
N = 600
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(N):
    step(); ev[i + 1].record()
torch.cuda.synchronize()
print('wall %.3f ms' % ((time.perf_counter() - t0) * 1e3))
d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
for a in range(0, N, 20):
    print('%3d-%3d: mean %.1f us  min %.1f max %.1f' % (a, a + 19, sum(d[a:a + 20]) / 20, min(d[a:a + 20]), max(d[a:a + 20])))
# then: idle 0.5 s and again 60
time.sleep(0.5)
ev[0].record()
for i in range(60):
    step(); ev[i + 1].record()
torch.cuda.synchronize()
d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(60)]
for a in range(0, 60, 20):
    print('after 0.5 s idle %3d-%3d: mean %.1f us' % (a, a + 19, sum(d[a:a + 20]) / 20))
# the bench's own protocol, repeated
for rep in range(5):
    args.steps, args.warmup = 20, 5
    dt = bench.time_steps(step, args, 1, dev)
    print('time_steps(20): %.4f ms/step' % (dt / 20 * 1e3))
