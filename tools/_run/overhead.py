"""Where does the fixed ~0.2 ms of a timed region go?  T(K) for small K, synchronize vs event spin."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from feta_tmlr_amd import _lib
args = bench.parse(['--no-cpu-baseline', '--no-literal'])
torch.cuda.set_device(0); dev = torch.device('cuda', 0); _lib.abi()
cpu, gpu = bench.make_batch(args, 0, dev)
enc = bench.build_encoder(args).to(dev); enc.train()
step, _, _ = bench.make_step(args, enc, gpu, 1, dev)
for _ in range(50): step()
torch.cuda.synchronize()
def t_sync(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def t_spin(K):
    ev = torch.cuda.Event()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    ev.record()
    while not ev.query(): pass
    t1 = time.perf_counter(); torch.cuda.synchronize(); return (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3
def t_launch(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); return (t1 - t0) * 1e3
def t_events(K):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(K): step()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)
for K in (1, 2, 5, 10, 20, 50, 200):
    s = sorted(t_sync(K) for _ in range(7))[3]
    sp = sorted(t_spin(K) for _ in range(7))[3]
    l = sorted(t_launch(K) for _ in range(7))[3]
    e = sorted(t_events(K) for _ in range(7))[3]
    print('K=%3d sync %.3f ms (%.4f/step)  spin %.3f / %.3f  host-launch-only %.3f  events %.3f (%.4f/step)' % (K, s, s / K, sp[0], sp[1], l, e, e / K), flush=True)
t0 = time.perf_counter()
for _ in range(100): torch.cuda.synchronize()
print('idle synchronize: %.1f us' % ((time.perf_counter() - t0) * 1e4))
