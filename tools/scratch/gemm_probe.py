"""probe: library GEMM time of the three products of the C x C linear (R=512, C=1024), default vs TunableOp,
sequential vs two streams"""
import os, sys, time, torch
dev = torch.device('cuda:0')
R, C = 512, 1024
x, w, b, dy = torch.randn(R, C, device=dev), torch.randn(C, C, device=dev), torch.randn(C, device=dev), torch.randn(R, C, device=dev)

def t(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

def graph_time(fn, it=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): fn()
    return t(g.replay, it) / 10

fwd = lambda: torch.addmm(b, x, w.t())
dx = lambda: dy.mm(w)
dw = lambda: dy.t().mm(x)
side = torch.cuda.Stream()
def both_seq():
    dy.mm(w); dy.t().mm(x)
def both_par():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        dy.t().mm(x)
    dy.mm(w)
    cur.wait_stream(side)
print('eager us: fwd %.2f dx %.2f dw %.2f' % (t(fwd), t(dx), t(dw)))
print('graph us: fwd %.2f dx %.2f dw %.2f seq(dx,dw) %.2f par(dx||dw) %.2f' % (graph_time(fwd), graph_time(dx), graph_time(dw), graph_time(both_seq), graph_time(both_par)))
if len(sys.argv) > 1:
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.set_max_tuning_duration(200)
    torch.cuda.tunable.set_filename('/tmp/tunable.csv')
    t0 = time.time()
    fwd(); dx(); dw(); torch.cuda.synchronize()
    print('tuning took %.1f s' % (time.time() - t0))
    print('tuned eager us: fwd %.2f dx %.2f dw %.2f' % (t(fwd), t(dx), t(dw)))
    print('tuned graph us: fwd %.2f dx %.2f dw %.2f seq %.2f par %.2f' % (graph_time(fwd), graph_time(dx), graph_time(dw), graph_time(both_seq), graph_time(both_par)))
    try:
        print(open('/tmp/tunable.csv').read()[:1500])
    except Exception as e:
        print('no csv', e)
