import sys, torch
sys.path.insert(0, '.')
from feta_tmlr_amd import _lib
dev = torch.device('cuda:0')
abi, st = _lib.abi(), _lib.stream_handle()
b, h, n, c, dh, d = 128, 4, 37, 1024, 16, 64
m = b * n
G = abi.coeff_bwd_groups(b, h)
RC = abi.rowlin_chunks(m)
print('G', G, 'RC', RC)
rnd = lambda *s: torch.randn(*s, device=dev)
part = rnd(G, 2 * c); ds, db, dw = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(c, c, device=dev)
dbp, dbias = rnd(b * h, dh), torch.empty(dh, device=dev)
dcoeff, dbl = rnd(b * h, c), torch.empty(c, device=dev)
cat, dwdb = rnd(RC, d * 2 * d + d), torch.empty(d * 2 * d + d, device=dev)
segs = {'ds+bcast': (part[:, :c], ds, dw), 'db': (part[:, c:], db), 'dbias': (dbp, dbias), 'db_lin': (dcoeff, dbl), 'cat': (cat, dwdb)}
def t(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for k, v in segs.items():
    print('%-10s alone %.2f us   with cat (mixed kernel) %.2f us' % (k, t(lambda: abi.colsum_multi([v], st)), t(lambda: abi.colsum_multi([v, segs['cat']], st))))
print('all 5: %.2f us' % t(lambda: abi.colsum_multi(list(segs.values()), st)))
print('old 3 launches: %.2f us' % t(lambda: (abi.colsum_multi([segs['cat']], st), abi.colsum_multi([segs['dbias'], segs['db_lin']], st), abi.colsum_multi([segs['ds+bcast'], segs['db']], st))))
