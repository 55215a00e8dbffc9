"""rows = H*B = 4096 (config 5): the C x C linear as library bf16 GEMMs with fp32 output (operands cast by framework ops)
against the tiled own kernels (bf16 compute, fp32 operands rounded when staged)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from feta_tmlr_amd import _lib
abi, st = _lib.abi(), _lib.stream_handle()
dev = torch.device('cuda:0')
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
c = 1024
for rows in (512, 2048, 4096):
    x, dy, w, b = torch.randn(rows, c, device=dev), torch.randn(rows, c, device=dev), torch.randn(c, c, device=dev) / 32, torch.randn(c, device=dev)
    y, dx, dw, db = (torch.empty(rows, c, device=dev), torch.empty(rows, c, device=dev), torch.empty(c, c, device=dev), torch.empty(c, device=dev))
    t_own_f = timeit(lambda: abi.lin_fwd(x, w, b, y, st, bf16=True))
    t_own_b = timeit(lambda: abi.lin_bwd(x, w, dy, dx, dw, db, st, bf16=True))
    try:
        def lib_f():
            return torch.mm(x.bfloat16(), w.bfloat16().t(), out_dtype=torch.float32) + b
        def lib_b():
            d16 = dy.bfloat16()
            return torch.mm(d16, w.bfloat16(), out_dtype=torch.float32), torch.mm(d16.t(), x.bfloat16(), out_dtype=torch.float32)
        t_lib_f, t_lib_b = timeit(lib_f), timeit(lib_b)
        ref = x.double() @ w.double().t() + b.double()
        e_lib = float((lib_f().double() - ref).abs().max() / ref.abs().max())
        abi.lin_fwd(x, w, b, y, st, bf16=True); torch.cuda.synchronize()
        e_own = float((y.double() - ref).abs().max() / ref.abs().max())
    except Exception as ex:
        t_lib_f = t_lib_b = float('nan'); e_lib = e_own = str(ex)[:80]
    print('rows %5d  own bf16 fwd %7.1f us bwd %7.1f us | library bf16 (casts + mm fp32-out) fwd %7.1f us bwd %7.1f us | rel err own %s lib %s'
          % (rows, t_own_f, t_own_b, t_lib_f, t_lib_b, e_own, e_lib))
