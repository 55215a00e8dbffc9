"""Tunes the library GEMMs of the coefficient generator's C x C linear (transformer/models.py:284; feta_tmlr_amd/
functional.py: torch.addmm / mm - the fp32 path keeps the library at C = 1024) with PyTorch's
TunableOp over the row counts H*B of the BASELINE shapes and writes the selected rocBLAS / hipBLASLt solutions to
feta_tmlr_amd/gemm_tuning_gfx950.csv, which feta_tmlr_amd/_lib.py hands to TunableOp (tuning disabled) when the library
is loaded.  Run on the GPU box:  python tools/tune_gemm.py [out.csv]"""
import os
import sys

import torch

out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/gemm_tuning_gfx950.csv'
os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
if os.path.exists(out):
    os.remove(out)
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_filename(out)
dev = torch.device('cuda:0')
c = 1024
w, bias = torch.randn(c, c, device=dev) / 32, torch.randn(c, device=dev)
for rows in (128, 256, 512, 1024, 2048, 4096):
    x, dy = torch.randn(rows, c, device=dev), torch.randn(rows, c, device=dev)
    for _ in range(2):
        y = torch.addmm(bias, x, w.t())      # forward
        dx = dy.mm(w)                         # dX
        dw = dy.t().mm(x)                     # dW
    torch.cuda.synchronize()
    print('rows', rows, 'tuned')
torch.cuda.tunable.write_file(out) if hasattr(torch.cuda.tunable, 'write_file') else None
print(open(out).read())
