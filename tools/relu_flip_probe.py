"""Why the bf16 storage leg's gradients of linear1.* and norm1.* sit at 4-11 % relative Frobenius error against the
fp64 oracle while every other parameter is at 0.3-3 % (tests/bench_checks.py): one FFN half with its two BatchNorms at
the BASELINE row count (4736 x 64, dim_feedforward 128), float64 arithmetic, with each bf16 rounding switched on
separately.  Result (this script, CPU): rounding x / W1 to bf16 flips the relu mask of ~400 of 606 k (row, unit) pairs
and THAT moves dbeta1 / dgamma1 / db1 / dW1 by 6 / 4 / 6 / 4 %; with the exact mask all roundings together give 0.3-0.5 %.
The gradient of a non-smooth function evaluated at bf16-perturbed pre-activations: any bf16 implementation has it, fp32
column sums (which the kernels do keep) cannot remove it."""
import torch
torch.manual_seed(0)
M,d,ff=4736,64,128
bf=lambda t:t.to(torch.bfloat16).double()
y1=torch.randn(M,d,dtype=torch.float64)
W1=torch.randn(ff,d,dtype=torch.float64)/8; b1=torch.randn(ff,dtype=torch.float64)*0.1
W2=torch.randn(d,ff,dtype=torch.float64)/11; b2=torch.randn(d,dtype=torch.float64)*0.1
g=torch.randn(M,d,dtype=torch.float64)   # gradient wrt BN2 output
def run(round_x=False, round_w=False, round_h=False, round_g2=False, mask_from=None):
    x=(y1-y1.mean(0))/y1.std(0,unbiased=False)          # BN1 output
    xr=bf(x) if round_x else x
    w1=bf(W1) if round_w else W1; w2=bf(W2) if round_w else W2
    pre=xr@w1.t()+b1
    h=torch.relu(pre)
    hs=bf(h) if round_h else h
    y2=x+hs@w2.t()+b2
    mean,var=y2.mean(0),y2.var(0,unbiased=False); rstd=(var+1e-5).rsqrt(); xh=(y2-mean)*rstd
    g2=rstd*(g-g.mean(0)-xh*(g*xh).mean(0))
    g2r=bf(g2) if round_g2 else g2
    mask=(h>0) if mask_from is None else mask_from
    dh=(g2r@w2)*mask
    dhr=bf(dh) if round_h else dh
    dx1=g2+dhr@w1
    return dict(dbeta1=dx1.sum(0), dgamma1=(dx1*x).sum(0), db1=dh.sum(0), dW1=dhr.t()@xr, mask=mask)
ref=run()
rel=lambda a,b:float((a-b).norm()/b.norm())
for name,kw in [('all rounded',dict(round_x=True,round_w=True,round_h=True,round_g2=True)),
                ('only x,w rounded (mask flips + products)',dict(round_x=True,round_w=True)),
                ('all rounded, exact mask',dict(round_x=True,round_w=True,round_h=True,round_g2=True,mask_from=ref['mask'])),
                ('only g2 rounded',dict(round_g2=True)),('only h/dh rounded',dict(round_h=True))]:
    r=run(**kw)
    print('%-45s'%name,' '.join('%s %.3f'%(k,rel(r[k],ref[k])) for k in ('dbeta1','dgamma1','db1','dW1')), 'flips',int((r['mask']!=ref['mask']).sum()))
