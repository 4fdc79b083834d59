import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops
def med(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))
g=torch.Generator(device='cuda').manual_seed(0)
for k,ci,co in [(3,16,12),(3,12,12),(3,12,8),(3,8,8),(3,8,4),(3,4,1),(5,16,16),(5,20,16)]:
    N,H=8,1024
    x=torch.randn(N,H,H,ci,device='cuda',generator=g); dz=torch.randn(N,H,H,co,device='cuda',generator=g)
    nk=k*k*ci*co
    dkb=torch.zeros(N,nk,device='cuda')
    t_old=med(lambda: ops.conv2d_wgrad(x,dz,(k,k,ci,co),pad_top=k//2,pad_left=k//2))
    try:
        t_new=med(lambda: ops.grouped_conv2d_wgrad(x,dz,(k,k,ci,co),dkb,pad_top=k//2,pad_left=k//2))
        dw=ops.conv2d_wgrad(x,dz,(k,k,ci,co),pad_top=k//2,pad_left=k//2)
        err=float((dkb.sum(0).view(k,k,ci,co)-dw).norm()/dw.norm())
    except Exception as e:
        t_new=float('nan'); err=str(e)[:80]
    print('k%d %2d->%2d: conv_wgrad %.3f ms, grouped mfma wgrad %.3f ms (mfma route %s) rel diff %s' % (k,ci,co,t_old,t_new,ops.grouped_uses_mfma((N,H,H,ci),(k,k,ci,co),(H,H),'wgrad'),err))
