import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
for k in (15, 7):
    x = torch.randn(8, 1024, 1024, 32, device='cuda', generator=g)
    w = torch.randn(k, k, 32, 32, device='cuda', generator=g) * 0.05
    b = torch.zeros(32, device='cuda')
    y = torch.empty_like(x); a = torch.empty_like(x)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, kw in (('plain', {}), ('residual', dict(residual=x)), ('residual+act_out', dict(residual=x, act_out=a))):
        fn = lambda: ops.conv2d_fwd(x, w, b, pad_top=k // 2, pad_left=k // 2, act='leaky_relu', out=y, **kw)
        fn(); torch.cuda.synchronize(); s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        print('k=%d %-18s %.3f ms' % (k, name, s.elapsed_time(e) / 10), flush=True)
