import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
out = torch.zeros(8, 1024, 1024, 32, device='cuda')
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for f in (2, 4, 16):
    x = torch.randn(8, 1024 // f, 1024 // f, 32, device='cuda', generator=g)
    k = torch.randn(f, f, 32, 32, device='cuda', generator=g) * 0.05
    b = torch.zeros(32, device='cuda')
    for beta in (0.0, 1.0):
        fn = lambda: ops.deconv_fwd(x, k, b, (1024, 1024), f, alpha=0.1, beta=beta, out=out)
        fn(); torch.cuda.synchronize(); s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        print('deconv f=%d beta=%g: %.3f ms' % (f, beta, s.elapsed_time(e) / 10), flush=True)
xc = torch.randn(8, 32, 32, 32, device='cuda', generator=g)
for beta in (0.0, 1.0):
    fn = lambda: ops.resize_fwd(xc, (1024, 1024), 'bicubic', alpha=0.1, beta=beta, out=out)
    fn(); torch.cuda.synchronize(); s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print('resize bicubic beta=%g: %.3f ms' % (beta, s.elapsed_time(e) / 10), flush=True)
