"""5 x 5, 20 -> 16 at 8 x 1024^2 (final/stage4/conv): the route the library picks (direct fp32 MFMA) against the spectral route forced."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops
def med(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))
g = torch.Generator(device='cuda').manual_seed(0)
for (k, ci, co) in ((5, 20, 16), (5, 16, 16), (7, 24, 20)):
    x = torch.randn(8, 1024, 1024, ci, device='cuda', generator=g); w = torch.randn(k, k, ci, co, device='cuda', generator=g) * 0.05
    dz = torch.randn(8, 1024, 1024, co, device='cuda', generator=g)
    for mode in ('auto', 'force'):
        ops.set_spectral_mode(mode)
        f = med(lambda: ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, act='leaky_relu'))
        wf = ops.flip_transpose_weights(w); dw = torch.zeros_like(w)
        def bwd():
            out = ops.conv2d_bwd_fused(x, dz, w.shape, wf, pad_top=k // 2, pad_left=k // 2, dw=dw)
            if out is None:
                ops.conv2d_wgrad(x, dz, w.shape, pad_top=k // 2, pad_left=k // 2, out=dw)
                ops.conv2d_fwd(dz, wf, None, pad_top=k - 1 - k // 2, pad_left=k - 1 - k // 2)
        b = med(bwd)
        print('k%d %d->%d %-5s fwd %.3f ms  bwd %.3f ms' % (k, ci, co, mode, f, b))
    ops.set_spectral_mode('auto')
