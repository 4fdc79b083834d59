"""The narrow 3x3 layers of the final stack at 8 x 1024^2: the vector-ALU kernel (conv_small_fwd_kernel, the route pcnn_conv2d_fwd takes) against the
grouped matrix-core kernel of the metalearning layers (gm_fwd_kernel, v_mfma_f32_4x4x1_16B_f32 with A-broadcast weights) called with ONE batch-shared
filter (w_sample_stride = 0) - VERDICT r4 item 3.  GPU box only."""
import ctypes
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops
from poisson_cnn_amd.ops import _p, conv_desc, _ld


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


N, H = 8, 1024
print('%-14s %12s %12s %10s %12s' % ('layer', 'valu ms', 'mfma4x4 ms', 'max |diff|', 'frac of HBM'))
for (k, ci, co) in [(3, 16, 12), (3, 12, 12), (3, 12, 8), (3, 8, 8), (3, 8, 4), (3, 4, 4), (3, 4, 1), (5, 16, 16)]:
    x = torch.randn(N, H, H, ci, device='cuda')
    w = torch.randn(k, k, ci, co, device='cuda') * 0.1
    b = torch.randn(co, device='cuda') * 0.1
    y = torch.empty(N, H, H, co, device='cuda')
    y2 = torch.empty(N, H, H, co, device='cuda')
    p = k // 2
    tv = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', out=y))
    d = conv_desc(x.shape, _ld(x), w.shape, (H, H), _ld(y2), p, p, 'CONSTANT', 0.0, act='leaky_relu')
    h = ops.handle()
    uses = h.lib.pcnn_grouped_conv2d_uses_mfma(ctypes.byref(d), 0)

    def grouped():
        h.call('pcnn_grouped_conv2d_fwd', ctypes.byref(d), _p(x), _p(w), 0, _p(b), 0, 0, _p(y2))
    tm = timeit(grouped) if uses else float('nan')
    diff = float((y - y2).abs().max()) if uses else float('nan')
    byts = 4.0 * N * H * H * (ci + co)
    print('k%d %2d->%2d      %12.3f %12.3f %10.2g   %5.2f / %5.2f' % (k, ci, co, tv, tm, diff, byts / tv / 1e-3 / 8e12, byts / tm / 1e-3 / 8e12), flush=True)
