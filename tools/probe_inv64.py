"""64-point inverse transform variants on single layers at 8 x 1024^2 (forced 64-point tiles): plain forward, forward with tanh + residual epilogue, fused
backward with the producer's activation backward in the data-gradient epilogue (POST).  A/B of library builds:  PCNN_LIBRARY=... python tools/probe_inv64.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops


def timeit(fn, iters=8):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


ops.set_spectral_mode('force')
ops.set_spectral_tile(64)
for (k, ci, co) in [(15, 32, 32), (13, 28, 28), (11, 32, 32)]:
    x = torch.randn(8, 1024, 1024, ci, device='cuda'); w = torch.randn(k, k, ci, co, device='cuda') * 0.01
    wf = w.flip(0, 1).permute(0, 1, 3, 2).contiguous()
    y = torch.empty(8, 1024, 1024, co, device='cuda'); dy = torch.randn_like(y); act = torch.randn_like(x); res = torch.randn_like(y)
    dw = torch.empty_like(w); db = torch.zeros(ci, device='cuda')
    f = timeit(lambda: ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, out=y))
    fr = timeit(lambda: ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, out=y, residual=res, act='tanh'))
    b0 = timeit(lambda: ops.conv2d_bwd_fused(x, dy, tuple(w.shape), wf, pad_top=k // 2, pad_left=k // 2, dw=dw))
    def post():
        p = ops.Post(act, 'leaky_relu', db, want_raw=True)
        ops.conv2d_bwd_fused(x, dy, tuple(w.shape), wf, pad_top=k // 2, pad_left=k // 2, dw=dw, residual=act, post=p)
        assert p.applied
    b1 = timeit(post)
    print('k%2d %d->%d  fwd %.3f  fwd+tanh+res %.3f  fused bwd %.3f  fused bwd + post + res + raw %.3f ms' % (k, ci, co, f, fr, b0, b1))
