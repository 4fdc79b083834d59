"""The inverse transform's epilogue variants on one layer at 8 x 1024^2: plain / + residual / + act_out / + both (forward), and the fused backward with and without
the producer's activation backward (POST) - what an extra input or output stream costs the spectral route.  GPU box only.  python tools/probe_epilogue_variants.py [k C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops  # noqa: E402


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(15, 32), (13, 28), (7, 32), (9, 24)]
    N, H = 8, 1024
    g = torch.Generator(device='cuda').manual_seed(0)
    for k, C in shapes:
        x = torch.randn(N, H, H, C, device='cuda', generator=g)
        w = torch.randn(k, k, C, C, device='cuda', generator=g) * 0.02
        b = torch.randn(C, device='cuda', generator=g) * 0.1
        res = torch.randn(N, H, H, C, device='cuda', generator=g)
        y, a = torch.empty_like(x), torch.empty_like(x)
        p = k // 2
        t0 = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', out=y))
        t1 = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', residual=res, out=y))
        t2 = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', out=y, act_out=a))
        t3 = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', residual=res, out=y, act_out=a))
        wf = ops.flip_transpose_weights(w)
        dw = torch.empty_like(w)
        tb = timeit(lambda: ops.conv2d_bwd_fused(x, res, w.shape, wf, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=None))
        tbr = timeit(lambda: ops.conv2d_bwd_fused(x, res, w.shape, wf, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=y))
        db = torch.empty(C, device='cuda')

        def post(raw, resid):
            ps = ops.Post(a, 'leaky_relu', db, raw)
            return ops.conv2d_bwd_fused(x, res, w.shape, wf, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=y if resid else None, post=ps)
        tp = timeit(lambda: post(False, False))
        tpr = timeit(lambda: post(True, False))
        tps = timeit(lambda: post(False, True))
        print('k=%2d C=%2d forward: plain %.3f | +residual %.3f | +act_out %.3f | +both %.3f ms   fused backward: plain %.3f | +residual %.3f | POST %.3f | POST+raw %.3f | POST+residual %.3f ms'
              % (k, C, t0, t1, t2, t3, tb, tbr, tp, tpr, tps), flush=True)


if __name__ == '__main__':
    main()
