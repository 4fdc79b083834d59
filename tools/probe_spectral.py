"""Times one convolution layer at 8 x 1024^2 through the three routes (direct fp32 MFMA, direct 3 x fp16 split, tiled spectral):
forward, data gradient (same kernel) and weight gradient.  GPU box only.   python tools/probe_spectral.py [k Cin Cout [N H]]"""
import os
import sys
import time

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
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[:3])] if len(a) >= 3 else [(15, 32, 32), (13, 28, 28), (9, 24, 24), (7, 32, 32), (7, 64, 32), (5, 20, 16), (11, 16, 32)]
    N, H = (a[3], a[4]) if len(a) >= 5 else (8, 1024)
    g = torch.Generator(device='cuda').manual_seed(0)
    for k, ci, co in shapes:
        x = torch.randn(N, H, H, ci, device='cuda', generator=g)
        w = torch.randn(k, k, ci, co, device='cuda', generator=g) * 0.02
        dz = torch.randn(N, H, H, co, device='cuda', generator=g)
        p = k // 2
        flop = 2.0 * N * H * H * k * k * ci * co
        res = {}
        ref = None
        for name, math, spec in (('direct fp32', 'fp32', 'off'), ('direct split', 'split_f16', 'off'), ('spectral', 'fp32', 'force')):
            ops.set_math_mode(math); ops.set_spectral_mode(spec)
            tf = timeit(lambda: ops.conv2d_fwd(x, w, None, pad_top=p, pad_left=p))
            tw = timeit(lambda: ops.conv2d_wgrad(x, dz, w.shape, pad_top=p, pad_left=p))
            y = ops.conv2d_fwd(x, w, None, pad_top=p, pad_left=p)
            dw = ops.conv2d_wgrad(x, dz, w.shape, pad_top=p, pad_left=p)
            if ref is None:
                ref = (y.double(), dw.double())
            ey = float((y.double() - ref[0]).norm() / ref[0].norm()); ew = float((dw.double() - ref[1]).norm() / ref[1].norm())
            res[name] = (tf, tw)
            print('k=%2d %2d->%2d %-12s fwd %7.3f ms (%6.1f TF-eq)  wgrad %7.3f ms (%6.1f TF-eq)   vs direct fp32: y %.2e dw %.2e'
                  % (k, ci, co, name, tf, flop / tf / 1e9, tw, flop / tw / 1e9, ey, ew), flush=True)
        del x, w, dz
    ops.set_math_mode('fp32'); ops.set_spectral_mode('auto')


if __name__ == '__main__':
    main()
