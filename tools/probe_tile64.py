"""32-point against 64-point tiles of the spectral route, layer by layer at the c4 workload size: forward, weight gradient and the fused
backward.  GPU box only.   python tools/probe_tile64.py [k Cin Cout H [N [tile]]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops  # noqa: E402


def timeit(fn, iters=4):
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
    shapes = [tuple(a[:4])] if len(a) >= 4 else [(15, 32, 32, 1024), (13, 28, 28, 1024), (13, 32, 28, 1024), (11, 16, 32, 1024), (9, 24, 24, 1024),
                                                  (9, 28, 24, 1024), (11, 32, 32, 512), (9, 32, 32, 342)]
    N = a[4] if len(a) >= 5 else 8
    tiles = (a[5],) if len(a) >= 6 else (32, 64)
    ops.set_spectral_mode('force')
    g = torch.Generator(device='cuda').manual_seed(0)
    for k, ci, co, H in shapes:
        x = torch.randn(N, H, H, ci, device='cuda', generator=g)
        w = torch.randn(k, k, ci, co, device='cuda', generator=g) * 0.02
        dz = torch.randn(N, H, H, co, device='cuda', generator=g)
        wf = ops.flip_transpose_weights(w)
        dw = torch.empty_like(w)
        p = k // 2
        ref = None
        for tile in tiles:
            ops.set_spectral_tile(tile)
            tf = timeit(lambda: ops.conv2d_fwd(x, w, None, pad_top=p, pad_left=p, act='leaky_relu'))
            tw = timeit(lambda: ops.conv2d_wgrad(x, dz, w.shape, pad_top=p, pad_left=p))
            tb = timeit(lambda: ops.conv2d_bwd_fused(x, dz, w.shape, wf, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=None))
            y = ops.conv2d_fwd(x, w, None, pad_top=p, pad_left=p)
            if ref is None:
                ref = y.double()
            err = float((y.double() - ref).norm() / ref.norm())
            print('k=%2d %2d->%2d @%4d  T=%d  fwd %7.3f ms  wgrad %7.3f ms  fused bwd %7.3f ms   vs T=32: %.2e' % (k, ci, co, H, tile, tf, tw, tb, err), flush=True)
        del x, w, dz
    ops.set_spectral_tile(0)
    ops.set_spectral_mode('auto')


if __name__ == '__main__':
    main()
