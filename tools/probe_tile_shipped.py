"""Tile size of the 13- and 15-tap layers at the SHIPPED training shapes (experiments/hpnn.json: batch 50, grids of 192..384 points per side): forward, weight
gradient and fused backward with 32- and 64-point tiles forced, FFT transforms.  pick_tile decides on ONE image (a sample's arithmetic must not depend on its
batch neighbours); until round 6 it asked for >= 36 64-point tiles per image, which sends these layers to 32-point tiles (3.2 spectrum values per output pixel
at 15 taps instead of 1.6) for every grid below ~300 points.   python tools/probe_tile_shipped.py [N]"""
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


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    ops.set_spectral_mode('force')
    print('%-24s | %-22s | %-22s | %-22s' % ('layer (batch %d)' % N, 'forward ms T32 / T64', 'wgrad ms T32 / T64', 'fused bwd ms T32 / T64'))
    tot = {32: [0.0] * 3, 64: [0.0] * 3}
    for (k, ci, co) in ((15, 32, 32), (13, 28, 28), (11, 16, 32), (9, 24, 24)):
        for (H, W) in ((192, 192), (210, 250), (288, 288), (300, 384), (384, 384)):
            x = torch.randn(N, H, W, ci, device='cuda')
            dz = torch.randn(N, H, W, co, device='cuda')
            w = torch.randn(k, k, ci, co, device='cuda') * 0.01
            wt = ops.flip_transpose_weights(w)
            b = torch.zeros(co, device='cuda')
            y = torch.empty(N, H, W, co, device='cuda')
            dw = torch.empty_like(w)
            p = k // 2
            r = {}
            for T in (32, 64):
                ops.set_spectral_tile(T)
                tf = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', out=y))
                tw = timeit(lambda: ops.conv2d_wgrad(x, dz, w.shape, pad_top=p, pad_left=p, out=dw))
                tb = timeit(lambda: ops.conv2d_bwd_fused(x, dz, w.shape, wt, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=None))
                r[T] = (tf, tw, tb)
                for i, v in enumerate(r[T]):
                    tot[T][i] += v
            print('k%2d %2d->%2d %3dx%3d       | %8.3f / %8.3f    | %8.3f / %8.3f    | %8.3f / %8.3f' % (k, ci, co, H, W, r[32][0], r[64][0], r[32][1], r[64][1], r[32][2], r[64][2]), flush=True)
            del x, dz, y
    print('sums: forward %.2f / %.2f, wgrad %.2f / %.2f, fused backward %.2f / %.2f ms' % (tot[32][0], tot[64][0], tot[32][1], tot[64][1], tot[32][2], tot[64][2]))


if __name__ == '__main__':
    main()
