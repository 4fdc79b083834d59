"""A/B of the transform kernel families of the spectral route (PCNN_XFORM_MFMA: DFT as a GEMM on the matrix cores; PCNN_XFORM_FFT: in-register
FFTs on the vector ALUs) on single layers at 8 x 1024^2: forward, weight gradient, fused backward.  GPU box only.
  python tools/probe_xform.py [N] [tile]      tile: 0 = pick_tile's choice (default), 32 / 64 = forced
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel averages."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops


def timeit(fn, iters=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ops.set_spectral_tile(tile)
    shapes = [(7, 32, 32, 1024), (5, 32, 32, 1024), (9, 24, 24, 1024), (7, 64, 32, 1024), (11, 16, 32, 1024), (15, 32, 32, 1024), (13, 28, 28, 1024), (15, 3, 4, 1024),
              (5, 16, 16, 1024), (11, 32, 32, 512), (7, 32, 32, 128)]
    print('tile %s, batch %d' % (tile or 'auto', N))
    print('%-20s | %-26s | %-26s | %-26s' % ('layer', 'forward ms mfma / fft', 'wgrad ms mfma / fft', 'fused bwd ms mfma / fft'))
    tot = {'mfma': [0.0, 0.0, 0.0], 'fft': [0.0, 0.0, 0.0]}
    for (k, ci, co, hw) in shapes:
        x = torch.randn(N, hw, hw, ci, device='cuda')
        dz = torch.randn(N, hw, hw, co, device='cuda')
        w = torch.randn(k, k, ci, co, device='cuda') * 0.01
        wt = ops.flip_transpose_weights(w)
        b = torch.zeros(co, device='cuda')
        y = torch.empty(N, hw, hw, co, device='cuda')
        dw = torch.empty_like(w)
        p = k // 2
        r = {}
        for xf in ('mfma', 'fft'):
            ops.set_spectral_transform(xf)
            tf = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', out=y))
            tw = timeit(lambda: ops.conv2d_wgrad(x, dz, w.shape, pad_top=p, pad_left=p, out=dw))
            fused = ops.conv2d_bwd_fused(x, dz, w.shape, wt, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=None)
            tb = timeit(lambda: ops.conv2d_bwd_fused(x, dz, w.shape, wt, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=None)) if fused is not None else float('nan')
            r[xf] = (tf, tw, tb)
            for i, v in enumerate((tf, tw, tb)):
                tot[xf][i] += v if v == v else 0.0
        print('k%2d %2d->%2d @%4d     | %10.3f / %10.3f    | %10.3f / %10.3f    | %10.3f / %10.3f' % (k, ci, co, hw, r['mfma'][0], r['fft'][0], r['mfma'][1], r['fft'][1],
                                                                                                  r['mfma'][2], r['fft'][2]), flush=True)
        del x, dz, y
    print('sums: forward %.2f / %.2f, wgrad %.2f / %.2f, fused backward %.2f / %.2f ms' % (tot['mfma'][0], tot['fft'][0], tot['mfma'][1], tot['fft'][1], tot['mfma'][2], tot['fft'][2]))


if __name__ == '__main__':
    main()
