"""Times the per-sample-filter ("metalearning") convolution layer on its own: forward, data gradient, filter gradient of one launch each
(csrc/grouped_conv.hip), for the reference's example layer - 19 x 19, 3 -> 4 channels, batch 10 at 200 x 200 (layers/metalearning_conv.py:171-184
times exactly this layer) - and for the layer shapes of configs.hpnn_metalearning() at 200 x 200.  HIP-event times, median of `reps`."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops

SHAPES = [(19, 3, 4, 10, 200), (17, 4, 6, 10, 200), (15, 6, 8, 10, 200), (13, 8, 8, 10, 200), (13, 8, 8, 10, 100), (11, 16, 8, 10, 200), (7, 8, 6, 10, 200), (5, 6, 4, 10, 200),
          (3, 20, 17, 10, 200), (7, 32, 32, 10, 200)]


def med(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts))


def main():
    g = torch.Generator(device='cuda').manual_seed(0)
    print('%-28s %9s %9s %9s %9s   %s' % ('layer (k, Cin->Cout, N x HxW)', 'fwd ms', 'dgrad ms', 'wgrad ms', 'sum ms', 'fwd TFLOP/s'))
    for k, Cin, Cout, N, H in SHAPES:
        nk = k * k * Cin * Cout
        x = torch.randn(N, H, H, Cin, device='cuda', generator=g)
        kb = torch.randn(N, nk + Cout, device='cuda', generator=g) / np.sqrt(k * k * Cin)
        dz = torch.randn(N, H, H, Cout, device='cuda', generator=g)
        dkb = torch.zeros_like(kb)
        pt = k // 2
        ws = (k, k, Cin, Cout)
        f = med(lambda: ops.grouped_conv2d_fwd(x, kb, ws, kb[:, nk:], pad_top=pt, pad_left=pt, out_hw=(H, H), pad_mode='CONSTANT', act='leaky_relu'))
        d = med(lambda: ops.grouped_conv2d_fwd(dz, kb, ws, None, pad_top=k - 1 - pt, pad_left=k - 1 - pt, out_hw=(H, H), flip_transpose=True))
        w = med(lambda: ops.grouped_conv2d_wgrad(x, dz, ws, dkb, pad_top=pt, pad_left=pt))
        fl = 2.0 * N * H * H * nk
        print('%-28s %9.3f %9.3f %9.3f %9.3f   %.2f' % ('%dx%d %d->%d, %d x %dx%d' % (k, k, Cin, Cout, N, H, H), f, d, w, f + d + w, fl / f / 1e9))


if __name__ == '__main__':
    main()
