"""One narrow layer at 8 x 1024^2 through the narrow-convolution kernels, for rocprofv3 counter passes; GPU box only.
python tools/probe_small.py [k Cin Cout [iters]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    k, ci, co = a[:3] if len(a) >= 3 else (3, 8, 8)
    iters = a[3] if len(a) >= 4 else 10
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(8, 1024, 1024, ci, device='cuda', generator=g)
    dz = torch.randn(8, 1024, 1024, co, device='cuda', generator=g)
    w = torch.randn(k, k, ci, co, device='cuda', generator=g) * 0.1
    y = torch.empty(8, 1024, 1024, co, device='cuda')
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fn in (('fwd', lambda: ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, act='leaky_relu', out=y)),
                     ('wgrad', lambda: ops.conv2d_wgrad(x, dz, w.shape, pad_top=k // 2, pad_left=k // 2))):
        fn()
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print('%s k=%d %d->%d: %.3f ms, %.0f GB/s algorithmic' % (name, k, ci, co, ms, (x.numel() + y.numel()) * 4 / ms / 1e6), flush=True)


if __name__ == '__main__':
    main()
