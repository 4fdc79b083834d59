"""Only the forward convolution of one layer on the spectral route (for per-kernel rocprofv3 timings of the three passes without the backward
kernels mixed in).   python tools/probe_fwd_only.py k Cin Cout H [N [tile [iters]]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
k, ci, co, H = a[:4]
N = a[4] if len(a) > 4 else 8
tile = a[5] if len(a) > 5 else 32
iters = a[6] if len(a) > 6 else 10
ops.set_spectral_mode('force')
ops.set_spectral_tile(tile)
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(N, H, H, ci, device='cuda', generator=g)
w = torch.randn(k, k, ci, co, device='cuda', generator=g) * 0.02
for _ in range(iters):
    y = ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, act='leaky_relu')
torch.cuda.synchronize()
print('ok', float(y.abs().max()))
