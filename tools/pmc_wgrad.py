"""Launches the 15x15 32->32 weight-gradient of the 8 x 1024^2 workload a few times (math mode from PCNN_MATH): target of the
rocprofv3 --pmc passes that measure its L2 hit rate and fabric traffic."""
import sys

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from poisson_cnn_amd import ops

N, H, W, k, ci, co = 8, 1024, 1024, 15, 32, 32
x = torch.randn(N, H, W, ci, device='cuda')
dz = torch.randn(N, H, W, co, device='cuda')
dw = torch.empty(k, k, ci, co, device='cuda')
for _ in range(3):
    ops.conv2d_wgrad(x, dz, dw.shape, pad_top=k // 2, pad_left=k // 2, out=dw)
torch.cuda.synchronize()
print('algorithmic bytes per launch: %.1f MB' % (4.0 * (N * H * W * (ci + co) + k * k * ci * co) / 1e6))
