"""Launches the dominant conv layer (15x15, 32->32, 8 x 1024^2, fwd) a few times: target of the rocprofv3 --pmc passes that
measure its HBM traffic (FETCH_SIZE / WRITE_SIZE)."""
import sys

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from poisson_cnn_amd import ops

N, H, W, k, ci, co = 8, 1024, 1024, 15, 32, 32
x = torch.randn(N, H, W, ci, device='cuda')
w = torch.randn(k, k, ci, co, device='cuda') * 0.01
b = torch.zeros(co, device='cuda')
y = torch.empty(N, H, W, co, device='cuda')
dz = torch.randn(N, H, W, co, device='cuda')
dw = torch.empty_like(w)
for _ in range(3):
    ops.conv2d_fwd(x, w, b, pad_top=k // 2, pad_left=k // 2, act='leaky_relu', out=y)
    ops.conv2d_wgrad(x, dz, w.shape, pad_top=k // 2, pad_left=k // 2, out=dw)
torch.cuda.synchronize()
print('algorithmic bytes per conv launch: %.1f MB' % (4.0 * (N * H * W * (ci + co) + k * k * ci * co) / 1e6))
