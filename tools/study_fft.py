"""Removal study of the FFT transform kernels (diagnostic library libpcnn_study.so, `make -C poisson_cnn_amd/csrc study`): one layer's forward convolution at
8 x 1024^2 with parts of the transform kernels compiled in but switched off by PCNN_FFT_STUDY bits (1 no stores, 2 no loads, 4 no FFT arithmetic, 8 no LDS
reads, 16 no LDS writes).  Run under rocprofv3 --kernel-trace --stats for the kernel's own time:   PCNN_LIBRARY=.../libpcnn_study.so PCNN_FFT_STUDY=<bits> python tools/study_fft.py k tile"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops

k = int(sys.argv[1]) if len(sys.argv) > 1 else 15
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ops.set_spectral_mode('force'); ops.set_spectral_tile(tile)
x = torch.randn(8, 1024, 1024, 32, device='cuda')
w = torch.randn(k, k, 32, 32, device='cuda') * 0.01
y = torch.empty(8, 1024, 1024, 32, device='cuda')
for _ in range(3):
    ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, out=y)
torch.cuda.synchronize()
t0 = time.perf_counter()
ITERS = int(os.environ.get("STUDY_ITERS", "10"))
for _ in range(ITERS):
    ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, out=y)
torch.cuda.synchronize()
print('study bits %s: k %d tile %d forward %.3f ms' % (os.environ.get('PCNN_FFT_STUDY', '0'), k, tile, (time.perf_counter() - t0) * 1e3 / ITERS))
