"""Times the two routes of the Dirichlet FD Poisson solve (pcnn_fd_poisson_dst: DST-I as fp64 matrix-core GEMMs, 8 n^3 FLOP; pcnn_fd_poisson_fft:
rocFFT on the odd extension) per sample at 256^2 ... 2048^2, HIP-event medians after a warm-up call (plan creation, rocFFT kernel compilation)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd.dataset import _kernels as K


def med(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts))


g = torch.Generator(device='cuda').manual_seed(0)
print('%-12s %6s %14s %14s %10s' % ('grid', 'batch', 'gemm ms/sample', 'fft ms/sample', 'rel diff'))
for H, N in ((256, 32), (512, 32), (1024, 8), (1536, 4), (2048, 2)):
    rhs = torch.randn(N, H, H, device='cuda', generator=g)
    e = [torch.randn(N, H, device='cuda', generator=g) for _ in range(4)]
    dx = torch.rand(N, device='cuda', generator=g) * 4.5e-2 + 5e-3
    tg = med(lambda: K.fd_poisson_dst(rhs, *e, dx, solver='gemm')) / N
    tf = med(lambda: K.fd_poisson_dst(rhs, *e, dx, solver='fft')) / N
    a, b = K.fd_poisson_dst(rhs, *e, dx, solver='gemm'), K.fd_poisson_dst(rhs, *e, dx, solver='fft')
    print('%-12s %6d %14.3f %14.3f %10.2g' % ('%dx%d' % (H, H), N, tg, tf, float((a.double() - b.double()).norm() / a.double().norm())))
