import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ctypes import c_int, c_int64
from poisson_cnn_amd import ops
from poisson_cnn_amd.ops import _p, handle
from poisson_cnn_amd.dataset import _kernels as K
rng = np.random.default_rng(0)
M, Nn, Kk, B = 43, 36, 43, 3
A = torch.tensor(rng.standard_normal((M, Kk)), device='cuda'); Bm = torch.tensor(rng.standard_normal((B, Kk, Nn)), device='cuda')
C = torch.empty((B, M, Nn), dtype=torch.float64, device='cuda')
handle().call('pcnn_batched_gemm_f64', c_int(B), c_int(M), c_int(Nn), c_int(Kk), _p(A), c_int64(0), c_int(Kk), _p(Bm), c_int64(Kk * Nn), c_int(Nn), _p(C), c_int64(M * Nn), c_int(Nn))
print('gemm A(bcast) err', float((C - A[None] @ Bm).abs().max()))
A2 = torch.tensor(rng.standard_normal((B, M, Kk)), device='cuda'); B2 = torch.tensor(rng.standard_normal((Kk, Nn)), device='cuda')
handle().call('pcnn_batched_gemm_f64', c_int(B), c_int(M), c_int(Nn), c_int(Kk), _p(A2), c_int64(M * Kk), c_int(Kk), _p(B2), c_int64(0), c_int(Nn), _p(C), c_int64(M * Nn), c_int(Nn))
print('gemm B(bcast) err', float((C - A2 @ B2[None]).abs().max()))
Vi, V, lam, ab = K.axis_decomposition(45, False, False, 'cuda')
print('decomp', float((V @ torch.diag(lam) @ Vi).diagonal().mean()), float((Vi @ V - torch.eye(43, dtype=torch.float64, device='cuda')).abs().max()), Vi.dtype, Vi.is_contiguous(), V.is_contiguous())
