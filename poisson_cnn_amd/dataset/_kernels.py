"""ctypes wrappers of the dataset kernels of libpcnn (include/pcnn.h, "dataset" section)."""
from ctypes import c_float, c_int, c_int64, c_void_p

import numpy as np
import torch

from .. import _lib, ops
from ..ops import _p, handle

_dst_cache = {}


def dst_matrices(n, device):
    """Device copies of the orthonormal DST-I matrix and eigenvalues for an n-point axis (cached per n: shapes change per batch)."""
    key = (n, str(device))
    if key not in _dst_cache:
        m = n - 2
        S = np.zeros((m, m), dtype=np.float64)
        lam = np.zeros(m, dtype=np.float64)
        rc = _lib.load().pcnn_dst_setup(c_int(n), S.ctypes.data_as(c_void_p), lam.ctypes.data_as(c_void_p))
        if rc != 0:
            raise RuntimeError('pcnn_dst_setup failed (%d)' % rc)
        _dst_cache[key] = (torch.tensor(S, device=device), torch.tensor(lam, device=device))
    return _dst_cache[key]


FFT_FROM = 2048          # solver='auto': rocFFT route from this many points per axis - measured crossover (profiles/r04_fd_solver_routes.txt: per sample
                         # 0.25 vs 0.35 ms at 1024^2, 0.80 vs 1.11 at 1536^2, 1.86 vs 1.66 at 2048^2; the odd extension has awkward lengths, 2046 = 2 3 11 31)


def fd_poisson_dst(rhs, left, right, bottom, top, dx, solver='auto'):
    """rhs (N,H,W); left/right (N,W); bottom/top (N,H); dx (N,) -> soln (N,H,W), all fp32 CUDA tensors.
    solver: 'gemm' (DST-I as fp64 matrix-core GEMMs, pcnn_fd_poisson_dst), 'fft' (rocFFT, pcnn_fd_poisson_fft) or 'auto' (fft from FFT_FROM
    points per axis; environment PCNN_FD_SOLVER overrides)."""
    import os
    from ctypes import c_size_t
    N, H, W = rhs.shape
    Sh, lh = dst_matrices(H, rhs.device)
    Sw, lw = dst_matrices(W, rhs.device)
    solver = os.environ.get('PCNN_FD_SOLVER', solver)
    if solver not in ('auto', 'gemm', 'fft'):
        raise ValueError("solver must be 'auto', 'gemm' or 'fft'")
    if solver == 'fft' or (solver == 'auto' and min(H, W) >= FFT_FROM):
        lib = _lib.load()
        lib.pcnn_fd_poisson_fft_workspace.restype = c_size_t
        ws = torch.empty((lib.pcnn_fd_poisson_fft_workspace(c_int(N), c_int(H), c_int(W)) + 7) // 8, dtype=torch.float64, device=rhs.device)
        soln = torch.empty_like(rhs)
        handle().call('pcnn_fd_poisson_fft', c_int(N), c_int(H), c_int(W), _p(rhs), _p(left), _p(right), _p(bottom), _p(top), _p(dx), _p(lh), _p(lw), _p(ws), _p(soln))
        return soln
    tmp = torch.empty(2 * N * (H - 2) * (W - 2), dtype=torch.float64, device=rhs.device)
    soln = torch.empty_like(rhs)
    handle().call('pcnn_fd_poisson_dst', c_int(N), c_int(H), c_int(W), _p(rhs), _p(left), _p(right), _p(bottom), _p(top), _p(dx),
                  _p(Sh), _p(lh), _p(Sw), _p(lw), _p(tmp), _p(soln))
    return soln


_axis_cache = {}


def axis_decomposition(n, neumann_lo, neumann_hi, device):
    """Eigen-decomposition M = V diag(lam) V^-1 of the 1-D operator (2 on the diagonal, -1 beside it, the mirrored neighbour counted twice
    in a Neumann end's row) on the unknowns of an n-point axis: Dirichlet ends are not unknowns.  With W = diag(1/2 at Neumann ends, 1
    elsewhere) W M is symmetric, so the generalised symmetric problem (W M) v = lam W v gives V with V^T W V = I, i.e. V^-1 = V^T W (fp64,
    host, cached per (n, end types); Dirichlet-Dirichlet reproduces the DST-I basis of pcnn_dst_setup).
    Returns device tensors (Vinv, V, lam) and the index range (a, b) of the unknowns."""
    key = (n, bool(neumann_lo), bool(neumann_hi), str(device))
    if key not in _axis_cache:
        import scipy.linalg
        a, b = (0 if neumann_lo else 1), (n - 1 if neumann_hi else n - 2)
        m = b - a + 1
        M = 2.0 * np.eye(m) - np.eye(m, k=1) - np.eye(m, k=-1)
        w = np.ones(m)
        if neumann_lo:
            M[0, 1] = -2.0
            w[0] = 0.5
        if neumann_hi:
            M[m - 1, m - 2] = -2.0
            w[m - 1] = 0.5
        Wm = np.diag(w)
        lam, V = scipy.linalg.eigh(Wm @ M, Wm)
        if neumann_lo and neumann_hi:
            lam[0] = 0.0                                        # exact null mode (constants)
        Vinv = V.T * w[None, :]
        # scipy hands V back in Fortran order and torch.tensor keeps those strides: the kernels read raw row-major memory
        _axis_cache[key] = (torch.tensor(np.ascontiguousarray(Vinv), device=device).contiguous(), torch.tensor(np.ascontiguousarray(V), device=device).contiguous(),
                            torch.tensor(np.ascontiguousarray(lam), device=device), (a, b))
    return _axis_cache[key]


def fd_poisson_mixed(rhs, left, right, bottom, top, dx, neumann=(False, False, False, False)):
    """5-point solve with per-edge Dirichlet / Neumann conditions; `neumann` = (left, right, bottom, top).  A Neumann edge's array holds
    du/dn (outward normal), a Dirichlet edge's its values.  rhs (N,H,W); left/right (N,W); bottom/top (N,H); dx (N,) -> soln (N,H,W)."""
    N, H, W = rhs.shape
    nl, nr, nb, nt = [bool(v) for v in neumann]
    Vih, Vh, lh, _ = axis_decomposition(H, nl, nr, rhs.device)
    Viw, Vw, lw, _ = axis_decomposition(W, nb, nt, rhs.device)
    mh, mw = Vh.shape[0], Vw.shape[0]
    key = ('T', W, nb, nt, str(rhs.device))
    if key not in _axis_cache:
        _axis_cache[key] = (Viw.t().contiguous(), Vw.t().contiguous())
    ViwT, VwT = _axis_cache[key]
    tmp = torch.empty(2 * N * mh * mw, dtype=torch.float64, device=rhs.device)
    soln = torch.empty_like(rhs)
    mask = int(nl) | (int(nr) << 1) | (int(nb) << 2) | (int(nt) << 3)
    handle().call('pcnn_fd_poisson_mixed', c_int(N), c_int(H), c_int(W), c_int(mask), _p(rhs.contiguous()), _p(left.contiguous()), _p(right.contiguous()),
                  _p(bottom.contiguous()), _p(top.contiguous()), _p(dx.contiguous()), _p(Vih), _p(Vh), _p(lh), _p(ViwT), _p(VwT), _p(lw), _p(tmp), _p(soln))
    return soln


def series_synthesis(coef, H, W, trig, out=None, accumulate=False):
    """coef (N,ka,kb) -> (N,H,W): sum c[A,B] f((A+1)x) f((B+1)y), f = sin (trig=0) / cos (trig=1)."""
    N, ka, kb = coef.shape
    if out is None:
        out = torch.empty((N, H, W), dtype=torch.float32, device=coef.device)
        accumulate = False
    handle().call('pcnn_series_synthesis', c_int(N), c_int(H), c_int(W), c_int(ka), c_int(kb), _p(coef.contiguous()), c_int(trig),
                  c_int(1 if accumulate else 0), _p(out))
    return out


def separable_sum(U, V, out=None, accumulate=False):
    """U (N,R,H), V (N,R,W) -> (N,H,W)"""
    N, R, H = U.shape
    W = V.shape[2]
    if out is None:
        out = torch.empty((N, H, W), dtype=torch.float32, device=U.device)
        accumulate = False
    handle().call('pcnn_separable_sum', c_int(N), c_int(H), c_int(W), c_int(R), _p(U.contiguous()), _p(V.contiguous()), c_int(1 if accumulate else 0), _p(out))
    return out


def set_max_magnitude(x, target):
    """In place: x[n] *= target[n]/max|x[n]|; returns the factors."""
    N = x.shape[0]
    f = torch.empty((N,), dtype=torch.float32, device=x.device)
    handle().call('pcnn_set_max_magnitude', c_int(N), c_int64(x.numel() // N), _p(target), _p(x), _p(f))
    return f


def scale_samples(x, s):
    N = x.shape[0]
    handle().call('pcnn_scale_samples', c_int(N), c_int64(x.numel() // N), _p(s), _p(x))
    return x


def max_abs_per_sample(x):
    """max|x[n]| via the loss partial-sum kernel (column 3)."""
    return ops.loss_partials(x, x, None)[:, 3].contiguous()


def resize_legacy_bicubic(x, out_hw):
    """tf.compat.v1.image.resize_images(BICUBIC, align_corners=True) of (N,h,w,C) (dataset/utils/image_resize.py:20)."""
    return ops.resize_fwd(x, out_hw, 'bicubic_legacy')
