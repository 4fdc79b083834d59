"""CPU oracle: restatement of losses/loss_wrapper.py, losses/integral_loss.py and
losses/physics_informed_loss.py.  Works on numpy fp64 arrays or fp64 torch tensors (autograd twin).

TEST INFRASTRUCTURE ONLY - see oracle/np_ops.py header.

The one reference-authored known answer for this path (losses/integral_loss.py:181-203:
integral of (xyz)^(2/3) over [0,1]x[0,2]x[1,3.5] = 4.84711 within 1 %) is pinned in
tests/test_oracle_golden.py through `integral_lp` (n-dimensional form).

Restatement note: the reference obtains the multilinear interpolant at each Gauss-Legendre point by
solving a 2^d x 2^d linear system per point in fp32 (integral_loss.py:150-169).  The unique multilinear
interpolant through the 2^d enclosing grid values is restated here in closed form (tensor-product
hat weights); neighbour indices are found in float32 exactly as integral_loss.py:5-15 does.
"""
import numpy as np

from . import np_ops

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def neighbour_indices(n, q):
    """find_neighbouring_indices_along_axis (losses/integral_loss.py:5-15), float32 arithmetic."""
    dx = np.float32(2.0 / (n - 1))
    v = (q.astype(np.float32) + np.float32(1.0)) / dx
    lo = np.floor(v).astype(np.int64)
    hi = np.ceil(v).astype(np.int64)
    hi = np.where(hi == lo, hi + 1, hi)
    return lo, hi


def axis_weights(n, n_quad):
    """Dense (n_quad, n) matrix B with B[q, lo]=1-t, B[q, hi]=t, and the GL weights."""
    q, w = np.polynomial.legendre.leggauss(n_quad)
    q32 = q.astype(np.float32)
    lo, hi = neighbour_indices(n, q32)
    coords = np.linspace(-1.0, 1.0, n)
    t = (q32.astype(np.float64) - coords[lo]) / (coords[hi] - coords[lo])
    B = np.zeros((n_quad, n))
    B[np.arange(n_quad), lo] += 1.0 - t
    B[np.arange(n_quad), hi] += t
    return B, w.astype(np.float32).astype(np.float64)


def integral_weight_map(shape, n_quadpts):
    """G such that the GL integral of the multilinear interpolant of a grid function v over [-1,1]^d is
    sum(G * v).  2-D: G = (w_y B_y)^T (w_x B_x)."""
    if isinstance(n_quadpts, int):
        n_quadpts = [n_quadpts] * len(shape)
    vecs = []
    for n, nq in zip(shape, n_quadpts):
        B, w = axis_weights(n, nq)
        vecs.append(w @ B)
    G = vecs[0]
    for v in vecs[1:]:
        G = np.multiply.outer(G, v)
    return G


def integral_lp(y_true, y_pred, n_quadpts, p=2, dx=None):
    """integral_loss.__call__ (losses/integral_loss.py:126-179) for (N,C,*spatial) inputs; returns (N,C)."""
    sp = y_true.shape[2:]
    G = integral_weight_map(sp, n_quadpts)
    d = (y_true - y_pred) ** p
    if _is_torch(d):
        G = torch.as_tensor(G, dtype=d.dtype)
        val = (d * G).sum(dim=tuple(range(2, d.ndim)))
    else:
        val = (d * G).sum(axis=tuple(range(2, d.ndim)))
    nd = len(sp)
    if dx is None:
        vol = 2.0 ** nd
        return val * vol / 2 ** nd
    sizes = np.asarray(dx) * (np.array(sp) - 1)            # (N, nd)
    return val * np.prod(sizes, axis=1)[:, None] / 2 ** nd


def linear_operator_loss(rhs, solution, dx, stencil_sizes, orders, normalize=False,
                         inputs_have_max_domain_size_squared_normalization=False):
    """losses/physics_informed_loss.py:35-50.  dx is (N,2)."""
    ops = _ops_for(solution)
    st = np_ops.build_fd_coefficients(stencil_sizes, orders, 2)
    dxn = np.asarray(dx.detach().numpy() if _is_torch(dx) else dx, dtype=np.float64)
    if inputs_have_max_domain_size_squared_normalization:
        sizes = dxn * (np.array(solution.shape[2:]) - 1)
        q = (sizes.max(axis=1, keepdims=True) / dxn) ** 2
    else:
        q = 1.0 / dxn ** 2
    kern = np.einsum('dij,bd->bij', st, q)
    ly, lx = st.shape[1] // 2, st.shape[2] // 2
    H, W = solution.shape[2:]
    tot = 0.0
    for b in range(solution.shape[0]):
        comp = ops.conv2d_valid(solution[b:b + 1], kern[b][:, :, None, None])
        err = (rhs[b:b + 1, :, ly:H - ly, lx:W - lx] - comp) ** 2
        if normalize:
            err = err / (abs(rhs[b]).max() ** 2)
        tot = tot + err.sum()
    return tot / (solution.shape[0] * solution.shape[1] * (H - 2 * ly) * (W - 2 * lx))


def _ops_for(x):
    if _is_torch(x):
        from . import torch_twin
        return torch_twin
    return np_ops


class loss_wrapper:
    """losses/loss_wrapper.py:6-71."""

    def __init__(self, ndims, integral_loss_weight, integral_loss_config, physics_informed_loss_weight,
                 physics_informed_loss_config, data_format='channels_first', mse_loss_weight=0.0, mae_loss_weight=0.0,
                 scale_sample_loss_by_target_peak_magnitude=False, global_batch_size=None):
        assert ndims == 2 and data_format == 'channels_first'
        self.w_int, self.w_pi, self.w_mse, self.w_mae = integral_loss_weight, physics_informed_loss_weight, mse_loss_weight, mae_loss_weight
        self.int_cfg = dict(integral_loss_config)
        self.pi_cfg = {k: v for k, v in physics_informed_loss_config.items() if k not in ('ndims', 'data_format')}
        self.scale = scale_sample_loss_by_target_peak_magnitude
        self.global_batch_size = global_batch_size

    def _supervised(self, per_sample, power, peaks, N):
        gbs = N if self.global_batch_size is None else self.global_batch_size
        wts = 1.0 / peaks ** power if self.scale else 1.0
        return (wts * per_sample).sum() / gbs

    def __call__(self, y_true, y_pred, rhs, dx):
        N = y_true.shape[0]
        red = tuple(range(1, y_true.ndim))
        if _is_torch(y_pred):
            y_true = torch.as_tensor(y_true, dtype=y_pred.dtype)
            peaks = y_true.abs().amax(dim=red) if self.scale else None
            mean = lambda v: v.mean(dim=red)
        else:
            peaks = np.abs(y_true).max(axis=red) if self.scale else None
            mean = lambda v: v.mean(axis=red)
        loss = 0.0
        if self.w_mse != 0.0:
            loss = loss + self.w_mse * self._supervised(mean((y_true - y_pred) ** 2), 2.0, peaks, N)
        if self.w_mae != 0.0:
            loss = loss + self.w_mae * self._supervised(mean(abs(y_true - y_pred)), 1.0, peaks, N)
        if self.w_pi != 0.0:
            loss = loss + self.w_pi * linear_operator_loss(rhs, y_pred, dx, **self.pi_cfg)
        if self.w_int != 0.0:
            p = self.int_cfg.get('Lp_norm_power', 2)
            per = integral_lp(y_true, y_pred, self.int_cfg['n_quadpts'], p)   # (N,C); reduce_mean over (1,C)
            per = per.mean(dim=1) if _is_torch(per) else per.mean(axis=1)
            loss = loss + self.w_int * self._supervised(per, float(p), peaks, N)
        return loss
