"""CPU oracle for the dataset path (poisson_CNN/dataset): fp64 numpy/scipy restatements.

TEST INFRASTRUCTURE ONLY - see oracle/np_ops.py header.  pyamg is not installed here; `multigrid_poisson_solve` restates
the LINEAR SYSTEM the reference hands to pyamg (dataset/solvers/multigrid.py:98-150: A = pyamg.gallery.poisson((H-2, W-2)),
b = poisson_RHS(...)) and solves it with a sparse direct solver, which is what the reference's tol=1e-10 multigrid
iteration converges to.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import np_ops


def poisson_matrix(m, n):
    """pyamg.gallery.poisson((m, n)): 5-point stencil, 4 on the diagonal, -1 for the four neighbours, C-order unknowns."""
    Tm = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(m, m))
    Tn = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(n, n))
    return (sp.kron(Tm, sp.eye(n)) + sp.kron(sp.eye(m), Tn)).tocsc()


def poisson_RHS(F, boundaries, h):
    """dataset/solvers/cholesky.py:45-119.  F (N,H,W); boundaries dict of (N,len); h (N,)."""
    F = np.array(F, dtype=np.float64, copy=True)
    for i in range(F.shape[0]):
        F[i] = -h[i] ** 2 * F[i]
        F[i, 1:-1, 1] += boundaries['bottom'][i, 1:-1]
        F[i, 1:-1, -2] += boundaries['top'][i, 1:-1]
        F[i, 1, 1:-1] += boundaries['left'][i, 1:-1]
        F[i, -2, 1:-1] += boundaries['right'][i, 1:-1]
    return F[:, 1:-1, 1:-1].reshape(F.shape[0], -1)


def multigrid_poisson_solve(rhses, boundaries, dx):
    """dataset/solvers/multigrid.py:98-150 with the iterative solve replaced by its limit (direct sparse solve)."""
    N, H, W = rhses.shape
    b = poisson_RHS(rhses, boundaries, dx)
    lu = spla.splu(poisson_matrix(H - 2, W - 2))
    out = np.zeros((N, H, W))
    for k in range(N):
        out[k, 1:-1, 1:-1] = lu.solve(b[k]).reshape(H - 2, W - 2)
    out[:, :, -1] = boundaries['top']
    out[:, :, 0] = boundaries['bottom']
    out[:, 0, :] = boundaries['left']
    out[:, -1, :] = boundaries['right']
    return out


def generate_smooth_function(grid_size, sin_coeff=None, cos_coeff=None):
    """dataset/utils/generate_smooth_function.py:45-62: einsum('AB,aA,bB->ab') of sin / cos tables on linspace(0, pi, n)."""
    H, W = grid_size
    x, y = np.linspace(0, np.pi, H), np.linspace(0, np.pi, W)
    out = np.zeros((H, W))
    for c, f in ((sin_coeff, np.sin), (cos_coeff, np.cos)):
        if c is not None:
            ka, kb = c.shape
            out += np.einsum('AB,aA,bB->ab', c, f(np.outer(x, np.arange(1, ka + 1))), f(np.outer(y, np.arange(1, kb + 1))))
    return out


def image_resize_legacy_bicubic(x, out_hw):
    """dataset/utils/image_resize.py:5-30: tf.compat.v1.image.resize_images(BICUBIC, align_corners=True) on (N,C,h,w)."""
    return np_ops.resize2d(x, out_hw, 'bicubic', half_pixel=False, align_corners=True)


def polynomial_and_second_derivative(roots, x):
    """p(x) = prod (x + r) via explicit polynomial coefficients (numpy.polynomial) - independent of the product-rule form."""
    P = np.polynomial.Polynomial.fromroots(-np.asarray(roots, dtype=np.float64))
    return P(x), P.deriv(2)(x)


def five_point_laplacian(u, dx):
    """(u[i+1,j] + u[i-1,j] + u[i,j+1] + u[i,j-1] - 4u)/dx^2 on the interior, uniform spacing."""
    return (u[..., 2:, 1:-1] + u[..., :-2, 1:-1] + u[..., 1:-1, 2:] + u[..., 1:-1, :-2] - 4 * u[..., 1:-1, 1:-1]) / dx ** 2
