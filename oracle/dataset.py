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


def _axis_operator(n, neumann_lo, neumann_hi):
    """1-D operator of the mixed-BC 5-point system on one axis' unknowns: -second difference, the mirrored neighbour counted twice in a
    Neumann end's row (second-order ghost node).  Returns (matrix, first unknown index, last unknown index, trapezoid weights)."""
    a, b = (0 if neumann_lo else 1), (n - 1 if neumann_hi else n - 2)
    m = b - a + 1
    M = sp.lil_matrix((m, m))
    M.setdiag(2.0)
    M.setdiag(-1.0, 1)
    M.setdiag(-1.0, -1)
    w = np.ones(m)
    if neumann_lo:
        M[0, 1] = -2.0
        w[0] = 0.5
    if neumann_hi:
        M[m - 1, m - 2] = -2.0
        w[m - 1] = 0.5
    return M.tocsr(), a, b, w


def mixed_bc_poisson_solve(rhses, boundaries, dx, neumann):
    """Reference for the mixed Dirichlet / Neumann 5-point solve: same grid and edge naming as
    multigrid_poisson_solve; neumann = dict edge -> bool.  A Neumann edge's array holds du/dn along the outward normal.  The all-Neumann
    system is singular: like Navier_Stokes_2D/solvers.py:258-259 it is closed with a Lagrange multiplier, here enforcing a zero
    trapezoidal integral (sum_ij w_i w_j u_ij = 0).  Sparse direct solve, one sample at a time."""
    N, H, W = rhses.shape
    Mh, i0, i1, wh = _axis_operator(H, neumann['left'], neumann['right'])
    Mw, j0, j1, ww = _axis_operator(W, neumann['bottom'], neumann['top'])
    mh, mw = i1 - i0 + 1, j1 - j0 + 1
    A = (sp.kron(Mh, sp.eye(mw)) + sp.kron(sp.eye(mh), Mw)).tocsc()
    singular = all(neumann[k] for k in ('left', 'right', 'bottom', 'top'))
    if singular:
        wvec = np.outer(wh, ww).reshape(-1)
        A = sp.bmat([[A, sp.csc_matrix(np.ones((mh * mw, 1)))], [sp.csc_matrix(wvec[None, :]), None]]).tocsc()
    lu = spla.splu(A)
    out = np.zeros((N, H, W))
    for n in range(N):
        h = dx[n]
        full = -h * h * np.array(rhses[n], dtype=np.float64)
        L, R, B, T = (np.asarray(boundaries[k][n], dtype=np.float64) for k in ('left', 'right', 'bottom', 'top'))
        if neumann['left']:
            full[0, :] += 2 * h * L
        else:
            full[1, :] += L
        if neumann['right']:
            full[-1, :] += 2 * h * R
        else:
            full[-2, :] += R
        if neumann['bottom']:
            full[:, 0] += 2 * h * B
        else:
            full[:, 1] += B
        if neumann['top']:
            full[:, -1] += 2 * h * T
        else:
            full[:, -2] += T
        b = full[i0:i1 + 1, j0:j1 + 1].reshape(-1)
        if singular:
            b = np.concatenate([b, [0.0]])
        u = lu.solve(b)[:mh * mw].reshape(mh, mw)
        # Dirichlet values: top, bottom, then left, right (the corner convention of multigrid.py:145-148)
        if not neumann['top']:
            out[n, :, -1] = T
        if not neumann['bottom']:
            out[n, :, 0] = B
        if not neumann['left']:
            out[n, 0, :] = L
        if not neumann['right']:
            out[n, -1, :] = R
        out[n, i0:i1 + 1, j0:j1 + 1] = u
    return out
