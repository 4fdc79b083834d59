"""GPU parity of the on-device dataset generators against the fp64 dataset oracle."""
import numpy as np
import pytest
import torch

from oracle import dataset as ods, np_ops

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device='cuda')


@pytest.mark.parametrize('H,W', [(9, 7), (33, 47), (64, 64), (130, 101)])
def test_dst_solve_matches_sparse_direct_solve(H, W):
    from poisson_cnn_amd.dataset import _kernels as K
    rng = np.random.default_rng(H)
    N = 3
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    rhs = f32(rng.standard_normal((N, H, W)))
    bc = {'left': f32(rng.standard_normal((N, W))), 'right': f32(rng.standard_normal((N, W))),
          'bottom': f32(rng.standard_normal((N, H))), 'top': f32(rng.standard_normal((N, H)))}
    dx = f32(rng.uniform(5e-3, 5e-2, N))
    ref = ods.multigrid_poisson_solve(rhs, bc, dx)
    got = K.fd_poisson_dst(dev(rhs), dev(bc['left']), dev(bc['right']), dev(bc['bottom']), dev(bc['top']), dev(dx)).cpu().numpy()
    assert rel(got, ref) < 2e-7            # fp64 solve, result rounded to fp32 once
    # discrete residual of the fp64 reference is ~0; ours is limited by the fp32 output rounding only
    assert np.array_equal(got[:, 0, :], bc['left'].astype(np.float32)) and np.array_equal(got[:, -1, :], bc['right'].astype(np.float32))
    assert np.array_equal(got[:, 1:-1, 0], bc['bottom'].astype(np.float32)[:, 1:-1]) and np.array_equal(got[:, 1:-1, -1], bc['top'].astype(np.float32)[:, 1:-1])


def test_gemm_f64_and_dst_orthonormality():
    from poisson_cnn_amd.dataset import _kernels as K
    S, lam = K.dst_matrices(70, torch.device('cuda'))
    Sn = S.cpu().numpy()
    assert np.allclose(Sn @ Sn, np.eye(68), atol=1e-13) and np.allclose(Sn, Sn.T)
    assert np.allclose(lam.cpu().numpy(), 2 - 2 * np.cos(np.pi * np.arange(1, 69) / 69))
    # A = S diag(lam) S reproduces the 1-D second-difference matrix
    T = Sn @ np.diag(lam.cpu().numpy()) @ Sn
    ref = 2 * np.eye(68) - np.eye(68, k=1) - np.eye(68, k=-1)
    assert np.allclose(T, ref, atol=1e-12)


def test_series_synthesis_and_helpers():
    from poisson_cnn_amd.dataset import _kernels as K
    rng = np.random.default_rng(1)
    N, H, W, ka, kb = 3, 45, 70, 5, 8
    c = rng.uniform(-1, 1, (N, ka, kb)).astype(np.float32).astype(np.float64)
    for trig, kw in ((0, 'sin_coeff'), (1, 'cos_coeff')):
        got = K.series_synthesis(dev(c), H, W, trig).cpu().numpy()
        ref = np.stack([ods.generate_smooth_function((H, W), **{kw: c[n]}) for n in range(N)])
        assert rel(got, ref) < 2e-6
    acc = K.series_synthesis(dev(c), H, W, 0)
    K.series_synthesis(dev(c), H, W, 1, out=acc, accumulate=True)
    ref = np.stack([ods.generate_smooth_function((H, W), c[n], c[n]) for n in range(N)])
    assert rel(acc.cpu().numpy(), ref) < 2e-6
    U, V = rng.standard_normal((N, 2, H)), rng.standard_normal((N, 2, W))
    assert rel(K.separable_sum(dev(U), dev(V)).cpu().numpy(), np.einsum('nra,nrb->nab', U, V)) < 2e-6
    x = dev(ref)
    f = K.set_max_magnitude(x, dev(np.array([1.0, 2.0, 0.5]))).cpu().numpy()
    assert np.allclose(np.abs(x.cpu().numpy()).max((1, 2)), [1.0, 2.0, 0.5], rtol=1e-6)
    assert np.allclose(f, np.array([1.0, 2.0, 0.5]) / np.abs(ref).max((1, 2)), rtol=1e-5)
    assert np.allclose(K.max_abs_per_sample(x).cpu().numpy(), [1.0, 2.0, 0.5], rtol=1e-6)
    ctrl = rng.uniform(-1, 1, (2, 1, 6, 9))
    got = K.resize_legacy_bicubic(dev(ctrl.transpose(0, 2, 3, 1)), (50, 61)).cpu().numpy()[..., 0]
    assert rel(got, ods.image_resize_legacy_bicubic(ctrl, (50, 61))[:, 0]) < 2e-6


def test_polynomial_tables_match_independent_form():
    from poisson_cnn_amd.dataset import _poly_and_second_derivative
    rng = np.random.default_rng(3)
    x = np.linspace(0, 1, 50)
    for d in (2, 3, 5, 6):
        roots = -rng.uniform(size=d)
        p, ddp = _poly_and_second_derivative(roots, x)
        rp, rddp = ods.polynomial_and_second_derivative(roots, x)
        assert np.allclose(p, rp, atol=1e-12) and np.allclose(ddp, rddp, atol=1e-10)


@pytest.mark.parametrize('homogeneous', [True, False])
def test_reverse_generator_pairs_are_consistent(homogeneous):
    """Like the reference's own consistency check (dataset/generators/reverse.py:332-357): the FD Laplacian of the generated
    solution reproduces the generated RHS (up to O(dx^2) truncation) once the normalisations are undone."""
    from poisson_cnn_amd.dataset import reverse_poisson_dataset_generator
    from poisson_cnn_amd import configs
    cfg = dict(configs.hpnn()['dataset'])
    cfg.update(batch_size=4, homogeneous_bc=homogeneous, return_boundaries=not homogeneous, normalizations=None,
               random_output_shape_range=[[220, 260], [220, 260]], fourier_coeff_grid_size_range=[[1, 4], [1, 4]])
    gen = reverse_poisson_dataset_generator(seed=5, **cfg)
    assert len(gen) == cfg['batches_per_epoch']
    inp, soln = gen[0]
    rhs, dx = inp[0].cpu().numpy().astype(np.float64), inp[-1].cpu().numpy().astype(np.float64)
    u = soln.cpu().numpy().astype(np.float64)
    N, _, H, W = u.shape
    assert rhs.shape == u.shape and dx.shape == (4, 1)
    # note: the reference's series use x = linspace(0, pi, n) but L = dx*n for the RHS coefficients (reverse.py:203), so the
    # pair satisfies the PDE on the grid spacing dx*n/(n-1); test with each axis' effective spacing
    for n in range(N):
        hy, hx = dx[n, 0] * H / (H - 1), dx[n, 0] * W / (W - 1)
        lap = (u[n, 0, 2:, 1:-1] - 2 * u[n, 0, 1:-1, 1:-1] + u[n, 0, :-2, 1:-1]) / hy ** 2 + (u[n, 0, 1:-1, 2:] - 2 * u[n, 0, 1:-1, 1:-1] + u[n, 0, 1:-1, :-2]) / hx ** 2
        r = rhs[n, 0, 1:-1, 1:-1]
        assert np.linalg.norm(lap - r) / np.linalg.norm(r) < 5e-2
    if homogeneous:
        assert np.abs(u[:, :, 0, :]).max() < 1e-5 * np.abs(u).max() and np.abs(u[:, :, :, -1]).max() < 1e-5 * np.abs(u).max()
    else:
        assert len(inp) == 6 and np.array_equal(inp[1].cpu().numpy(), soln[:, :, 0, :].cpu().numpy())


def test_reverse_generator_normalizations_and_neumann():
    from poisson_cnn_amd.dataset import reverse_poisson_dataset_generator, reverse_poisson_dataset_generator_homogeneous_neumann
    from poisson_cnn_amd import configs
    cfg = dict(configs.hpnn()['dataset'])
    cfg.update(batch_size=3, random_output_shape_range=[[120, 160], [120, 160]])
    inp, soln = reverse_poisson_dataset_generator(seed=1, **cfg)[0]
    assert len(inp) == 2 and np.allclose(inp[0].abs().amax(dim=(1, 2, 3)).cpu().numpy(), 1.0, rtol=1e-6)
    ncfg = dict(configs.hpnn_neumann()['dataset'])
    ncfg.update(batch_size=3, random_output_shape_range=[[120, 160], [120, 160]])
    g = reverse_poisson_dataset_generator_homogeneous_neumann(seed=2, **ncfg)
    g.fixed_output_shape = (128, 144)
    inp, soln = g[0]
    u = soln.cpu().numpy().astype(np.float64)
    assert u.shape == (3, 1, 128, 144)
    # cosine series: zero normal derivative (one-sided difference is O(dx^2) small), zero-mean RHS up to quadrature error
    # (even extension => u' = u''' = 0 at the wall, so the 2nd-order one-sided derivative stencil vanishes to O((k h)^4))
    assert np.abs(-3 * u[:, :, 0, :] + 4 * u[:, :, 1, :] - u[:, :, 2, :]).max() < 5e-3 * np.abs(u).max()
    assert np.abs(-3 * u[:, :, :, -1] + 4 * u[:, :, :, -2] - u[:, :, :, -3]).max() < 5e-3 * np.abs(u).max()


def test_numerical_generator_solves_the_fd_system():
    from poisson_cnn_amd.dataset import numerical_dataset_generator
    gen = numerical_dataset_generator(batch_size=3, batches_per_epoch=2, randomize_rhs_smoothness=True, randomize_boundary_smoothness=True, seed=3,
                                      return_rhs=True, return_boundaries=True, return_dx=True, random_output_shape_range=[[60, 90], [60, 90]])
    inp, soln = gen[0]
    assert len(inp) == 6                      # [rhs, left, top, right, bottom, dx]
    rhs, left, top, right, bottom, dx = [t.cpu().numpy().astype(np.float64) for t in inp]
    u = soln.cpu().numpy().astype(np.float64)
    N, _, H, W = u.shape
    assert left.shape == (N, 1, W) and top.shape == (N, 1, H) and dx.shape == (N, 1)
    assert np.allclose(np.abs(rhs).max((1, 2, 3)), 1.0, rtol=1e-6) and np.allclose(np.abs(left).max((1, 2)), 1.0, rtol=1e-6)
    ref = ods.multigrid_poisson_solve(rhs[:, 0], {'left': left[:, 0], 'right': right[:, 0], 'top': top[:, 0], 'bottom': bottom[:, 0]}, dx[:, 0])
    assert rel(u[:, 0], ref) < 2e-7
    # and the reference solution satisfies the 5-point equations to fp64 round-off (oracle self-check)
    for n in range(N):
        res = ods.five_point_laplacian(ref[n], dx[n, 0]) - rhs[n, 0, 1:-1, 1:-1]
        assert np.abs(res).max() < 1e-7 * np.abs(rhs[n]).max() / dx[n, 0] ** 2 * dx[n, 0] ** 2 + 1e-6


def test_multigrid_poisson_solve_entry_point():
    from poisson_cnn_amd.dataset import multigrid_poisson_solve, cholesky_poisson_solve
    rng = np.random.default_rng(2)
    N, H, W = 2, 40, 31
    rhs = rng.standard_normal((N, 1, H, W)).astype(np.float32)
    bc = {'left': rng.standard_normal((N, 1, W)).astype(np.float32), 'right': rng.standard_normal((N, 1, W)).astype(np.float32),
          'bottom': rng.standard_normal((N, 1, H)).astype(np.float32), 'top': rng.standard_normal((N, 1, H)).astype(np.float32)}
    dx = rng.uniform(0.01, 0.05, (N, 1)).astype(np.float32)
    ref = ods.multigrid_poisson_solve(rhs[:, 0].astype(np.float64), {k: v[:, 0].astype(np.float64) for k, v in bc.items()}, dx[:, 0].astype(np.float64))
    got = multigrid_poisson_solve(rhs, bc, dx, tol=1e-10).cpu().numpy()
    assert got.shape == (N, 1, H, W) and rel(got[:, 0], ref) < 2e-7
    assert rel(cholesky_poisson_solve(rhs, bc, dx).cpu().numpy()[:, 0], ref) < 2e-7
