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


@pytest.mark.parametrize('solver', ['gemm', 'fft'])
@pytest.mark.parametrize('H,W', [(9, 7), (33, 47), (64, 64), (130, 101)])
def test_dst_solve_matches_sparse_direct_solve(H, W, solver):
    """Both routes of the Dirichlet solve - the DST-I as fp64 matrix-core GEMMs and as rocFFT transforms of the odd extension (BASELINE.json
    north star) - against the sparse direct solve of the reference's linear system."""
    from poisson_cnn_amd.dataset import _kernels as K
    rng = np.random.default_rng(H)
    N = 3
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    rhs = f32(rng.standard_normal((N, H, W)))
    bc = {'left': f32(rng.standard_normal((N, W))), 'right': f32(rng.standard_normal((N, W))),
          'bottom': f32(rng.standard_normal((N, H))), 'top': f32(rng.standard_normal((N, H)))}
    dx = f32(rng.uniform(5e-3, 5e-2, N))
    ref = ods.multigrid_poisson_solve(rhs, bc, dx)
    got = K.fd_poisson_dst(dev(rhs), dev(bc['left']), dev(bc['right']), dev(bc['bottom']), dev(bc['top']), dev(dx), solver=solver).cpu().numpy()
    assert rel(got, ref) < 2e-7            # fp64 solve, result rounded to fp32 once
    # discrete residual of the fp64 reference is ~0; ours is limited by the fp32 output rounding only
    assert np.array_equal(got[:, 0, :], bc['left'].astype(np.float32)) and np.array_equal(got[:, -1, :], bc['right'].astype(np.float32))
    assert np.array_equal(got[:, 1:-1, 0], bc['bottom'].astype(np.float32)[:, 1:-1]) and np.array_equal(got[:, 1:-1, -1], bc['top'].astype(np.float32)[:, 1:-1])


@pytest.mark.parametrize('H,W,N', [(512, 512, 4), (1024, 1024, 2), (1025, 700, 1)])
def test_rocfft_route_agrees_with_the_gemm_route_at_full_size(H, W, N):
    """At BASELINE.json's grid sizes the GEMM-DST is the checker of the rocFFT route (the sparse direct solve takes minutes there): the two
    fp64 pipelines must agree to the fp32 rounding of the stored solution, and the 5-point residual of the rocFFT solution must be at the
    level that rounding allows.  solver='auto' picks rocFFT from K.FFT_FROM (2048, the measured crossover) points per axis."""
    from poisson_cnn_amd.dataset import _kernels as K
    g = torch.Generator(device='cuda').manual_seed(H + W)
    rhs = torch.randn(N, H, W, device='cuda', generator=g)
    left, right = torch.randn(N, W, device='cuda', generator=g), torch.randn(N, W, device='cuda', generator=g)
    bottom, top = torch.randn(N, H, device='cuda', generator=g), torch.randn(N, H, device='cuda', generator=g)
    dx = torch.rand(N, device='cuda', generator=g) * 4.5e-2 + 5e-3
    a = K.fd_poisson_dst(rhs, left, right, bottom, top, dx, solver='gemm')
    b = K.fd_poisson_dst(rhs, left, right, bottom, top, dx, solver='fft')
    assert float((a.double() - b.double()).norm() / a.double().norm()) < 1e-7 and float((a - b).abs().max() / a.abs().max()) < 3e-7
    c = K.fd_poisson_dst(rhs, left, right, bottom, top, dx, solver='auto')
    assert torch.equal(c, b if min(H, W) >= K.FFT_FROM else a)
    u, f, h = b.double(), rhs.double(), dx.double().reshape(-1, 1, 1)
    lap = (u[:, 2:, 1:-1] + u[:, :-2, 1:-1] + u[:, 1:-1, 2:] + u[:, 1:-1, :-2] - 4 * u[:, 1:-1, 1:-1]) / h ** 2
    res = (lap - f[:, 1:-1, 1:-1]).abs().amax(dim=(1, 2))
    bound = 8 * 2.0 ** -24 * u.abs().amax(dim=(1, 2)) / h.reshape(-1) ** 2 + 1e-4 * f.abs().amax(dim=(1, 2))
    assert bool((res <= bound).all()), (res, bound)


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


@pytest.mark.parametrize('neumann', [(True, True, True, True), (False, True, True, False), (True, False, False, False), (False, False, False, False),
                                     (True, True, False, False)])
def test_mixed_bc_fd_solve_matches_sparse_direct(neumann):
    """discrete Neumann / mixed-BC solve on the fp64 GEMM path vs the scipy sparse direct solve of the same system
    (Lagrange-multiplier closure of the singular all-Neumann case, as Navier_Stokes_2D/solvers.py:258-259)."""
    from oracle import dataset as ods
    from poisson_cnn_amd.dataset import mixed_bc_poisson_solve
    rng = np.random.default_rng(sum(int(b) << i for i, b in enumerate(neumann)))
    N, H, W = 3, 45, 38
    rhs = rng.standard_normal((N, H, W))
    bc = {k: rng.standard_normal((N, W if k in ('left', 'right') else H)) for k in ('left', 'right', 'bottom', 'top')}
    dx = rng.uniform(5e-3, 5e-2, N)
    rhs, dx = rhs.astype(np.float32).astype(np.float64), dx.astype(np.float32).astype(np.float64)
    bc = {k: v.astype(np.float32).astype(np.float64) for k, v in bc.items()}
    types = dict(zip(('left', 'right', 'bottom', 'top'), ['neumann' if b else 'dirichlet' for b in neumann]))
    ref = ods.mixed_bc_poisson_solve(rhs, bc, dx, dict(zip(('left', 'right', 'bottom', 'top'), neumann)))
    got = mixed_bc_poisson_solve(rhs, bc, dx, types).cpu().numpy()[:, 0]
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-7
    if not any(neumann):       # all Dirichlet: the same numbers as the DST-I solver of the reference's system
        from poisson_cnn_amd.dataset import multigrid_poisson_solve
        assert np.abs(multigrid_poisson_solve(rhs, bc, dx).cpu().numpy()[:, 0] - got).max() < 1e-6 * np.abs(got).max()


def test_mixed_bc_residual_at_1024_and_generator():
    """5-point residual of the all-Neumann and of a mixed solve at 1024 x 1024 (the data are made compatible for the singular case), and the
    numerical generator with boundary_types producing a mixed-BC batch."""
    from poisson_cnn_amd.dataset import mixed_bc_poisson_solve, numerical_dataset_generator
    g = torch.Generator(device='cuda').manual_seed(3)
    N, H, W = 2, 1024, 1024
    t = torch.linspace(0, 1, H, device='cuda', dtype=torch.float64)
    f = (torch.cos(3 * np.pi * t)[None, :, None] * torch.cos(2 * np.pi * t)[None, None, :]).expand(N, H, W).contiguous().float()   # zero trapezoid mean
    z = torch.zeros(N, H, device='cuda')
    dx = torch.tensor([0.01, 0.02], device='cuda')
    for types, zero_mean in ((dict(left='neumann', right='neumann', bottom='neumann', top='neumann'), True),
                             (dict(left='dirichlet', right='neumann', bottom='neumann', top='dirichlet'), False)):
        u = mixed_bc_poisson_solve(f, dict(left=z, right=z, bottom=z, top=z), dx, types)[:, 0].double()
        h2 = dx.double()[:, None, None] ** 2
        up = torch.nn.functional.pad(u[:, None], (1, 1, 1, 1), mode='reflect')[:, 0]        # mirrored ghost nodes = homogeneous Neumann closure
        lap = (up[:, 2:, 1:-1] + up[:, :-2, 1:-1] + up[:, 1:-1, 2:] + up[:, 1:-1, :-2] - 4 * u) / h2
        res = (lap - f.double()).abs()
        i0, i1 = (0 if types['left'] == 'neumann' else 1), (H if types['right'] == 'neumann' else H - 1)
        j0, j1 = (0 if types['bottom'] == 'neumann' else 1), (W if types['top'] == 'neumann' else W - 1)
        bound = 16 * 2.0 ** -24 * u.abs().amax(dim=(1, 2)) / dx.double() ** 2 + 1e-9
        assert bool((res[:, i0:i1, j0:j1].amax(dim=(1, 2)) <= bound).all()), (res[:, i0:i1, j0:j1].amax(dim=(1, 2)), bound)
        if zero_mean:
            w = torch.ones(H, device='cuda', dtype=torch.float64); w[0] = w[-1] = 0.5
            assert float((u * w[None, :, None] * w[None, None, :]).sum(dim=(1, 2)).abs().max()) < 1e-6 * float(u.abs().sum(dim=(1, 2)).max())
    gen = numerical_dataset_generator(batch_size=4, batches_per_epoch=1, rhs_smoothness=5, boundary_smoothness=5, seed=1, output_shape=[96, 80], return_rhs=True,
                                      return_boundaries=True, return_dx=True, boundary_types=dict(left='neumann', top='neumann'))
    inp, soln = gen[0]
    assert tuple(soln.shape) == (4, 1, 96, 80) and torch.isfinite(soln).all()
    rhs, left, top, right, bottom, dxs = inp
    assert torch.equal(soln[:, 0, -1, :], right[:, 0]) and torch.equal(soln[:, 0, :-1, 0], bottom[:, 0, :-1])    # Dirichlet edges carry their values (the corner takes the right edge's)
    # Neumann edge: second-order one-sided check of du/dn on the left edge (outward normal = -i): (u[1] - u[-1ghost]) ... use the discrete closure itself
    u = soln[:, 0].double(); h = dxs.double().reshape(-1, 1)
    ghost = u[:, 1, 1:-1] + 2 * h * left[:, 0, 1:-1].double()
    lap0 = (ghost + u[:, 1, 1:-1] + u[:, 0, 2:] + u[:, 0, :-2] - 4 * u[:, 0, 1:-1]) / h ** 2
    assert float((lap0 - rhs[:, 0, 0, 1:-1].double()).abs().max()) < 1e-3 * float(rhs.abs().max()) + 32 * 2.0 ** -24 * float(u.abs().max() / h.min() ** 2)
