"""On-device reference-solution generators: drop-ins for poisson_CNN/dataset/generators (reference paths relative to
poisson_CNN/).  Same constructor kwargs (the "dataset" section of experiments/hpnn*.json loads unchanged) and the Keras
Sequence protocol: `len(gen)`, `gen[idx] -> ([rhs, (boundaries...), dx], soln)`, every tensor a float32 CUDA tensor in the
reference's channels_first layout.

  numerical_dataset_generator                              <- dataset/generators/numerical.py:152-216 (+ numerical_dataset :74-150)
  reverse_poisson_dataset_generator                        <- dataset/generators/reverse.py:101-330
  reverse_poisson_dataset_generator_homogeneous_neumann    <- dataset/generators/reverse_neumann.py:9-66

Only the O(10)-sized random parameters (shapes, spacings, coefficient grids, polynomial roots, control points) are drawn on
the host (numpy Generator seeded with `seed`; the reference's TF RNG stream cannot be reproduced, so parity is
distributional for the random draws and exact for everything computed from them); every H x W field is synthesised,
solved and normalised on the GPU.
"""
import math

import numpy as np
import torch

from . import _kernels as K
from .. import ops


# ----------------------------------------------------------------------------- host-side samplers (dataset/utils)
def _integrate_piecewise_sigmoid(L1, L2):
    """dataset/utils/generate_uniformly_distributed_aspect_ratios.py:4-23."""
    pre = np.maximum((L2[..., 0] - L1[..., 0]) * L2[..., 0], 0.0)
    post = np.maximum((L1[..., 1] - L2[..., 1]) * L2[..., 1], 0.0)
    lo = np.maximum(L2[..., 0], L1[..., 0])
    hi = np.minimum(L2[..., 1], L1[..., 1])
    return pre + post + 0.5 * (hi ** 2 - lo ** 2)


def generate_uniformly_distributed_aspect_ratios(output_shape_range, dx_range=None, samples=1, rng=None):
    """dataset/utils/generate_uniformly_distributed_aspect_ratios.py:59-85 -> (samples, ndims-1)."""
    rng = rng or np.random.default_rng()
    osr = np.asarray(output_shape_range, dtype=np.float64)
    if dx_range is None:
        dsr = osr - 1.0
    else:
        dxr = np.asarray(dx_range, dtype=np.float64)
        dsr = np.stack([(osr[:, 0] - 1) * dxr[:, 0], (osr[:, 1] - 1) * dxr[:, 1]], -1)
    max_ar = dsr[0, 1] / dsr[1:, 0]
    min_ar = dsr[0, 0] / dsr[1:, 1]
    L1 = np.tile(dsr[0][None], (dsr.shape[0] - 1, 1))
    L2 = dsr[1:]
    prop = (L2[:, 1] * (L1[:, 1] - L1[:, 0]) - _integrate_piecewise_sigmoid(L1, L2)) / ((L2[:, 1] - L2[:, 0]) * (L1[:, 1] - L1[:, 0]))
    under = (rng.uniform(size=(samples, prop.shape[0])) < prop[None]).astype(np.float64)
    ub = min(float(np.min(max_ar)), 1.0)
    lb = max(float(np.max(min_ar)), 1.0)
    u = rng.uniform(size=under.shape)
    return under * ((ub - min_ar) * u + min_ar) + (1 - under) * ((max_ar - lb) * u + lb)


def generate_output_shapes_and_grid_spacings_from_aspect_ratios(aspect_ratios, random_output_shape_range, random_dx_range, constant_dx=False,
                                                                samples=None, rng=None):
    """dataset/utils/generate_output_shapes_and_grid_spacings_from_aspect_ratios.py:4-41 -> (npts (ndims,), dx (samples, ndims))."""
    rng = rng or np.random.default_rng()
    ar = np.asarray(aspect_ratios, dtype=np.float64)
    ndims = ar.shape[1] + 1
    osr = np.asarray(random_output_shape_range, dtype=np.int64)
    dxr = np.asarray(random_dx_range, dtype=np.float64)
    nx_min, nx_max = osr[0, 0] - 0.4999999999, osr[0, 1] + 0.4999999999
    nx = int(np.round(rng.uniform() * (nx_max - nx_min) + osr[0, 0]))
    if constant_dx:
        dx = np.tile(rng.uniform(size=(samples, 1)) * (dxr[0, 1] - dxr[0, 0]) + dxr[0, 0], (1, ndims))
        other = (nx / ar[0]).astype(np.int64)
    else:
        dx0 = (dxr[0, 1] - dxr[0, 0]) * rng.uniform(size=(ar.shape[0], 1)) + dxr[0, 0]
        Lx = (nx - 1) * dx0[:, 0]
        L = np.concatenate([np.ones((ar.shape[0], 1)), ar], -1) * Lx[:, None]
        other = (rng.uniform(size=(ndims - 1,)) * (osr[1:, 1] - osr[1:, 0])).astype(np.int64) + osr[1:, 0]
        dx = np.concatenate([dx0, L[:, 1:] / (other - 1)[None]], 1)
    npts = np.concatenate([[nx], other]).astype(np.float64)
    maxpts, minpts = osr.max(1), osr.min(1)
    over = max(1.0, float(np.max(npts / maxpts)))
    under = min(1.0, float(np.min(npts / minpts)))
    scale = over if over > 1.0 else max(under, float(np.max(npts / maxpts)))
    return (npts / scale).astype(np.int64), dx


def _range2(value_range, ndims):
    """handle_grid_parameters_range (dataset/generators/reverse.py:10-21)."""
    v = np.asarray(value_range, dtype=np.float64)
    if v.ndim == 1:
        assert v[0] <= v[1], 'Upper bound must be larger than or equal to the lower bound!'
        v = np.tile(v[None], (ndims, 1))
    assert v.shape == (ndims, 2) and np.all(v[:, 1] >= v[:, 0]) and np.all(v[:, 0] >= 0)
    return v


def _process_normalizations(n):
    """dataset/generators/reverse.py:23-36."""
    out = {'rhs_max_magnitude': False, 'max_domain_size_squared': False, 'soln_max_magnitude': False}
    if isinstance(n, dict):
        out.update(n)
        if isinstance(out['rhs_max_magnitude'], bool) and out['rhs_max_magnitude']:
            out['rhs_max_magnitude'] = 1.0
    return out


def _poly_and_second_derivative(roots, x):
    """p(x) = prod_r (x + roots[r]) and d2p/dx2 on the points x.  The reference differentiates the product twice with tf.gradients and
    patches the NaNs that produces wherever a factor vanishes (dataset/generators/reverse.py:39-71); here the product is expanded once
    (degree <= ~12, roots in [-1, 0]: exact to ~1e-13 in float64) and both are Horner evaluations - no patch, no per-pair products."""
    from numpy.polynomial import polynomial as npoly
    c = npoly.polyfromroots(-np.asarray(roots, dtype=np.float64))
    return npoly.polyval(x, c), (npoly.polyval(x, npoly.polyder(c, 2)) if c.shape[0] > 2 else np.zeros_like(x))


def _streams(seed, shard):
    """The generators' two host-side random streams.  `shard` = (rank, world_size) of a data-parallel run: the SHAPE stream - everything
    that decides the grid size (H, W) of a step, drawn for the global batch as the reference's single generator does
    (dataset/generators/reverse.py:192-193: one shape per batch, which MirroredStrategy then splits) - is seeded identically on every rank;
    the DATA stream (coefficients, control points, magnitudes) is seeded `seed + rank`.  Without a shard both are ONE stream (single process)."""
    if shard is None:
        rng = np.random.default_rng(seed)
        return rng, rng, (0, 1)
    rank, world = int(shard[0]), int(shard[1])
    if not 0 <= rank < world:
        raise ValueError('shard = (rank, world_size) with 0 <= rank < world_size, got %r' % (shard,))
    return np.random.default_rng(seed + rank), np.random.default_rng([int(seed), 0x5AFE]), (rank, world)


# ----------------------------------------------------------------------------- analytic ("reverse") generators
class reverse_poisson_dataset_generator:
    def __init__(self, batch_size, batches_per_epoch, random_output_shape_range, fourier_coeff_grid_size_range, taylor_degree_range=None,
                 grid_spacings_range=None, ndims=None, homogeneous_bc=False, return_rhses=True, return_boundaries=True, return_dx=True,
                 normalizations=None, uniform_grid_spacing=False, seed=0, device=None, shard=None):
        self.batch_size, self.batches_per_epoch = int(batch_size), int(batches_per_epoch)
        self.ndims = 2 if ndims is None else ndims
        if self.ndims != 2:
            raise NotImplementedError('2-D only')
        self.homogeneous_bc = homogeneous_bc
        self.grid_spacings_range = _range2(grid_spacings_range, 2)
        self.random_output_shape_range = _range2(random_output_shape_range, 2).astype(np.int64)
        self.fourier_coeff_grid_size_range = _range2(fourier_coeff_grid_size_range, 2)
        self.taylor_degree_range = _range2(taylor_degree_range, 2) if taylor_degree_range is not None else None
        self.return_rhses, self.return_boundaries, self.return_dx = return_rhses, return_boundaries, return_dx
        self.normalizations = _process_normalizations(normalizations)
        self.uniform_grid_spacing = uniform_grid_spacing
        self.rng, self.shape_rng, self.shard = _streams(seed, shard)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.neumann = False
        self.fixed_output_shape = None     # set to (H, W) to pin the grid (benchmarks / data-parallel ranks sharing one shape)

    def __len__(self):
        return self.batches_per_epoch

    # -- host-side draws
    def _grid_sizes(self, rng_range):
        """generate_grid_sizes (reverse.py:164-171)."""
        return ((rng_range[:, 1] - rng_range[:, 0]) * self.rng.uniform(size=(2,)) + rng_range[:, 0] + 1).astype(np.int64)

    def _shape_and_spacings(self):
        """generate_grid_sizes_and_spacings_with_uniform_AR (reverse.py:173-177)."""
        # one draw for the GLOBAL batch from the stream every rank shares (one grid shape per step on all replicas, reverse.py:192-193);
        # a rank keeps its own rows of the per-sample spacings
        rank, world = self.shard
        gb = self.batch_size * world
        ar = generate_uniformly_distributed_aspect_ratios(self.random_output_shape_range, None if self.uniform_grid_spacing else self.grid_spacings_range,
                                                          gb, self.shape_rng)
        shape, dx = generate_output_shapes_and_grid_spacings_from_aspect_ratios(ar, self.random_output_shape_range, self.grid_spacings_range,
                                                                                constant_dx=self.uniform_grid_spacing, samples=gb, rng=self.shape_rng)
        dx = dx[rank * self.batch_size:(rank + 1) * self.batch_size]
        if self.fixed_output_shape is not None:
            shape = np.asarray(self.fixed_output_shape, dtype=np.int64)
        return shape, dx

    def _dev(self, a):
        """host array -> float32 device tensor through a pinned staging buffer and an asynchronous copy on the current stream: a pageable upload
        (torch.tensor(..., device=...)) synchronises the stream, ten times per batch (profiles/r06_train_shipped.txt); torch's pinned-memory cache
        keeps the staging buffer alive until the copy has run."""
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        if self.device.type != 'cuda':
            return t.to(self.device)
        return t.pin_memory().to(self.device, non_blocking=True)

    def _taylor_tables(self, H, W, domain_sizes):
        """1-D tables of generate_soln_and_rhs_taylor (reverse.py:231-256): X, X'', Y, Y'' per sample."""
        N = self.batch_size
        degs = self._grid_sizes(self.taylor_degree_range)
        tabs = []
        for k, n in enumerate((H, W)):
            pd = int(degs[k]) - 1                                     # polynomials_and_their_2nd_derivatives: poly_deg -= 1
            coeffs = 2 * self.rng.uniform(size=(N, pd)) - 1
            if self.homogeneous_bc:
                roots = -self.rng.uniform(size=(N, pd, max(pd - 1, 0)))
                roots = np.concatenate([np.zeros((N, pd, 1)), -np.ones((N, pd, 1)), roots], -1)
            else:
                roots = -self.rng.uniform(size=(N, pd, pd + 2))
            x = np.linspace(0.0, 1.0, n)
            P, D = np.zeros((N, n)), np.zeros((N, n))
            for b in range(N):
                for i in range(pd):
                    p, ddp = _poly_and_second_derivative(roots[b, i, :i + 2], x)
                    P[b] += coeffs[b, i] * p
                    D[b] += coeffs[b, i] * ddp / domain_sizes[b, k] ** 2
            tabs.append((P, D))
        return tabs

    def __getitem__(self, idx=0):
        N = self.batch_size
        ncoef = np.stack([self._grid_sizes(self.fourier_coeff_grid_size_range) for _ in range(N)])       # (N,2)
        shape, dx = self._shape_and_spacings()
        H, W = int(shape[0]), int(shape[1])
        ka, kb = int(ncoef[:, 0].max()), int(ncoef[:, 1].max())
        coef = np.zeros((N, ka, kb))
        for b in range(N):
            coef[b, :ncoef[b, 0], :ncoef[b, 1]] = 2 * self.rng.uniform(size=(ncoef[b, 0], ncoef[b, 1])) - 1
        domain_sizes = dx * np.array([H, W], dtype=np.float64)[None]     # L = dx * n (reverse.py:203), not dx*(n-1)
        A, B = np.arange(1, ka + 1, dtype=np.float64), np.arange(1, kb + 1, dtype=np.float64)
        adj = -((math.pi * A[None, :, None] / domain_sizes[:, 0, None, None]) ** 2 + (math.pi * B[None, None, :] / domain_sizes[:, 1, None, None]) ** 2)
        trig = 1 if self.neumann else 0
        if not self.neumann and not self.homogeneous_bc:
            cosc = np.zeros((N, ka, kb))
            for b in range(N):
                cosc[b, :ncoef[b, 0], :ncoef[b, 1]] = 2 * self.rng.uniform(size=(ncoef[b, 0], ncoef[b, 1])) - 1
        soln = K.series_synthesis(self._dev(coef), H, W, trig)
        rhs = K.series_synthesis(self._dev(coef * adj), H, W, trig)
        if not self.neumann and not self.homogeneous_bc:
            K.series_synthesis(self._dev(cosc), H, W, 1, out=soln, accumulate=True)
            K.series_synthesis(self._dev(cosc * adj), H, W, 1, out=rhs, accumulate=True)
        if not self.neumann and self.taylor_degree_range is not None:
            (X, Xdd), (Y, Ydd) = self._taylor_tables(H, W, domain_sizes)
            soln_t = K.separable_sum(self._dev(X[:, None]), self._dev(Y[:, None]))
            rhs_t = K.separable_sum(self._dev(np.stack([Xdd, X], 1)), self._dev(np.stack([Y, Ydd], 1)))
            s = K.max_abs_per_sample(rhs) / K.max_abs_per_sample(rhs_t)      # reverse.py:299-306
            K.scale_samples(rhs_t, s)
            K.scale_samples(soln_t, s)
            ops.axpby(1.0, rhs_t.view(N, H, W, 1), 1.0, rhs.view(N, H, W, 1))
            ops.axpby(1.0, soln_t.view(N, H, W, 1), 1.0, soln.view(N, H, W, 1))
        nz = self.normalizations
        if nz['rhs_max_magnitude'] is not False:                             # reverse.py:287-290
            f = K.set_max_magnitude(rhs, torch.full((N,), float(nz['rhs_max_magnitude']), device=self.device))
            K.scale_samples(soln, f)
        if nz['soln_max_magnitude'] is not False:
            K.set_max_magnitude(soln, torch.ones((N,), device=self.device))
        if nz['max_domain_size_squared']:
            K.scale_samples(soln, self._dev(1.0 / domain_sizes.max(1) ** 2))
        rhs, soln = rhs.view(N, 1, H, W), soln.view(N, 1, H, W)
        problem = []
        if self.return_rhses:
            problem.append(rhs)
        if self.return_boundaries:
            problem += [soln[:, :, 0, :], soln[:, :, -1, :], soln[:, :, :, 0], soln[:, :, :, -1]]      # _boundary_slices order (reverse.py:141-146)
        if self.return_dx:
            problem.append(self._dev(dx[:, :1] if self.uniform_grid_spacing else dx))
        return problem, soln


class reverse_poisson_dataset_generator_homogeneous_neumann(reverse_poisson_dataset_generator):
    """Cosine-only series: zero normal derivative on every edge, zero-mean RHS (dataset/generators/reverse_neumann.py:9-66)."""

    def __init__(self, batch_size, batches_per_epoch, random_output_shape_range, fourier_coeff_grid_size_range, grid_spacings_range=None, ndims=None,
                 return_rhses=True, return_dx=True, normalizations=None, uniform_grid_spacing=False, seed=0, device=None, shard=None):
        super().__init__(batch_size, batches_per_epoch, random_output_shape_range, fourier_coeff_grid_size_range, None, grid_spacings_range, ndims,
                         homogeneous_bc=False, return_rhses=return_rhses, return_boundaries=False, return_dx=return_dx, normalizations=normalizations,
                         uniform_grid_spacing=uniform_grid_spacing, seed=seed, device=device, shard=shard)
        self.neumann = True


# ----------------------------------------------------------------------------- numerical (FD) generator
_BOUNDARY_KEYS = ('left', 'top', 'right', 'bottom')


class numerical_dataset_generator:
    """Random smooth RHS / Dirichlet BCs (bicubic up-sampling of random control points) -> 5-point FD solve on the GPU (DST-I).
    Output order with return_keras_style: [rhs, left, top, right, bottom, dx] (dataset/generators/numerical.py:204-216)."""

    def __init__(self, batch_size=1, batches_per_epoch=1, randomize_rhs_smoothness=False, rhs_random_smoothness_range=(5, 20),
                 randomize_boundary_smoothness=False, boundary_random_smoothness_range=None, randomize_rhs_max_magnitude=False,
                 rhs_random_max_magnitude=1.0, randomize_boundary_max_magnitudes=False, boundary_random_max_magnitudes=None,
                 return_keras_style=True, exclude_zero_boundaries=False, seed=0, device=None, shard=None, **numerical_dataset_arguments):
        self.batch_size, self.batches_per_epoch = int(batch_size), int(batches_per_epoch)
        self.randomize_rhs_smoothness, self.rhs_random_smoothness_range = randomize_rhs_smoothness, rhs_random_smoothness_range
        self.randomize_boundary_smoothness = randomize_boundary_smoothness
        self.boundary_random_smoothness_range = boundary_random_smoothness_range or {k: [5, 20] for k in _BOUNDARY_KEYS}
        self.randomize_rhs_max_magnitude, self.rhs_random_max_magnitude = randomize_rhs_max_magnitude, rhs_random_max_magnitude
        self.randomize_boundary_max_magnitudes = randomize_boundary_max_magnitudes
        self.boundary_random_max_magnitudes = boundary_random_max_magnitudes or {k: 1.0 for k in _BOUNDARY_KEYS}
        self.return_keras_style, self.exclude_zero_boundaries = return_keras_style, exclude_zero_boundaries
        self.nda = dict(numerical_dataset_arguments)
        self.nda.pop('normalizations', None)
        self.rng, self.shape_rng, self.shard = _streams(seed, shard)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())

    def __len__(self):
        return self.batches_per_epoch

    def _dev(self, a):
        """host array -> float32 device tensor through a pinned staging buffer and an asynchronous copy on the current stream: a pageable upload
        (torch.tensor(..., device=...)) synchronises the stream, ten times per batch (profiles/r06_train_shipped.txt); torch's pinned-memory cache
        keeps the staging buffer alive until the copy has run."""
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        if self.device.type != 'cuda':
            return t.to(self.device)
        return t.pin_memory().to(self.device, non_blocking=True)

    def _smooth_field(self, ctrl, out_hw, max_magnitude):
        """generate_random_RHS / generate_random_boundaries core (numerical.py:10-72): control points -> legacy bicubic
        (align_corners) up-sampling -> per-sample max-magnitude scaling."""
        N = ctrl.shape[0]
        x = self._dev(ctrl).view(N, ctrl.shape[1], ctrl.shape[2], 1)
        y = K.resize_legacy_bicubic(x, out_hw).view(N, out_hw[0], out_hw[1])
        if max_magnitude != np.inf:
            K.set_max_magnitude(y, torch.full((N,), float(max_magnitude), device=self.device))
        return y

    def numerical_dataset(self, output_shape='random', dx='random', boundaries='random', rhses='random', rhs_smoothness=None, boundary_smoothness=None,
                          rhs_max_magnitude=1.0, boundary_max_magnitude=None, nonzero_boundaries=_BOUNDARY_KEYS, solver_method='multigrid',
                          return_rhs=True, return_boundaries=False, return_dx=False, return_shape=False, random_output_shape_range=((64, 85), (64, 85)),
                          random_dx_range=(0.005, 0.05), normalize_by_domain_size=False, uniformly_distributed_aspect_ratios=True, boundary_types=None):
        """numerical_dataset (dataset/generators/numerical.py:74-150).  solver_method 'multigrid' / 'multigrid_gpu' / 'cholesky' all map
        to the DST-I direct solve of the same 5-point system.

        boundary_types (extension): dict edge -> 'dirichlet' | 'neumann' (default all Dirichlet).  A Neumann
        edge's random smooth boundary function is its normal derivative du/dn; the discrete mixed-BC system is solved directly
        (mixed_bc_poisson_solve) - a true mixed Dirichlet/Neumann ground truth for BASELINE configs[2]."""
        N, rng, srng = self.batch_size, self.rng, self.shape_rng
        rank, world = self.shard
        boundary_max_magnitude = boundary_max_magnitude or {k: 1.0 for k in _BOUNDARY_KEYS}
        if isinstance(output_shape, str) and output_shape == 'random':       # the grid shape comes from the stream all ranks share (see _streams)
            if uniformly_distributed_aspect_ratios:
                ar = generate_uniformly_distributed_aspect_ratios(random_output_shape_range, None, 1, srng)
                shape, dxg = generate_output_shapes_and_grid_spacings_from_aspect_ratios(ar, random_output_shape_range, [list(random_dx_range)], constant_dx=True,
                                                                                         samples=N * world, rng=srng)
                output_shape = [int(s) for s in shape]
                if isinstance(dx, str) and dx == 'random':
                    dx = dxg[rank * N:(rank + 1) * N, :1]
            else:
                output_shape = [int(srng.integers(r[0], r[1])) for r in random_output_shape_range]
        if isinstance(dx, str) and dx == 'random':
            dx = rng.uniform(size=(N, 1)) * (random_dx_range[1] - random_dx_range[0]) + random_dx_range[0]
        elif isinstance(dx, float):
            dx = np.ones((N, 1)) * dx
        dx = np.asarray(dx, dtype=np.float64).reshape(N, 1)
        H, W = output_shape
        if isinstance(rhses, str) and rhses == 'random':
            nc = rhs_smoothness
            if nc is None:
                nc = [int(rng.integers(5, int(n // 1.5))) for n in (H, W)]
            elif isinstance(nc, int):
                nc = [nc, nc]
            rhs = self._smooth_field(2 * rng.uniform(size=(N, nc[0], nc[1])) - 1, (H, W), rhs_max_magnitude)
        elif isinstance(rhses, str) and rhses == 'zero':
            rhs = torch.zeros((N, H, W), dtype=torch.float32, device=self.device)
        else:
            rhs = self._dev(np.asarray(rhses)).view(N, H, W)
        lengths = {'left': W, 'right': W, 'top': H, 'bottom': H}          # numerical.py:54
        if isinstance(boundaries, str):
            nonzero = list(nonzero_boundaries) if boundaries == 'random' else []
            sm = boundary_smoothness
            if isinstance(sm, int):
                sm = {k: sm for k in _BOUNDARY_KEYS}
            elif sm is None:
                sm = {k: int(rng.integers(5, int(lengths[k] // 1.5))) for k in _BOUNDARY_KEYS}
            bc = {}
            for k in _BOUNDARY_KEYS:
                if k in nonzero:
                    bc[k] = self._smooth_field((2 * rng.uniform(size=(N, sm[k])) - 1)[:, None, :], (1, lengths[k]), boundary_max_magnitude[k]).view(N, lengths[k])
                else:
                    bc[k] = torch.zeros((N, lengths[k]), dtype=torch.float32, device=self.device)
        else:
            bc = {k: self._dev(np.asarray(boundaries[k])).view(N, lengths[k]) for k in _BOUNDARY_KEYS}
        if boundary_types is not None and any(str(v).lower() == 'neumann' for v in boundary_types.values()):
            soln = K.fd_poisson_mixed(rhs, bc['left'], bc['right'], bc['bottom'], bc['top'], self._dev(dx[:, 0]), _neumann_flags(boundary_types))
        else:
            soln = K.fd_poisson_dst(rhs, bc['left'], bc['right'], bc['bottom'], bc['top'], self._dev(dx[:, 0]))
        if normalize_by_domain_size:
            K.scale_samples(soln, self._dev(10.0 / (dx[:, 0] ** 2 * (H - 1) * (W - 1))))
        inp = []
        if return_rhs:
            inp.append(rhs.view(N, 1, H, W))
        if return_boundaries:
            inp.append({k: v.view(N, 1, -1) for k, v in bc.items()})
        if return_dx:
            inp.append(self._dev(dx))
        if return_shape:
            inp.append(torch.tensor([N, 1, H, W], dtype=torch.int32))
        soln = soln.view(N, 1, H, W)
        return (inp, soln) if inp else soln

    def __getitem__(self, idx=0):
        nda, rng = dict(self.nda), self.rng
        if self.randomize_rhs_smoothness:
            nda['rhs_smoothness'] = int(rng.integers(self.rhs_random_smoothness_range[0], self.rhs_random_smoothness_range[1]))
        if self.randomize_boundary_smoothness:
            nda['boundary_smoothness'] = {k: int(rng.integers(v[0], v[1])) for k, v in self.boundary_random_smoothness_range.items()}
        if self.randomize_rhs_max_magnitude:
            nda['rhs_max_magnitude'] = rng.uniform() * self.rhs_random_max_magnitude
        if self.randomize_boundary_max_magnitudes:
            nda['boundary_max_magnitude'] = {k: rng.uniform() * v for k, v in self.boundary_random_max_magnitudes.items()}
        inp, out = self.numerical_dataset(**nda)
        if self.return_keras_style and nda.get('return_boundaries', False):
            loc = 1 if nda.get('return_rhs', True) else 0
            b = inp.pop(loc)
            keys = _BOUNDARY_KEYS if (not self.exclude_zero_boundaries or 'nonzero_boundaries' not in nda) else nda['nonzero_boundaries']
            inp[loc:loc] = [b[k] for k in keys]
        return inp, out


def _neumann_flags(boundary_types):
    bt = {k: 'dirichlet' for k in _BOUNDARY_KEYS}
    for k, v in (boundary_types or {}).items():
        if k not in bt or str(v).lower() not in ('dirichlet', 'neumann'):
            raise ValueError("boundary_types maps 'left'/'right'/'bottom'/'top' to 'dirichlet' or 'neumann' (got %r: %r)" % (k, v))
        bt[k] = str(v).lower()
    return tuple(bt[k] == 'neumann' for k in ('left', 'right', 'bottom', 'top'))


# ----------------------------------------------------------------------------- solver entry points (dataset/solvers)
def multigrid_poisson_solve(rhses, boundaries, dx, dy=None, system_matrix=None, tol=1e-10, solver_init_parameters=None, solver_run_parameters=None,
                            use_pyamgx=False, initial_guesses=None, device=None):
    """Drop-in for dataset/solvers/multigrid.py:98-150: solves pyamg.gallery.poisson((H-2, W-2)) u = poisson_RHS(rhses, boundaries, dx)
    and writes the Dirichlet values - by the direct DST-I diagonalisation on the GPU instead of Ruge-Stuben V-cycles / AMGX
    (tol, solver parameters and initial guesses are accepted and irrelevant for a direct solve; like the reference only `dx` is used).
    rhses (N,1,H,W) or (N,H,W); boundaries: dict with 'left'/'right' (N,W) and 'bottom'/'top' (N,H) (extra singleton dims allowed)."""
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).to(device=dev, dtype=torch.float32)
    r = t(rhses)
    if r.dim() == 4:
        r = r[:, 0]
    N, H, W = r.shape
    b = {k: t(boundaries[k]).reshape(-1, W if k in ('left', 'right') else H).expand(N, -1).contiguous() for k in _BOUNDARY_KEYS}
    d = t(dx).reshape(-1)
    d = d.expand(N).contiguous() if d.numel() == 1 else d.reshape(N, -1)[:, 0].contiguous()   # (N,), (N,1) or (N,2): first column
    out = K.fd_poisson_dst(r.contiguous(), b['left'], b['right'], b['bottom'], b['top'], d)
    return out.view(N, 1, H, W)


def cholesky_poisson_solve(rhses, boundaries, h, system_matrix=None, system_matrix_is_decomposed=False, device=None):
    """dataset/solvers/cholesky.py:122-186 (dense Cholesky of the same 5-point matrix): same system, same direct DST-I solve."""
    return multigrid_poisson_solve(rhses, boundaries, h, device=device)


def mixed_bc_poisson_solve(rhses, boundaries, dx, boundary_types, device=None):
    """5-point FD Poisson solve with per-edge Dirichlet / Neumann conditions on the reference's vertex-centred grid (same edge naming and
    array shapes as multigrid_poisson_solve).  The reference has no such solver in its dataset path - its only Neumann data are analytic
    cosine series (dataset/generators/reverse_neumann.py) - but its Navier-Stokes projection step solves exactly this system
    (Navier_Stokes_2D/solvers.py:225-335, zero-integral constraint for the singular all-Neumann case, :258-259).
    boundary_types: dict edge -> 'dirichlet' | 'neumann'; a Neumann edge's array holds du/dn along the outward normal."""
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).to(device=dev, dtype=torch.float32)
    r = t(rhses)
    if r.dim() == 4:
        r = r[:, 0]
    N, H, W = r.shape
    b = {k: t(boundaries[k]).reshape(-1, W if k in ('left', 'right') else H).expand(N, -1).contiguous() for k in _BOUNDARY_KEYS}
    d = t(dx).reshape(-1)
    d = d.expand(N).contiguous() if d.numel() == 1 else d.reshape(N, -1)[:, 0].contiguous()
    out = K.fd_poisson_mixed(r.contiguous(), b['left'], b['right'], b['bottom'], b['top'], d, _neumann_flags(boundary_types))
    return out.view(N, 1, H, W)
