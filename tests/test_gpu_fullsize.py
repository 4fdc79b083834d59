"""Size-independent properties at BASELINE.json's full sizes (8 x 1024^2 per GPU), where the fp64 oracle would take hours:
bilinearity / adjointness of the three convolution kernels (the identities <conv(x; w), g> = <w, wgrad(x, g)> = <x, dgrad(g; w)>
that make them one operator and its two transposes), linearity, per-sample independence of the whole model, finiteness of a full
training step, and the discrete residual of the on-device FD Poisson solve."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dot64(a, b):
    return float((a.double() * b.double()).sum())


# (taps, Cin, Cout) at 8 x 1024^2: the final stack's 15- and 13-tap layers and the 11-tap layer of the pre-bottleneck stack - at this size the
# 11-tap layer switches to 64-point tiles (pick_tile: >= 256 tiles per image, never true at 512^2), which no oracle fixture of the BACKWARD pass
# covers: the identities below tie its forward, weight-gradient, data-gradient and fused-backward kernels to each other
FULL_SHAPES = [(15, 32, 32, 'fp32'), (15, 32, 32, 'split_f16'), (13, 28, 28, 'fp32'), (11, 16, 32, 'fp32'), (9, 24, 24, 'fp32')]


@pytest.mark.parametrize('k,ci,co,mode', FULL_SHAPES)
@pytest.mark.parametrize('pad_mode', ['SYMMETRIC', 'CONSTANT'])
def test_conv_adjoint_identities_at_full_size(k, ci, co, mode, pad_mode):
    from poisson_cnn_amd import ops
    prev = ops.get_math_mode()
    ops.set_math_mode(mode)
    try:
        g = torch.Generator(device='cuda').manual_seed(1)
        N, H, W = 8, 1024, 1024
        x = torch.randn(N, H, W, ci, device='cuda', generator=g)
        w = torch.randn(k, k, ci, co, device='cuda', generator=g) * 0.02
        gy = torch.randn(N, H, W, co, device='cuda', generator=g)
        pt = k // 2
        y = ops.conv2d_fwd(x, w, None, pad_top=pt, pad_left=pt, pad_mode=pad_mode)
        ref = dot64(y, gy)
        # <y, g> sums 2.7e8 random-sign terms: the rounding noise of y (~1e-6 relative per element) enters as a random walk
        tol = 2e-6 * abs(ref) + 3e-6 * np.sqrt(y.numel()) * float(y.double().pow(2).mean().sqrt()) * float(gy.double().pow(2).mean().sqrt())
        dw = ops.conv2d_wgrad(x, gy, w.shape, pad_top=pt, pad_left=pt, pad_mode=pad_mode)
        assert abs(dot64(w, dw) - ref) < tol
        wf = ops.flip_transpose_weights(w)
        if pad_mode == 'CONSTANT':
            dx = ops.conv2d_fwd(gy, wf, None, pad_top=k - 1 - pt, pad_left=k - 1 - pt)
        else:
            gp = ops.conv2d_fwd(gy, wf, None, pad_top=k - 1, pad_left=k - 1, out_hw=(H + k - 1, W + k - 1))
            dx = ops.pad_fold_bwd(gp, (H, W), ((pt, pt), (pt, pt)), pad_mode)
        assert abs(dot64(x, dx) - ref) < tol
        # the fused backward of the spectral route (one pass: dz's spectrum shared by both gradients) obeys the same two identities
        dwf = torch.zeros_like(w)
        out = ops.conv2d_bwd_fused(x, gy, w.shape, wf, pad_top=pt, pad_left=pt, pad_mode=pad_mode, dw=dwf)
        if out is not None:
            dxf = out if pad_mode == 'CONSTANT' else ops.pad_fold_bwd(out, (H, W), ((pt, pt), (pt, pt)), pad_mode)
            assert abs(dot64(w, dwf) - ref) < tol and abs(dot64(x, dxf) - ref) < tol
            assert float((dwf - dw).double().norm() / dw.double().norm()) < 5e-6 and float((dxf - dx).double().norm() / dx.double().norm()) < 5e-6
        else:
            assert mode == 'split_f16'
        del dwf, out
        # linearity in x
        x2 = torch.randn(N, H, W, ci, device='cuda', generator=g)
        y2 = ops.conv2d_fwd(x2, w, None, pad_top=pt, pad_left=pt, pad_mode=pad_mode)
        y12 = ops.conv2d_fwd(0.75 * x - 1.5 * x2, w, None, pad_top=pt, pad_left=pt, pad_mode=pad_mode)
        err = (y12 - (0.75 * y - 1.5 * y2)).double().norm() / y12.double().norm()
        assert float(err) < 2e-6
    finally:
        ops.set_math_mode(prev)


@pytest.mark.parametrize('f', [8, 32, 128])
def test_transposed_convolution_adjoint_identities_at_full_size(f):
    """8 x 1024^2 x 32 channels, the merge of hpnn.json's bottleneck branches (blocks/bottleneck_block_deconvupsample -> layers/deconvupscale.py:100-106):
    <deconv(x; K), g> = <x, deconv_bwd_data(g; K)> = <K, deconv_bwd_filter(x, g)> tie the three kernels of csrc/deconv_mfma.hip to each other at the size
    where no oracle runs - factor 8 (many coarse pixels, few taps), 32 and 128 (an 8 x 8 coarse grid with 16 384 taps: the data gradient's four waves share the
    taps) - and the accumulate mode of the forward (the branch merge: y <- beta y + alpha deconv) against its write mode."""
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(f)
    N, H, C = 8, 1024, 32
    hc = H // f
    x = torch.randn(N, hc, hc, C, device='cuda', generator=g)
    K = torch.randn(f, f, C, C, device='cuda', generator=g) * 0.1
    gy = torch.randn(N, H, H, C, device='cuda', generator=g)
    y = ops.deconv_fwd(x, K, None, (H, H), f)
    dx = ops.deconv_bwd_data(gy, K, (hc, hc), f)
    dK = ops.deconv_bwd_filter(x, gy, f)
    a, b, c = dot64(y, gy), dot64(x, dx), dot64(K, dK)
    scale = float(y.double().norm() * gy.double().norm())
    assert abs(a - b) < 2e-6 * scale and abs(a - c) < 2e-6 * scale, (a, b, c)
    acc = gy.clone()
    ops.deconv_fwd(x, K, None, (H, H), f, alpha=0.5, beta=2.0, out=acc)
    ref = 2.0 * gy.double() + 0.5 * y.double()
    assert float((acc.double() - ref).norm() / ref.norm()) < 1e-6


@pytest.mark.parametrize('mode', ['fp32', 'split_f16'])
def test_model_at_c4_size_sample_independence_and_train_step(mode):
    """hpnn.json at 8 x 1024^2 (bench workload c4; fp32 is the benchmarked library default, split_f16 the opt-in mode): every sample's solution is independent of its batch
    neighbours (bit-exact: inference-mode BN, per-tile arithmetic), the boundary ring is zero (Dirichlet), and one full training step
    leaves finite loss, gradients and weights."""
    from poisson_cnn_amd import configs, ops
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    from poisson_cnn_amd.train import Adam
    prev = ops.get_math_mode()
    ops.set_math_mode(mode)
    try:
        full = configs.hpnn()
        model = Homogeneous_Poisson_NN_Legacy(**full['model'])
        g = torch.Generator().manual_seed(4)
        rhs = torch.rand((8, 1, 1024, 1024), generator=g) * 2 - 1
        rhs = (rhs / rhs.abs().amax(dim=(1, 2, 3), keepdim=True)).cuda()
        dx = (torch.rand((8, 1), generator=g) * 4.5e-2 + 5e-3).cuda()
        y = model([rhs, dx])
        assert torch.isfinite(y).all()
        assert float(y[:, :, 0, :].abs().max()) == 0.0 and float(y[:, :, :, -1].abs().max()) == 0.0
        y3 = model([rhs[3:4].contiguous(), dx[3:4].contiguous()])
        assert torch.equal(y3, y[3:4])
        model.compile(loss=loss_wrapper(global_batch_size=8, **full['training']['loss_parameters']), optimizer=Adam(**full['training']['optimizer_parameters']))
        w0 = model.store.flat_w.clone()
        logs = model.train_step(((rhs, dx), torch.randn((8, 1, 1024, 1024), generator=g).cuda() * 0.1))
        assert np.isfinite(float(logs['loss'])) and np.isfinite(float(logs['mse']))
        assert torch.isfinite(model.store.flat_g).all() and float(model.store.flat_g.abs().max()) > 0
        assert torch.isfinite(model.store.flat_w).all() and not torch.equal(w0, model.store.flat_w)
        # determinism: every reduction of the backward pass adds its partial sums in a fixed order (split-K weight gradients, the fused mixing kernel's
        # tile sums, per-lane bias sums of the POST epilogues, the transposed convolution's cross-wave sums), also with one stream per bottleneck branch:
        # two evaluations of the same step give bit-identical gradients
        tgt = torch.randn((8, 1, 1024, 1024), generator=g).cuda() * 0.1
        dxc = dx.reshape(8, -1)[:, :1].contiguous()
        model._loss_and_grads(rhs, dxc, tgt)
        g1 = model.store.flat_g.clone()
        model._loss_and_grads(rhs, dxc, tgt)
        torch.cuda.synchronize()
        assert torch.equal(g1, model.store.flat_g)
    finally:
        ops.set_math_mode(prev)


def test_fd_solve_residual_at_1024():
    """5-point residual of the DST-I direct solve on a 1024 x 1024 grid (max|lap_h u - f| <= 1e-4 max|f| in fp32)."""
    from poisson_cnn_amd.dataset import _kernels as K
    g = torch.Generator(device='cuda').manual_seed(2)
    N, H, W = 2, 1024, 1024
    t = torch.linspace(0, 1, H, device='cuda')
    rhs = torch.sin(3 * np.pi * t)[None, :, None] * torch.cos(2 * np.pi * t)[None, None, :] + 0.3 * torch.randn(N, 1, 1, device='cuda', generator=g)
    rhs = rhs.expand(N, H, W).contiguous()
    edges = [0.5 * torch.sin((j + 1) * np.pi * t)[None, :].expand(N, H).contiguous() for j in range(4)]
    dx = torch.tensor([0.01, 0.03], device='cuda')
    u = K.fd_poisson_dst(rhs, edges[0], edges[1], edges[2], edges[3], dx).double()
    lap = (u[:, 2:, 1:-1] + u[:, :-2, 1:-1] + u[:, 1:-1, 2:] + u[:, 1:-1, :-2] - 4 * u[:, 1:-1, 1:-1]) / dx.double()[:, None, None] ** 2
    res = (lap - rhs.double()[:, 1:-1, 1:-1]).abs().amax(dim=(1, 2))
    # the solve is fp64; the residual is the fp32 rounding of u (2^-24 |u|) amplified by 4/dx^2
    bound = 8 * (2.0 ** -24) * u.abs().amax(dim=(1, 2)) / dx.double() ** 2
    assert bool((res <= bound + 1e-4 * rhs.abs().amax()).all()), (res, bound)
