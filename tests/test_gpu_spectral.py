"""The tiled spectral convolution route (csrc/spectral_conv.hip: overlap-save on 32 x 32 tiles, DFT as fp32 MFMA products) against the
same fp64 oracle and at the same per-layer tolerance as the direct implicit-GEMM kernels: forward with every epilogue option, data
gradient, weight gradient, all padding modes, ragged channel counts, tiles that overhang the image, and the whole model."""
import numpy as np
import pytest
import torch

from oracle import np_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['mfma', 'fft'])
def spectral_forced(request):
    """Every test of this file runs through BOTH transform kernel families of the route: the DFT as a GEMM on the matrix cores and the in-register
    FFTs on the vector ALUs (csrc/spectral_fft.hip, round 5) - same spectrum layout, same tolerances."""
    from poisson_cnn_amd import ops
    prev, prev_x = ops.get_spectral_mode(), ops.get_spectral_transform()
    ops.set_spectral_mode('force')
    ops.set_spectral_transform(request.param)
    yield
    ops.set_spectral_mode(prev)
    ops.set_spectral_transform(prev_x)


def test_route_is_taken_and_differs_from_direct_only_by_rounding():
    """Same layer through both routes: results agree to fp32 rounding but are not bit-identical (different summation order) - i.e. the
    forced mode really runs the spectral kernels."""
    from poisson_cnn_amd import ops
    assert ops.handle().lib.pcnn_get_spectral_mode(ops.handle()._h) == 1
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(2, 70, 83, 32, device='cuda', generator=g)
    w = torch.randn(15, 15, 32, 32, device='cuda', generator=g) * 0.02
    ys = ops.conv2d_fwd(x, w, None, pad_top=7, pad_left=7)
    ops.set_spectral_mode('off')
    yd = ops.conv2d_fwd(x, w, None, pad_top=7, pad_left=7)
    err = float((ys - yd).double().norm() / yd.double().norm())
    assert 0 < err < 2e-6


def test_forward_cases():
    import test_gpu_conv as t
    for case in t.CASES:
        t.test_padded_conv_matches_oracle(*case)
    t.test_conv_epilogue_bn_residual_slices()
    t.test_flip_transpose_and_data_gradient()


def test_backward_cases():
    import test_gpu_ops as t
    for case in t.BWD_CASES:
        t.test_conv_backward_matches_autograd(*case)


@pytest.mark.parametrize('k,Cin,Cout,H,W', [(15, 32, 32, 120, 101), (13, 28, 28, 75, 140), (7, 64, 32, 90, 90), (5, 20, 16, 64, 64),
                                         # <= 16 channels: 2 / 4 / 8 x-adjacent tiles share the 32 lanes (ragged last groups included)
                                         (15, 3, 4, 100, 171), (13, 4, 16, 75, 140), (7, 16, 16, 90, 95), (7, 8, 5, 64, 230), (9, 2, 7, 40, 33)])
def test_many_tiles_forward_and_weight_gradient(k, Cin, Cout, H, W):
    """Several tiles per image in both directions (interior tiles, overhanging last tiles, N > 1) against torch-CPU fp64."""
    import torch.nn.functional as F
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(k + Cin)
    N, p = 3, k // 2
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xt, wt.permute(3, 2, 0, 1), padding=p)
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), device='cuda')
    dzd = torch.tensor(np.ascontiguousarray(dz.transpose(0, 2, 3, 1)), device='cuda')
    wd = torch.tensor(w, device='cuda')
    got = ops.conv2d_fwd(xd, wd, None, pad_top=p, pad_left=p).cpu().numpy().transpose(0, 3, 1, 2)
    ref = y.detach().numpy()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-6
    dw = ops.conv2d_wgrad(xd, dzd, w.shape, pad_top=p, pad_left=p).cpu().numpy()
    assert np.linalg.norm(dw - wt.grad.numpy()) / np.linalg.norm(wt.grad.numpy()) < 5e-6
    dx = ops.conv2d_fwd(dzd, ops.flip_transpose_weights(wd), None, pad_top=k - 1 - p, pad_left=k - 1 - p).cpu().numpy().transpose(0, 3, 1, 2)
    assert np.linalg.norm(dx - xt.grad.numpy()) / np.linalg.norm(xt.grad.numpy()) < 2e-6


def test_whole_model_forward_and_gradients():
    import test_gpu_model as t
    t.test_hpnn_forward_matches_oracle('dirichlet')
    t.test_forward_matches_committed_golden_vectors()
    t.test_tiny_model_train_step('neumann', 6e-4)
    t.test_hpnn_train_step_gradients('tf.nn.tanh', 3e-4)


def test_adjoint_identities_at_full_size():
    """8 x 1024^2, 15 x 15, 32 -> 32: <conv(x; w), g> = <w, wgrad(x, g)> = <x, dgrad(g; w)> through the spectral route."""
    import test_gpu_fullsize as t
    t.test_conv_adjoint_identities_at_full_size(15, 32, 32, 'fp32', 'CONSTANT')
    t.test_conv_adjoint_identities_at_full_size(15, 32, 32, 'fp32', 'SYMMETRIC')


@pytest.mark.parametrize('k,Cin,Cout,mode', [(15, 32, 32, 'CONSTANT'), (13, 28, 28, 'CONSTANT'), (7, 64, 32, 'CONSTANT'), (11, 16, 32, 'SYMMETRIC'), (9, 24, 20, 'REFLECT'),
                                             (6, 20, 16, 'SYMMETRIC'), (15, 3, 4, 'SYMMETRIC'), (7, 16, 12, 'CONSTANT'), (9, 8, 8, 'REFLECT'), (13, 4, 16, 'CONSTANT')])
def test_fused_backward_shares_the_gradient_spectrum(k, Cin, Cout, mode):
    """pcnn_conv2d_bwd_spectral: data gradient and (input-partitioned, tap-reversed) weight gradient from ONE transform of dz, against
    autograd of the fp64 twin - all padding modes (the padded-domain form for SYMMETRIC / REFLECT), even and odd filters, 64 input channels,
    images that the tile grid overhangs, and the residual (skip-connection) add."""
    from oracle import torch_twin
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(k * 7 + Cin)
    N, H, W = 2, 61, 83
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32).astype(np.float64)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32).astype(np.float64)
    dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32).astype(np.float64)
    xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
    (torch_twin.padded_conv2d(xt, wt, None, mode, 0.0, 'linear') * torch.tensor(dz)).sum().backward()
    pt, pb = np_ops.advanced_pad_amounts(k)
    xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), dtype=torch.float32, device='cuda')
    dzd = torch.tensor(np.ascontiguousarray(dz.transpose(0, 2, 3, 1)), dtype=torch.float32, device='cuda')
    wd = torch.tensor(w, dtype=torch.float32, device='cuda')
    dw = torch.empty_like(wd)
    res = torch.tensor(rng.standard_normal((N, H, W, Cin)), dtype=torch.float32, device='cuda') if mode == 'CONSTANT' else None
    out = ops.conv2d_bwd_fused(xd, dzd, wd.shape, ops.flip_transpose_weights(wd), pad_top=pt, pad_left=pt, pad_mode=mode, dw=dw, residual=res)
    assert out is not None
    dx = out - res if mode == 'CONSTANT' else ops.pad_fold_bwd(out, (H, W), ((pt, pb), (pt, pb)), mode)
    ref_dx, ref_dw = xt.grad.numpy(), wt.grad.numpy()
    assert np.linalg.norm(dx.cpu().numpy().transpose(0, 3, 1, 2) - ref_dx) / np.linalg.norm(ref_dx) < 3e-6
    assert np.linalg.norm(dw.cpu().numpy() - ref_dw) / np.linalg.norm(ref_dw) < 5e-6


def test_fused_backward_in_chunks_under_a_workspace_limit():
    """The fused backward under pcnn_set_workspace_limit: the tile grid is walked in several launches, the fused mixing kernel (spec_mixw_kernel: channel
    mixing of the data gradient + weight-gradient GEMM from ONE read of the dz spectrum) ACCUMULATES its partial sums from chunk to chunk.  The data
    gradient is bit-identical to the unchunked call (a tile's arithmetic does not depend on its launch), the weight gradient agrees to rounding (its
    tile sum is split differently) and with the fp64 oracle."""
    import ctypes
    import torch.nn.functional as F
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(21)
    N, H, W, k, C = 4, 150, 150, 7, 32
    x = torch.randn(N, H, W, C, device='cuda', generator=g)
    dz = torch.randn(N, H, W, C, device='cuda', generator=g)
    w = torch.randn(k, k, C, C, device='cuda', generator=g) * 0.03
    wf = ops.flip_transpose_weights(w)
    ops.set_spectral_mode('force')
    h = ops.handle()
    try:
        dw0 = torch.empty_like(w)
        dx0 = ops.conv2d_bwd_fused(x, dz, w.shape, wf, pad_top=3, pad_left=3, pad_mode='CONSTANT', pad_value=0.0, dw=dw0, residual=None).clone()
        h.call('pcnn_set_workspace_limit', ctypes.c_size_t(100 << 20))           # 144 tiles -> four launches of 36
        dw1 = torch.empty_like(w)
        dx1 = ops.conv2d_bwd_fused(x, dz, w.shape, wf, pad_top=3, pad_left=3, pad_mode='CONSTANT', pad_value=0.0, dw=dw1, residual=None)
        assert torch.equal(dx0, dx1)
        assert float((dw1 - dw0).norm() / dw0.norm()) < 1e-6
    finally:
        h.call('pcnn_set_workspace_limit', ctypes.c_size_t(0))
        ops.set_spectral_mode('auto')
    xt = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wt = w.permute(3, 2, 0, 1).double().cpu().requires_grad_(True)
    (F.conv2d(xt, wt, padding=3) * dz.permute(0, 3, 1, 2).double().cpu()).sum().backward()
    assert float((dx1.permute(0, 3, 1, 2).double().cpu() - xt.grad).norm() / xt.grad.norm()) < 3e-6
    assert float((dw1.permute(3, 2, 0, 1).double().cpu() - wt.grad).norm() / wt.grad.norm()) < 5e-6


def test_random_narrow_shapes_through_the_packed_route():
    """Twenty seeded random layers of <= 16 channels (2 / 4 / 8 tiles per work item): even and odd filters 4 ... 15, images from one tile
    group up to a few, ragged last groups, every padding mode, N up to 3 - forward, data gradient and weight gradient of the fused backward
    against fp64 autograd."""
    from oracle import torch_twin
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(2024)
    for case in range(20):
        k = int(rng.integers(4, 16))
        Cin, Cout = int(rng.integers(1, 17)), int(rng.integers(1, 17))
        N, H, W = int(rng.integers(1, 4)), int(rng.integers(k + 3, 110)), int(rng.integers(k + 3, 130))
        mode = ['CONSTANT', 'SYMMETRIC', 'REFLECT'][case % 3]
        if k <= 5 and k in (3, 5):
            continue                                       # the narrow vector-ALU kernels own 3x3 / 5x5 below 17 channels
        x = rng.standard_normal((N, Cin, H, W)).astype(np.float32).astype(np.float64)
        w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32).astype(np.float64)
        dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32).astype(np.float64)
        xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
        y = torch_twin.padded_conv2d(xt, wt, None, mode, 0.0, 'linear')
        (y * torch.tensor(dz)).sum().backward()
        pt, pb = np_ops.advanced_pad_amounts(k)
        xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), dtype=torch.float32, device='cuda')
        dzd = torch.tensor(np.ascontiguousarray(dz.transpose(0, 2, 3, 1)), dtype=torch.float32, device='cuda')
        wd = torch.tensor(w, dtype=torch.float32, device='cuda')
        tag = 'case %d: k=%d %d->%d N=%d %dx%d %s' % (case, k, Cin, Cout, N, H, W, mode)
        got = ops.conv2d_fwd(xd, wd, None, pad_top=pt, pad_left=pt, pad_mode=mode).cpu().numpy().transpose(0, 3, 1, 2)
        ref = y.detach().numpy()
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-6, tag
        dw = torch.empty_like(wd)
        out = ops.conv2d_bwd_fused(xd, dzd, wd.shape, ops.flip_transpose_weights(wd), pad_top=pt, pad_left=pt, pad_mode=mode, dw=dw)
        assert out is not None, tag
        dx = out if mode == 'CONSTANT' else ops.pad_fold_bwd(out, (H, W), ((pt, pb), (pt, pb)), mode)
        ref_dx, ref_dw = xt.grad.numpy(), wt.grad.numpy()
        assert np.linalg.norm(dx.cpu().numpy().transpose(0, 3, 1, 2) - ref_dx) / np.linalg.norm(ref_dx) < 3e-6, tag
        assert np.linalg.norm(dw.cpu().numpy() - ref_dw) / np.linalg.norm(ref_dw) < 5e-6, tag


def test_random_wide_shapes():
    """Twelve seeded random layers with 17 ... 64 input and 17 ... 32 output channels (one or two channel groups, ragged counts)."""
    import torch.nn.functional as F
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(7)
    for case in range(12):
        k = int(rng.integers(2, 16))
        Cin, Cout = int(rng.integers(17, 65)), int(rng.integers(17, 33))
        N, H, W = int(rng.integers(1, 3)), int(rng.integers(k + 3, 90)), int(rng.integers(k + 3, 100))
        x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
        dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
        pt, pb = np_ops.advanced_pad_amounts(k)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(F.pad(xt, (pt, pb, pt, pb)), wt.permute(3, 2, 0, 1))
        (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
        xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), device='cuda')
        dzd = torch.tensor(np.ascontiguousarray(dz.transpose(0, 2, 3, 1)), device='cuda')
        wd = torch.tensor(w, device='cuda')
        tag = 'case %d: k=%d %d->%d N=%d %dx%d' % (case, k, Cin, Cout, N, H, W)
        got = ops.conv2d_fwd(xd, wd, None, pad_top=pt, pad_left=pt).cpu().numpy().transpose(0, 3, 1, 2)
        assert np.linalg.norm(got - y.detach().numpy()) / np.linalg.norm(y.detach().numpy()) < 2e-6, tag
        dw = ops.conv2d_wgrad(xd, dzd, w.shape, pad_top=pt, pad_left=pt).cpu().numpy()
        assert np.linalg.norm(dw - wt.grad.numpy()) / np.linalg.norm(wt.grad.numpy()) < 5e-6, tag
        dx = ops.conv2d_fwd(dzd, ops.flip_transpose_weights(wd), None, pad_top=k - 1 - pt, pad_left=k - 1 - pt).cpu().numpy().transpose(0, 3, 1, 2)
        assert np.linalg.norm(dx - xt.grad.numpy()) / np.linalg.norm(xt.grad.numpy()) < 2e-6, tag
