"""The 64-point tiles of the spectral route (csrc/spectral64.hip: streamed transforms, accumulator-resident spectra) against the fp64
torch-CPU reference at the per-layer tolerances of the 32-point tiles: forward with every epilogue option, data gradient, weight
gradient, the fused backward, all padding modes, ragged channel counts, one-tile and many-tile images with overhanging last tiles, and
agreement with the 32-point route on the same layer."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['mfma', 'fft'])
def tile64_forced(request):
    """Every test of this file runs through BOTH transform kernel families on 64-point tiles: the streamed DFT-as-GEMM kernels of csrc/spectral64.hip and
    the in-register FFTs of csrc/spectral_fft.hip (fft64_fwd_kernel / fft64_inv_kernel, round 5)."""
    from poisson_cnn_amd import ops
    prev = ops.get_spectral_mode(), ops.get_spectral_tile(), ops.get_spectral_transform()
    ops.set_spectral_mode('force')
    ops.set_spectral_tile(64)
    ops.set_spectral_transform(request.param)
    yield
    ops.set_spectral_mode(prev[0])
    ops.set_spectral_tile(prev[1])
    ops.set_spectral_transform(prev[2])


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def nhwc(a):
    return torch.tensor(np.ascontiguousarray(a.transpose(0, 2, 3, 1)), device='cuda')


def test_tile64_route_is_taken():
    """The same layer on 64- and on 32-point tiles: equal to fp32 rounding, not bit-identical (different transforms and summation order)."""
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(2, 150, 131, 32, device='cuda', generator=g)
    w = torch.randn(15, 15, 32, 32, device='cuda', generator=g) * 0.02
    y64 = ops.conv2d_fwd(x, w, None, pad_top=7, pad_left=7)
    ops.set_spectral_tile(32)
    y32 = ops.conv2d_fwd(x, w, None, pad_top=7, pad_left=7)
    err = float((y64 - y32).double().norm() / y32.double().norm())
    assert 0 < err < 2e-6


@pytest.mark.parametrize('k,Cin,Cout,H,W', [(15, 32, 32, 120, 101), (13, 28, 28, 75, 140), (11, 17, 32, 64, 64), (9, 24, 20, 130, 57),
                                         (15, 64, 32, 70, 110), (11, 32, 64, 100, 60), (13, 20, 3, 50, 50), (15, 32, 32, 40, 33)])
def test_forward_data_gradient_and_weight_gradient(k, Cin, Cout, H, W):
    """One-tile and many-tile images, overhanging last tiles, N > 1, ragged channel groups, 64 channels on either side."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(k + Cin + H)
    N, p = 3, k // 2
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xt, wt.permute(3, 2, 0, 1), padding=p)
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    xd, dzd, wd = nhwc(x), nhwc(dz), torch.tensor(w, device='cuda')
    got = ops.conv2d_fwd(xd, wd, None, pad_top=p, pad_left=p).cpu().numpy().transpose(0, 3, 1, 2)
    assert rel(got, y.detach().numpy()) < 2e-6
    if Cout <= 32:
        dw = ops.conv2d_wgrad(xd, dzd, w.shape, pad_top=p, pad_left=p).cpu().numpy()
        assert rel(dw, wt.grad.numpy()) < 5e-6
    if Cin <= 32 or Cin == 64:
        dx = ops.conv2d_fwd(dzd, ops.flip_transpose_weights(wd), None, pad_top=k - 1 - p, pad_left=k - 1 - p).cpu().numpy().transpose(0, 3, 1, 2)
        assert rel(dx, xt.grad.numpy()) < 2e-6


@pytest.mark.parametrize('mode', ['CONSTANT', 'SYMMETRIC', 'REFLECT'])
@pytest.mark.parametrize('act', ['leaky_relu', 'tanh', 'linear'])
def test_padding_modes_and_fused_epilogue(mode, act):
    """tf.pad modes applied by the window loader; bias, activation, act_out, BN affine and residual from the accumulators; channel slices of
    wider buffers on both sides (ldx, ldy)."""
    from oracle import np_ops
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(5)
    N, H, W, Cin, Cout, k = 2, 97, 70, 24, 28, 11
    xw = rng.standard_normal((N, H, W, 40)).astype(np.float32)
    x = xw[..., 8:8 + Cin]
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, Cout).astype(np.float32), rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    p = k // 2
    xp = np_ops.pad2d(x.transpose(0, 3, 1, 2).astype(np.float64), ((p, p), (p, p)), mode, 0.3)
    z = F.conv2d(torch.tensor(xp), torch.tensor(w.astype(np.float64)).permute(3, 2, 0, 1)).numpy() + b[None, :, None, None]
    a = {'leaky_relu': np.where(z > 0, z, 0.2 * z), 'tanh': np.tanh(z), 'linear': z}[act]
    yref = a * sc[None, :, None, None] + sh[None, :, None, None] + res.transpose(0, 3, 1, 2)
    xd = torch.tensor(xw, device='cuda')[..., 8:8 + Cin]
    out = torch.zeros(N, H, W, 64, device='cuda')
    aout = torch.empty(N, H, W, Cout, device='cuda')
    ops.conv2d_fwd(xd, torch.tensor(w, device='cuda'), torch.tensor(b, device='cuda'), pad_top=p, pad_left=p, pad_mode=mode, pad_value=0.3, act=act,
                   bn_scale=torch.tensor(sc, device='cuda'), bn_shift=torch.tensor(sh, device='cuda'), residual=torch.tensor(res, device='cuda'),
                   out=out[..., 32:32 + Cout], act_out=aout)
    assert rel(out[..., 32:32 + Cout].cpu().numpy().transpose(0, 3, 1, 2), yref) < 2e-6
    assert rel(aout.cpu().numpy().transpose(0, 3, 1, 2), a) < 2e-6
    assert float(out[..., :32].abs().max()) == 0.0 and float(out[..., 32 + Cout:].abs().max()) == 0.0


@pytest.mark.parametrize('mode,k,Cin,Cout', [('CONSTANT', 15, 32, 32), ('SYMMETRIC', 11, 32, 32), ('SYMMETRIC', 13, 28, 24), ('CONSTANT', 9, 64, 32)])
def test_fused_backward(mode, k, Cin, Cout):
    """pcnn_conv2d_bwd_spectral on 64-point tiles: the data gradient (with the skip-connection add) and the weight gradient from ONE transform
    of dz, against autograd of the padded convolution."""
    from oracle import np_ops
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(k)
    N, H, W, p = 2, 140, 90, k // 2
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    skip = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    if mode == 'CONSTANT':
        y = F.conv2d(xt, wt.permute(3, 2, 0, 1), padding=p)
    else:
        idx_y = np_ops.pad2d(np.arange(H, dtype=np.float64)[None, None, :, None], ((p, p), (0, 0)), mode, 0.0)[0, 0, :, 0].astype(np.int64)
        idx_x = np_ops.pad2d(np.arange(W, dtype=np.float64)[None, None, None, :], ((0, 0), (p, p)), mode, 0.0)[0, 0, 0, :].astype(np.int64)
        y = F.conv2d(xt[:, :, torch.tensor(idx_y)][:, :, :, torch.tensor(idx_x)], wt.permute(3, 2, 0, 1))
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    xd, dzd, wd = nhwc(x), nhwc(dz), torch.tensor(w, device='cuda')
    dw = torch.zeros_like(wd)
    wf = ops.flip_transpose_weights(wd)
    res = nhwc(skip) if mode == 'CONSTANT' else None
    out = ops.conv2d_bwd_fused(xd, dzd, w.shape, wf, pad_top=p, pad_left=p, pad_mode=mode, pad_value=0.0, dw=dw, residual=res)
    assert out is not None, 'the layer must be eligible for the fused spectral backward'
    assert rel(dw.cpu().numpy(), wt.grad.numpy()) < 5e-6
    if mode == 'CONSTANT':
        assert rel(out.cpu().numpy().transpose(0, 3, 1, 2), xt.grad.numpy() + skip) < 2e-6
    else:
        dx = ops.pad_fold_bwd(out, (H, W), ((p, p), (p, p)), mode)
        assert rel(dx.cpu().numpy().transpose(0, 3, 1, 2), xt.grad.numpy()) < 2e-6


def test_workspace_limit_shrinks_the_launches_not_the_result():
    """pcnn_set_workspace_limit: the caller's cap makes the route run in smaller tile chunks; results are bit-identical; a cap below what one
    minimal chunk needs is an ordinary error."""
    import ctypes
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(4, 300, 300, 32, device='cuda', generator=g)
    w = torch.randn(13, 13, 32, 32, device='cuda', generator=g) * 0.02
    y0 = ops.conv2d_fwd(x, w, None, pad_top=6, pad_left=6)
    h = ops.handle()
    try:
        h.call('pcnn_set_workspace_limit', ctypes.c_size_t(160 << 20))
        y1 = ops.conv2d_fwd(x, w, None, pad_top=6, pad_left=6)
        assert torch.equal(y0, y1)
        h.call('pcnn_set_workspace_limit', ctypes.c_size_t(8 << 20))
        with pytest.raises(RuntimeError, match='workspace'):
            ops.conv2d_fwd(x, w, None, pad_top=6, pad_left=6)
    finally:
        h.call('pcnn_set_workspace_limit', ctypes.c_size_t(0))
    assert torch.equal(ops.conv2d_fwd(x, w, None, pad_top=6, pad_left=6), y0)


@pytest.mark.parametrize('ylim,xlim,C', [(64, 64, 32), (64, 64, 20), (50, 50, 32), (37, 64, 7), (15, 15, 64)])
def test_the_two_forms_of_the_forward_transform_write_the_same_spectrum(ylim, xlim, C):
    """Round 4: the 64-point forward transform with a second radix-2 step on the y axis (spec64_fwd4_kernel: classes fy mod 4, sixteen-point
    transforms, Re / Im stacked in one accumulator tile) against round 3's parity form, row by row on one window (pcnn_debug_tile_spectrum64) -
    full and masked windows, ragged channel groups - and against numpy's FFT for the rows whose meaning spectral_common.h states."""
    import ctypes
    import os
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.ops import _p
    g = torch.Generator(device='cuda').manual_seed(ylim + C)
    x = torch.randn(64, 64, C, device='cuda', generator=g)
    groups = (C + 31) // 32

    def spectrum(radix):
        os.environ['PCNN_FWD64_RADIX'] = str(radix)
        try:
            out = torch.zeros(groups * 4096, 32, device='cuda')
            ops.handle().call('pcnn_debug_tile_spectrum64', ctypes.c_int(64), ctypes.c_int(64), ctypes.c_int(C), _p(x), ctypes.c_int(ylim), ctypes.c_int(xlim), _p(out))
            torch.cuda.synchronize()
            return out.cpu().numpy().astype(np.float64)
        finally:
            os.environ.pop('PCNN_FWD64_RADIX', None)
    a, b = spectrum(2), spectrum(4)
    scale = np.abs(a).max()
    assert np.abs(a - b).max() < 2e-6 * scale
    xm = x.cpu().numpy().astype(np.float64).copy()
    xm[ylim:, :, :] = 0
    xm[:, xlim:, :] = 0
    X = np.fft.fft2(xm, axes=(0, 1))                                        # X[fy][fx][c]
    for fx, fy in ((5, 9), (31, 63), (1, 0), (17, 32)):
        row = 128 + 128 * (fx - 1) + fy
        assert abs(b[row, 0] - X[fy, fx, 0].real) < 1e-5 * scale and abs(b[row + 64, 0] - X[fy, fx, 0].imag) < 1e-5 * scale
    for fx, base in ((0, 0), (32, 64)):                                     # the two real columns: half-complex rows
        for fy in (0, 1, 2, 3, 13, 30, 31, 32):
            assert abs(b[base + fy, 0] - X[fy, fx, 0].real) < 1e-5 * scale
            if 0 < fy < 32:
                assert abs(b[base + 32 + fy, 0] - X[fy, fx, 0].imag) < 1e-5 * scale
