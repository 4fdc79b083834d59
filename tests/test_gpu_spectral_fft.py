"""The transform kernels of the spectral route as in-register FFTs on the vector ALUs (csrc/spectral_fft.hip, csrc/fft_regs.h; round 5) against
the matrix-core DFT-as-GEMM kernels they replace (csrc/spectral_conv.hip) and against the fp64 oracle:

  * the tile spectra the two forward transforms write agree ROW BY ROW to <= 1e-6 of the spectrum's magnitude - every padding mode, boundary and
    interior windows, masked (gradient-style) tiles, the filter-style one-tile image, tile packing for <= 16 channels, ragged channel groups;
  * forward / weight gradient / fused backward of whole layers through the FFT kernels against the oracle at the tolerances of the matrix-core
    route (tests/test_gpu_spectral.py runs its whole file under both transforms; here: the epilogue variants of the inverse kernel one by one).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PAD = {'CONSTANT': 0, 'SYMMETRIC': 1, 'REFLECT': 2}


def _spectrum(x, Vy, Vx, oy, ox, mode, pad_value, ylim, xlim, pack, xform):
    from poisson_cnn_amd import ops
    H, W, C = x.shape
    tiles = -(-H // Vy) * -(-(-(-W // Vx)) // pack)
    groups = 1 if pack > 1 else -(-C // 32)
    out = torch.zeros(tiles * groups, 1024, 32, device='cuda')
    prev = ops.get_spectral_transform()
    ops.set_spectral_transform(xform)
    try:
        ops.handle().call('pcnn_debug_forward_spectrum32', H, W, C, ctypes.c_void_p(x.data_ptr()), Vy, Vx, oy, ox, PAD[mode], float(pad_value), ylim, xlim, pack,
                          ctypes.c_void_p(out.data_ptr()), out.numel())
        torch.cuda.synchronize()
    finally:
        ops.set_spectral_transform(prev)
    return out


@pytest.mark.parametrize('H,W,C,k,mode,pad_value,ylim,xlim,pack', [
    (70, 83, 32, 7, 'CONSTANT', 0.0, 32, 32, 1),          # forward-style windows with halo, boundary + interior tiles
    (70, 83, 32, 15, 'CONSTANT', 1.5, 32, 32, 1),         # non-zero padding constant
    (64, 100, 20, 9, 'SYMMETRIC', 0.0, 32, 32, 1),        # ragged channel group, index-mapped boundary
    (50, 41, 40, 11, 'REFLECT', 0.0, 32, 32, 1),          # two channel groups (32 + 8)
    (90, 75, 32, 7, 'CONSTANT', 0.0, 26, 26, 1),          # masked: the tile's own V x V values, zero elsewhere (no halo)
    (7, 7, 64, 1, 'CONSTANT', 0.0, 7, 7, 1),              # a filter as a one-tile image (kh x kw corner of the tile)
    (40, 171, 4, 15, 'CONSTANT', 0.0, 32, 32, 8),         # tile packing: 8 x-adjacent tiles of a 4-channel image per item
    (60, 140, 16, 13, 'SYMMETRIC', 0.0, 32, 32, 2),       # 2 tiles of 16 channels
    (64, 230, 5, 7, 'REFLECT', 0.0, 26, 26, 4),           # packed + masked + ragged last group
])
def test_forward_spectra_agree_row_by_row(H, W, C, k, mode, pad_value, ylim, xlim, pack):
    g = torch.Generator(device='cuda').manual_seed(H + W + C)
    x = torch.randn(H, W, C, device='cuda', generator=g)
    V = 33 - k
    halo = ylim == 32
    oy = ox = (k // 2) if halo else 0
    ref = _spectrum(x, V, V, oy, ox, mode, pad_value, ylim, xlim, pack, 'mfma')
    got = _spectrum(x, V, V, oy, ox, mode, pad_value, ylim, xlim, pack, 'fft')
    scale = float(ref.abs().max())
    assert scale > 0
    err = float((got - ref).abs().max()) / scale
    assert err < 1e-6, err
    # ... and per tile in the L2 sense (a whole wrong row of a small-magnitude tile must not hide behind the global maximum)
    num = (got - ref).double().flatten(1).norm(dim=1)
    den = ref.double().flatten(1).norm(dim=1).clamp_min(1e-30)
    assert float((num / den).max()) < 1e-6


@pytest.mark.parametrize('act,bn,res,act_out', [('linear', False, False, False), ('leaky_relu', True, True, True), ('tanh', False, True, False), ('tanh', True, False, True),
                                               ('relu', False, False, True)])
def test_inverse_epilogues_match_the_oracle(act, bn, res, act_out):
    from oracle import np_ops
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(11)
    N, H, W, Cin, Cout, k = 2, 61, 75, 24, 28, 9
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32) * 0.1
    sc = (1 + 0.2 * rng.standard_normal(Cout)).astype(np.float32)
    sh = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    r = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    ACT = {'linear': 'linear', 'leaky_relu': 'tf.nn.leaky_relu', 'tanh': 'tf.nn.tanh', 'relu': 'tf.nn.relu'}
    a = np_ops.padded_conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 'SYMMETRIC', 0.0, ACT[act])
    ref = a * sc.reshape(1, -1, 1, 1) + sh.reshape(1, -1, 1, 1) if bn else a
    if res:
        ref = ref + r
    xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), device='cuda')
    rd = torch.tensor(np.ascontiguousarray(r.transpose(0, 2, 3, 1)), device='cuda')
    prev_m, prev_x, prev_t = ops.get_spectral_mode(), ops.get_spectral_transform(), ops.get_spectral_tile()
    ops.set_spectral_mode('force'); ops.set_spectral_transform('fft'); ops.set_spectral_tile(32)
    try:
        ao = torch.empty(N, H, W, Cout, device='cuda') if act_out else None
        y = ops.conv2d_fwd(xd, torch.tensor(w, device='cuda'), torch.tensor(b, device='cuda'), pad_top=k // 2, pad_left=k // 2, pad_mode='SYMMETRIC', act=act,
                           bn_scale=torch.tensor(sc, device='cuda') if bn else None, bn_shift=torch.tensor(sh, device='cuda') if bn else None,
                           residual=rd if res else None, act_out=ao)
        got = y.cpu().numpy().transpose(0, 3, 1, 2)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-6
        if act_out:
            assert np.linalg.norm(ao.cpu().numpy().transpose(0, 3, 1, 2) - a) / np.linalg.norm(a) < 2e-6
    finally:
        ops.set_spectral_mode(prev_m); ops.set_spectral_transform(prev_x); ops.set_spectral_tile(prev_t)


def test_layer_results_of_the_two_transforms_agree():
    """Same layer, same inputs, both kernel families (forward, weight gradient, fused backward with the POST epilogue of the data gradient)."""
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(3)
    N, H, W, Cin, Cout, k = 2, 96, 110, 32, 32, 7
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
    w = torch.randn(k, k, Cin, Cout, device='cuda', generator=g) / (k * Cin ** 0.5)
    dz = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    prev_m, prev_x, prev_t = ops.get_spectral_mode(), ops.get_spectral_transform(), ops.get_spectral_tile()
    ops.set_spectral_mode('force'); ops.set_spectral_tile(32)
    res = {}
    try:
        for xf in ('mfma', 'fft'):
            ops.set_spectral_transform(xf)
            y = ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2)
            dw = ops.conv2d_wgrad(x, dz, tuple(w.shape), pad_top=k // 2, pad_left=k // 2)
            dx = ops.conv2d_fwd(dz, ops.flip_transpose_weights(w), None, pad_top=k // 2, pad_left=k // 2)
            res[xf] = [t.double() for t in (y, dw, dx)]
    finally:
        ops.set_spectral_mode(prev_m); ops.set_spectral_transform(prev_x); ops.set_spectral_tile(prev_t)
    for a, b in zip(res['fft'], res['mfma']):
        assert 0 <= float((a - b).norm() / b.norm()) < 1e-6


def _layout64(X):
    """The 4096 spectrum rows of one 64 x 64 window (csrc/spectral_common.h) from its full complex 2-D DFT X[fy][fx][c]."""
    C = X.shape[2]
    out = np.zeros((4096, C))
    for base, fx in ((0, 0), (64, 32)):                                  # the two real columns: half-complex along y
        out[base:base + 33] = X[0:33, fx].real
        out[base + 33:base + 64] = X[1:32, fx].imag
    for fx in range(1, 32):
        out[128 + 128 * (fx - 1):128 + 128 * (fx - 1) + 64] = X[:, fx].real
        out[128 + 128 * (fx - 1) + 64:128 + 128 * fx] = X[:, fx].imag
    return out


@pytest.mark.parametrize('ylim,xlim,C', [(64, 64, 32), (64, 64, 20), (50, 50, 32), (37, 64, 7), (15, 15, 64), (64, 64, 48)])
def test_forward_spectra_64_every_row_against_numpy_and_the_matrix_core_form(ylim, xlim, C):
    """The 64-point forward transform as in-register FFTs (fft64_fwd_kernel: 16-channel items, two x-parity phases through LDS, radix-4 step on the y
    axis) writes, for ONE window, every one of the 4 096 spectrum rows that numpy's FFT predicts (layout: csrc/spectral_common.h) and agrees with the
    matrix-core kernels row by row - full and masked windows, ragged channel groups (the 16-channel halves of a 20- / 7- / 48-channel image)."""
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.ops import _p
    g = torch.Generator(device='cuda').manual_seed(ylim + C)
    x = torch.randn(64, 64, C, device='cuda', generator=g)
    groups = (C + 31) // 32

    def spectrum(xf):
        prev = ops.get_spectral_transform()
        ops.set_spectral_transform(xf)
        try:
            out = torch.full((groups * 4096, 32), float('nan'), device='cuda')
            ops.handle().call('pcnn_debug_tile_spectrum64', 64, 64, C, _p(x), ylim, xlim, _p(out))
            torch.cuda.synchronize()
            return out.cpu().numpy().astype(np.float64)
        finally:
            ops.set_spectral_transform(prev)
    ref, got = spectrum('mfma'), spectrum('fft')
    xm = x.cpu().numpy().astype(np.float64).copy()
    xm[ylim:, :, :] = 0
    xm[:, xlim:, :] = 0
    X = np.fft.fft2(xm, axes=(0, 1))
    scale = np.abs(X).max()
    for gi in range(groups):
        c0, c1 = 32 * gi, min(C, 32 * gi + 32)
        want = _layout64(X[:, :, c0:c1])
        blk = got[4096 * gi:4096 * (gi + 1)]
        assert np.isfinite(blk).all()
        assert np.abs(blk[:, :c1 - c0] - want).max() < 2e-6 * scale
        assert c1 - c0 == 32 or np.abs(blk[:, c1 - c0:]).max() == 0.0    # channels that do not exist: zero spectrum rows
        assert np.abs(blk - ref[4096 * gi:4096 * (gi + 1)]).max() < 2e-6 * scale


@pytest.mark.parametrize('k,Cin,Cout,H,W,mode', [(15, 32, 32, 150, 131, 'CONSTANT'), (13, 28, 28, 75, 140, 'SYMMETRIC'), (11, 16, 32, 128, 128, 'REFLECT'), (9, 24, 40, 70, 201, 'CONSTANT')])
def test_layers_on_64_point_tiles_through_the_fft_forward_transform(k, Cin, Cout, H, W, mode):
    """Whole layers on 64-point tiles with the FFT forward transform (boundary and interior windows, several tiles, N > 1): forward, weight gradient and
    fused backward against torch-CPU fp64 at the tolerances of the matrix-core route."""
    import torch.nn.functional as F
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(k + Cin)
    N, p = 2, k // 2
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    tmode = {'CONSTANT': 'constant', 'SYMMETRIC': 'symmetric', 'REFLECT': 'reflect'}[mode]
    xp = torch.tensor(np.pad(x.astype(np.float64), ((0, 0), (0, 0), (p, p), (p, p)), mode=tmode), requires_grad=True)
    y = F.conv2d(xp, wt.permute(3, 2, 0, 1))
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), device='cuda')
    dzd = torch.tensor(np.ascontiguousarray(dz.transpose(0, 2, 3, 1)), device='cuda')
    wd = torch.tensor(w, device='cuda')
    prev_m, prev_x, prev_t = ops.get_spectral_mode(), ops.get_spectral_transform(), ops.get_spectral_tile()
    ops.set_spectral_mode('force'); ops.set_spectral_transform('fft'); ops.set_spectral_tile(64)
    try:
        got = ops.conv2d_fwd(xd, wd, None, pad_top=p, pad_left=p, pad_mode=mode).cpu().numpy().transpose(0, 3, 1, 2)
        ref = y.detach().numpy()
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-6
        if Cout <= 32:
            dw = ops.conv2d_wgrad(xd, dzd, w.shape, pad_top=p, pad_left=p, pad_mode=mode).cpu().numpy()
            assert np.linalg.norm(dw - wt.grad.numpy()) / np.linalg.norm(wt.grad.numpy()) < 5e-6
    finally:
        ops.set_spectral_mode(prev_m); ops.set_spectral_transform(prev_x); ops.set_spectral_tile(prev_t)
