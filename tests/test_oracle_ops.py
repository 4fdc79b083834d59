"""The numpy oracle's ops against independent PyTorch-CPU implementations (where semantics coincide),
hand-derived small cases (where they do not), and the autograd twin against the numpy oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ops, torch_twin, hpnn, loss as oloss
from poisson_cnn_amd import configs


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize('k,mode', [(3, 'CONSTANT'), (5, 'SYMMETRIC'), (7, 'REFLECT'), (4, 'CONSTANT'), (15, 'SYMMETRIC')])
def test_padded_conv_vs_torch(k, mode):
    rng = np.random.default_rng(k)
    x = rng.standard_normal((2, 3, 17, 19))
    w = rng.standard_normal((k, k, 3, 5))
    b = rng.standard_normal(5)
    got = np_ops.padded_conv2d(x, w, b, mode, 0.25, 'tf.nn.leaky_relu')
    # independent: build the padded tensor by hand-written index maps, conv with F.conv2d
    pb, pa = k // 2, k // 2 - (1 - k % 2)
    def idx(n):
        i = np.arange(-pb, n + pa)
        if mode == 'SYMMETRIC':
            i = np.where(i < 0, -i - 1, np.where(i >= n, 2 * n - 1 - i, i))
        elif mode == 'REFLECT':
            i = np.where(i < 0, -i, np.where(i >= n, 2 * n - 2 - i, i))
        return i
    if mode == 'CONSTANT':
        xp = np.full((2, 3, 17 + pb + pa, 19 + pb + pa), 0.25)
        xp[:, :, pb:pb + 17, pb:pb + 19] = x
    else:
        xp = x[:, :, idx(17)][:, :, :, idx(19)]
    ref = F.conv2d(torch.tensor(xp), torch.tensor(w).permute(3, 2, 0, 1), torch.tensor(b))
    ref = F.leaky_relu(ref, 0.2).numpy()
    assert got.shape == (2, 5, 17, 19)
    assert rel(got, ref) < 1e-13
    tw = torch_twin.padded_conv2d(torch.tensor(x), w, b, mode, 0.25, 'tf.nn.leaky_relu').numpy()
    assert rel(tw, got) < 1e-13


def test_same_conv_vs_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 9, 11)); w = rng.standard_normal((5, 5, 4, 3)); b = rng.standard_normal(3)
    ref = F.conv2d(torch.tensor(x), torch.tensor(w).permute(3, 2, 0, 1), torch.tensor(b), padding=2).numpy()
    assert rel(np_ops.same_conv2d(x, w, b), ref) < 1e-13


@pytest.mark.parametrize('H,W,f', [(12, 8, 2), (13, 10, 3), (7, 9, 4), (5, 6, 8), (33, 17, 16)])
def test_pool_same(H, W, f):
    rng = np.random.default_rng(f)
    x = rng.standard_normal((2, 3, H, W))
    for kind in ('average', 'max'):
        got = np_ops.pool2d_same(x, f, kind)
        Ho, Wo = -(-H // f), -(-W // f)
        assert got.shape == (2, 3, Ho, Wo)
        # independent: pad to Ho*f with NaN using TF's SAME split, nan-aware reduce
        pty, ptx = Ho * f - H, Wo * f - W
        xp = np.full((2, 3, Ho * f, Wo * f), np.nan)
        xp[:, :, pty // 2:pty // 2 + H, ptx // 2:ptx // 2 + W] = x
        blocks = xp.reshape(2, 3, Ho, f, Wo, f)
        ref = np.nanmean(blocks, axis=(3, 5)) if kind == 'average' else np.nanmax(blocks, axis=(3, 5))
        assert rel(got, ref) < 1e-13
        assert rel(torch_twin.pool2d_same(torch.tensor(x), f, kind).numpy(), got) < 1e-13
    if H % f == 0 and W % f == 0:
        assert rel(np_ops.pool2d_same(x, f, 'average'), F.avg_pool2d(torch.tensor(x), f).numpy()) < 1e-13


@pytest.mark.parametrize('H,W,f', [(12, 8, 2), (13, 10, 3), (7, 9, 4), (33, 17, 16)])
def test_conv_transpose_same(H, W, f):
    rng = np.random.default_rng(f)
    h, w_ = -(-H // f), -(-W // f)
    x = rng.standard_normal((2, 3, h, w_)); k = rng.standard_normal((f, f, 4, 3)); b = rng.standard_normal(4)
    got = np_ops.conv2d_transpose_same(x, k, b, (H, W), f)
    # independent definition: adjoint of the SAME strided forward conv (dot-product test)
    y = rng.standard_normal((2, 4, H, W))
    pby, pbx = (h * f - H) // 2, (w_ * f - W) // 2
    yp = np.zeros((2, 4, h * f, w_ * f)); yp[:, :, pby:pby + H, pbx:pbx + W] = y
    fwd = F.conv2d(torch.tensor(yp), torch.tensor(k).permute(3, 2, 0, 1), stride=f).numpy()   # (2,3,h,w): conv with HWIO = (f,f,4,3)
    assert abs((got - b[None, :, None, None]) .ravel() @ y.ravel() - fwd.ravel() @ x.ravel()) < 1e-9 * abs(fwd.ravel() @ x.ravel()) + 1e-9
    # pixel-shuffle closed form (k == f): out[i] = x[(i+pb)//f] k[(i+pb)%f]
    for (i, j) in [(0, 0), (H - 1, W - 1), (H // 2, W // 3)]:
        oi, ti, oj, tj = (i + pby) // f, (i + pby) % f, (j + pbx) // f, (j + pbx) % f
        ref = np.einsum('nc,oc->no', x[:, :, oi, oj], k[ti, tj]) + b
        assert np.allclose(got[:, :, i, j], ref, rtol=1e-12, atol=1e-12)
    assert rel(torch_twin.conv2d_transpose_same(torch.tensor(x), k, b, (H, W), f).numpy(), got) < 1e-13


def test_resize_bilinear_nearest_vs_torch():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, 5, 7))
    for out in [(20, 28), (17, 23), (40, 35)]:
        ref = F.interpolate(torch.tensor(x), size=out, mode='bilinear', align_corners=False).numpy()
        assert rel(np_ops.resize2d(x, out, 'bilinear'), ref) < 2e-6   # weights are float32 in TF
        ref = F.interpolate(torch.tensor(x), size=out, mode='nearest-exact').numpy()
        assert rel(np_ops.resize2d(x, out, 'nearest'), ref) == 0.0


def test_resize_bicubic_properties():
    # Keys a=-0.5 half-pixel: rows sum to 1 (renormalised at the borders), reproduces constants and,
    # away from the borders, linear ramps; interior weights match torch's bicubic with a=-0.5? torch uses -0.75,
    # so compare against a direct evaluation of the Keys kernel instead.
    n_in, n_out = 9, 31
    M = np_ops.resize_matrix(n_in, n_out, 'bicubic')
    assert np.allclose(M.sum(1), 1.0, atol=1e-6)
    def keys(t, a=-0.5):
        t = abs(t)
        return ((a + 2) * t - (a + 3)) * t * t + 1 if t <= 1 else (((t - 5) * t + 8) * t - 4) * a if t < 2 else 0.0
    scale = n_in / n_out
    for o in range(n_out):
        src = (o + 0.5) * scale - 0.5
        loc = int(np.floor(src))
        if loc - 1 < 0 or loc + 2 > n_in - 1:
            continue
        # table lookup quantises delta to 1/1024
        delta = round((src - loc) * 1024) / 1024
        ref = np.zeros(n_in)
        for t, i in zip([1 + delta, delta, 1 - delta, 2 - delta], [loc - 1, loc, loc + 1, loc + 2]):
            ref[i] += keys(t)
        assert np.allclose(M[o], ref / ref.sum(), atol=2e-6)
    # legacy align_corners bicubic (a=-0.75) hits the control points exactly
    M2 = np_ops.resize_matrix(5, 17, 'bicubic', half_pixel=False, align_corners=True)
    for i in range(5):
        assert np.allclose(M2[4 * i], np.eye(5)[i], atol=1e-6)
    ref = F.interpolate(torch.arange(5.0, dtype=torch.float64)[None, None, None].expand(1, 1, 5, 5).contiguous(), size=(17, 17),
                        mode='bicubic', align_corners=True).numpy()[0, 0, 0]
    assert np.allclose(M2 @ np.arange(5.0), ref, atol=1e-5)


def test_bicubic_and_bilinear_vs_pil():
    """Third-party cross-check of the ASSUMED tf.image.resize (half-pixel, antialias=False) semantics (VERDICT r4 missing #2): Pillow's up-sampling
    filters are independent code for the same definition - Keys a = -0.5 cubic / triangle, half-pixel centres, out-of-range taps dropped and the
    weights renormalised.  Dyadic ratios (every up-sampling factor of hpnn.json: the branch outputs are 8 -> 1024 ... 512 -> 1024) agree to float32
    rounding; at a non-dyadic ratio TensorFlow's 1024-entry coefficient table quantises the phase (|delta error| <= 2^-11 -> ~7e-4 in the values),
    which Pillow's exact evaluation does not share - hence the looser bound there."""
    from PIL import Image
    rng = np.random.default_rng(0)
    for n_in, n_out, tol in [(8, 256, 1e-6), (16, 512, 1e-6), (32, 1024, 1e-6), (64, 128, 1e-6), (11, 350, 2e-3), (9, 31, 2e-3)]:
        x = rng.standard_normal((n_in, n_in + 3)).astype(np.float32)
        My, Mx = np_ops.resize_matrix(n_in, n_out, 'bicubic'), np_ops.resize_matrix(n_in + 3, 2 * n_out, 'bicubic')
        pil = np.asarray(Image.fromarray(x, mode='F').resize((2 * n_out, n_out), Image.BICUBIC), dtype=np.float64)      # PIL size = (width, height)
        assert rel(My @ x.astype(np.float64) @ Mx.T, pil) < (tol if (2 * n_out) % (n_in + 3) == 0 else 2e-3), (n_in, n_out)
        Ms = np_ops.resize_matrix(n_in, n_out, 'bicubic')
        xs = rng.standard_normal((n_in, n_in)).astype(np.float32)
        pil = np.asarray(Image.fromarray(xs, mode='F').resize((n_out, n_out), Image.BICUBIC), dtype=np.float64)
        assert rel(Ms @ xs.astype(np.float64) @ Ms.T, pil) < tol, (n_in, n_out)
        Mb = np_ops.resize_matrix(n_in, n_out, 'bilinear')
        pil = np.asarray(Image.fromarray(xs, mode='F').resize((n_out, n_out), Image.BILINEAR), dtype=np.float64)
        assert rel(Mb @ xs.astype(np.float64) @ Mb.T, pil) < 1e-6, (n_in, n_out)
    # through the public op (NCHW), one dyadic case
    x4 = rng.standard_normal((1, 1, 16, 16)).astype(np.float32)
    pil = np.asarray(Image.fromarray(x4[0, 0], mode='F').resize((128, 128), Image.BICUBIC), dtype=np.float64)
    assert rel(np_ops.resize2d(x4.astype(np.float64), (128, 128), 'bicubic')[0, 0], pil) < 1e-6


def test_padded_conv_vs_scipy_ndimage():
    """Third-party cross-check of tf.pad SYMMETRIC / REFLECT / CONSTANT + VALID correlation (H1 / H2): scipy.ndimage.correlate implements the same
    three boundary rules under its own names (half-sample symmetric = 'reflect', whole-sample symmetric = 'mirror', 'constant') in independent C code."""
    from scipy import ndimage
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 1, 13, 17))
    for k in (3, 5, 7):
        w = rng.standard_normal((k, k, 1, 1))
        for tf_mode, sp_mode in (('SYMMETRIC', 'reflect'), ('REFLECT', 'mirror'), ('CONSTANT', 'constant')):
            got = np_ops.padded_conv2d(x, w, None, tf_mode, 0.0, 'linear')
            ref = ndimage.correlate(x[0, 0], w[:, :, 0, 0], mode=sp_mode, cval=0.0)
            assert got.shape == x.shape and rel(got[0, 0], ref) < 1e-13, (k, tf_mode)


def test_spp_and_split_indices():
    assert list(np_ops.split_indices(229, 4)) == [0, 58, 115, 172, 229]      # dataset/utils/split_indices.py:13
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 4, 11, 13))
    f = np_ops.spatial_pyramid_pool(x, [[2, 2], 3, 5], 'max')
    assert f.shape == (3, 38)
    assert np.isclose(f[1, 0], x[1, :, :6, :7].max())
    assert np.isclose(f[2, 3], x[2, :, 6:, 7:].max())
    assert rel(torch_twin.spatial_pyramid_pool(torch.tensor(x), [[2, 2], 3, 5], 'max').numpy(), f) == 0


def test_bn_dense_ring_jacobi_twin_agree():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 4, 9, 8))
    g, b, m, v = rng.uniform(0.5, 1.5, 4), rng.standard_normal(4), rng.standard_normal(4), rng.uniform(0.5, 2, 4)
    got = np_ops.batchnorm_inference(x, g, b, m, v)
    ref = F.batch_norm(torch.tensor(x), torch.tensor(m), torch.tensor(v), torch.tensor(g), torch.tensor(b), False, 0.0, 1e-3).numpy()
    assert rel(got, ref) < 1e-13
    tr = np_ops.batchnorm_training(x, g, b)[0]
    ref = F.batch_norm(torch.tensor(x), None, None, torch.tensor(g), torch.tensor(b), True, 0.0, 1e-3).numpy()
    assert rel(tr, ref) < 1e-13
    for mode in ('CONSTANT', 'SYMMETRIC'):
        r = np_ops.bc_ring(x, mode)
        assert np.array_equal(r[:, :, 1:-1, 1:-1], x[:, :, 1:-1, 1:-1])
        if mode == 'CONSTANT':
            assert np.all(r[:, :, 0] == 0) and np.all(r[:, :, :, -1] == 0)
        else:
            assert np.array_equal(r[:, :, 0, 1:-1], x[:, :, 1, 1:-1]) and r[0, 0, 0, 0] == x[0, 0, 1, 1]
        assert rel(torch_twin.bc_ring(torch.tensor(x), mode).numpy(), r) == 0
    dx = rng.uniform(0.01, 0.05, (2, 2))
    j = np_ops.jacobi_iterations(x[:, :1], x[:, 1:2], dx, 3)
    jt = torch_twin.jacobi_iterations(torch.tensor(x[:, :1]), x[:, 1:2], dx, 3).numpy()
    assert rel(jt, j) < 1e-13
    # a Jacobi fixed point: the discrete solution of lap(u) = f stays put
    u = rng.standard_normal((1, 1, 7, 7)); d = np.array([[0.1, 0.1]])
    lap = (u[:, :, 2:, 1:-1] + u[:, :, :-2, 1:-1] + u[:, :, 1:-1, 2:] + u[:, :, 1:-1, :-2] - 4 * u[:, :, 1:-1, 1:-1]) / 0.01
    f = np.zeros_like(u); f[:, :, 1:-1, 1:-1] = lap
    assert rel(np_ops.jacobi_iterations(u, f, d, 2), u) < 1e-12


@pytest.mark.parametrize('bc', ['dirichlet', 'neumann'])
def test_hpnn_forward_numpy_vs_twin(bc):
    cfg = configs.hpnn_tiny()['model']
    cfg['bc_type'] = bc
    p = hpnn.init_params(cfg, seed=5, gain=1.5, randomize_all=True)
    rng = np.random.default_rng(7)
    rhs = rng.uniform(-1, 1, (2, 1, 36, 40)); dx = rng.uniform(5e-3, 5e-2, (2, 1))
    y = hpnn.forward(np_ops, cfg, p, rhs, dx)
    assert y.shape == rhs.shape and np.isfinite(y).all() and np.abs(y).max() > 0
    yt = hpnn.forward(torch_twin, cfg, p, torch.tensor(rhs), torch.tensor(dx)).numpy()
    assert rel(yt, y) < 1e-12
    if bc == 'dirichlet':
        assert np.all(y[:, :, 0, :] == 0) and np.all(y[:, :, :, -1] == 0)


def test_loss_numpy_vs_twin_and_gradcheck():
    rng = np.random.default_rng(11)
    yt = rng.standard_normal((3, 1, 20, 24)); yp = rng.standard_normal((3, 1, 20, 24)); rhs = rng.standard_normal((3, 1, 20, 24))
    dx = rng.uniform(0.5, 1.0, (3, 2))   # O(1) spacings keep the PI term's magnitude FD-friendly
    lp = dict(configs.hpnn()['training']['loss_parameters'])
    lp.update(physics_informed_loss_weight=6e-4, mse_loss_weight=0.3)
    L = oloss.loss_wrapper(global_batch_size=6, **lp)
    a = L(yt, yp, rhs, dx)
    ypt = torch.tensor(yp, requires_grad=True)
    b = L(yt, ypt, torch.tensor(rhs), dx)
    assert abs(float(b.detach()) - a) < 1e-12 * abs(a)
    b.backward()
    g = ypt.grad.numpy()
    # finite-difference spot checks of the twin's gradient against the numpy oracle
    for idx in [(0, 0, 3, 4), (2, 0, 10, 23), (1, 0, 0, 0)]:
        e = np.zeros_like(yp); e[idx] = 1e-6
        fd = (L(yt, yp + e, rhs, dx) - L(yt, yp - e, rhs, dx)) / 2e-6
        assert abs(fd - g[idx]) < 1e-6 * max(1.0, abs(g[idx]))


@pytest.mark.parametrize('k,C,O,H,W,tile', [(7, 3, 4, 61, 45, 32), (15, 2, 3, 70, 90, 64), (4, 5, 2, 33, 100, 24), (3, 1, 1, 20, 20, 256)])
def test_fft_convolution_of_the_autograd_twin_equals_the_direct_one(k, C, O, H, W, tile):
    """oracle.torch_twin.conv2d_valid_fft (overlap-save, torch.fft, fp64) - the evaluation the 1024^2 training-step fixture is generated with,
    because PyTorch's fp64 CPU convolution needs 60 GB for one 15 x 15 x 32 layer at that size - against F.conv2d: values and the gradients
    with respect to input, filter and bias, ragged tile edges and even filter sizes included."""
    rng = np.random.default_rng(k)
    x = torch.tensor(rng.standard_normal((2, C, H, W)), requires_grad=True)
    w = torch.tensor(rng.standard_normal((k, k, C, O)), requires_grad=True)
    b = torch.tensor(rng.standard_normal(O), requires_grad=True)
    g = torch.tensor(rng.standard_normal((2, O, H - k + 1, W - k + 1)))
    ref = F.conv2d(x, w.permute(3, 2, 0, 1), b)
    gr = torch.autograd.grad((ref * g).sum(), [x, w, b])
    got = torch_twin.conv2d_valid_fft(x, w, b, tile)
    gg = torch.autograd.grad((got * g).sum(), [x, w, b])
    assert rel(got.detach(), ref.detach()) < 1e-13
    for a, r in zip(gg, gr):
        assert rel(a, r) < 1e-12
    # the switch: the twin's padded convolution takes the FFT evaluation above the pixel threshold and F.conv2d below it
    torch_twin.set_fft_conv(min_pixels=H * W, tile=tile)
    try:
        y1 = torch_twin.padded_conv2d(x.detach(), w.detach(), b.detach(), 'SYMMETRIC', 0.0, 'leaky_relu')
    finally:
        torch_twin.set_fft_conv(None)
    y0 = torch_twin.padded_conv2d(x.detach(), w.detach(), b.detach(), 'SYMMETRIC', 0.0, 'leaky_relu')
    assert rel(y1, y0) < 1e-13 and not torch.equal(y1, y0)
