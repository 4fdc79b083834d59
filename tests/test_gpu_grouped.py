"""Per-sample-filter ("metalearning") convolutions as ONE launch per batch (csrc/grouped_conv.hip) against the per-sample launches of the
ordinary kernels they replace - forward, data gradient (filter read flipped and transposed in the kernel), filter and bias gradients, all
padding modes, even and odd filter sizes, 1-D layers (kh = 1), transposed convolutions - and the launch count of the metalearning layers,
which must not depend on the batch size (layers/metalearning_conv.py:18,30 serialise the samples with tf.map_fn)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.fixture(params=['mfma', 'valu'])
def route(request):
    """Both implementations of the per-sample-filter convolution: the matrix-core grouped implicit GEMM (the default wherever the shape
    allows it) and the vector-ALU kernels (PCNN_GROUPED_VALU=1; the library reads the variable at every call)."""
    import os
    if request.param == 'valu':
        os.environ['PCNN_GROUPED_VALU'] = '1'
    yield request.param
    os.environ.pop('PCNN_GROUPED_VALU', None)


# (k, Cin, Cout, H, W, padding): the reference's example layer (19 x 19, 3 -> 4, layers/metalearning_conv.py:171-184), the layer shapes of
# configs.hpnn_metalearning(), odd / even / padded channel counts (Cin = 6 -> 8 lanes, Cout = 6 -> two register quads), ragged tile edges,
# more than 64 K steps per filter row (two B registers in the weight gradient), even filter sizes, a 1-row image
MFMA_SHAPES = [(19, 3, 4, 70, 93, 'CONSTANT'), (17, 4, 6, 50, 66, 'SYMMETRIC'), (15, 6, 8, 33, 130, 'REFLECT'), (13, 8, 8, 64, 64, 'CONSTANT'),
               (7, 8, 6, 41, 29, 'SYMMETRIC'), (5, 1, 3, 23, 71, 'CONSTANT'), (4, 2, 5, 37, 65, 'REFLECT'), (3, 16, 7, 9, 200, 'CONSTANT'),
               (11, 5, 1, 30, 30, 'SYMMETRIC'), (2, 7, 8, 1, 77, 'CONSTANT'), (11, 16, 8, 40, 70, 'CONSTANT'), (11, 8, 16, 33, 47, 'SYMMETRIC'),
               (3, 8, 16, 20, 90, 'REFLECT'), (5, 4, 12, 31, 64, 'CONSTANT')]


@pytest.mark.parametrize('k,Cin,Cout,H,W,mode', MFMA_SHAPES)
def test_grouped_conv_mfma_route_matches_the_oracle(k, Cin, Cout, H, W, mode):
    """The matrix-core route (forward, data gradient through the flipped / transposed pack, filter gradient) against fp64 torch autograd of
    the oracle's padded convolution, sample by sample."""
    from oracle import torch_twin as T
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(1000 * k + 10 * Cin + Cout)
    N = 3
    nk = k * k * Cin * Cout
    wshape = (k, k, Cin, Cout)
    pt, pb = k // 2, k // 2 - (1 - k % 2)
    assert ops.grouped_uses_mfma((N, H, W, Cin), wshape, (H, W), 'fwd')
    assert ops.grouped_uses_mfma((N, H, W, Cin), wshape, (H, W), 'wgrad') or (k, Cin, Cout) == (11, 8, 16)      # 3 x 2 x 4 accumulator quads: vector-ALU filter gradient
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
    kb = torch.randn(N, nk + Cout, device='cuda', generator=g) / np.sqrt(k * k * Cin)
    dz = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    y = ops.grouped_conv2d_fwd(x, kb, wshape, kb[:, nk:], pad_top=pt, pad_left=pt, out_hw=(H, W), pad_mode=mode, pad_value=0.25, act='tanh')
    dkb = torch.zeros_like(kb)
    ops.grouped_conv2d_wgrad(x, dz, wshape, dkb, pad_top=pt, pad_left=pt, pad_mode=mode, pad_value=0.25)
    # data gradient of the LINEAR layer w.r.t. the padded input (the layers fold the padding afterwards): full correlation with the flipped filter
    gp = ops.grouped_conv2d_fwd(dz, kb, wshape, None, pad_top=k - 1, pad_left=k - 1, out_hw=(H + k - 1, W + k - 1), flip_transpose=True)
    xt = x.double().cpu().permute(0, 3, 1, 2)
    kt = kb.double().cpu().requires_grad_(True)
    dzt = dz.double().cpu().permute(0, 3, 1, 2)
    for n in range(N):
        xp = T.pad2d(xt[n:n + 1], ((pt, pb), (pt, pb)), mode, 0.25).requires_grad_(True)
        z = T.conv2d_valid(xp, kt[n, :nk].reshape(*wshape), kt[n, nk:])
        (z * dzt[n:n + 1]).sum().backward()
        assert rel(y[n].cpu().permute(2, 0, 1), torch.tanh(z.detach())[0]) < 2e-6
        assert rel(gp[n].cpu().permute(2, 0, 1), xp.grad[0]) < 2e-6
    assert rel(dkb.cpu()[:, :nk], kt.grad[:, :nk]) < 5e-6


@pytest.mark.parametrize('k,Cin,Cout,H,W,mode,act', [(5, 3, 4, 40, 37, 'CONSTANT', 'leaky_relu'), (19, 3, 4, 50, 61, 'SYMMETRIC', 'tanh'), (4, 6, 8, 33, 45, 'REFLECT', 'linear'),
                                                      (13, 8, 8, 64, 64, 'CONSTANT', 'leaky_relu'), (3, 20, 17, 30, 70, 'SYMMETRIC', 'linear'), (7, 5, 32, 41, 29, 'CONSTANT', 'tanh')])
def test_grouped_conv_matches_per_sample_launches(k, Cin, Cout, H, W, mode, act, route):
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(k * 100 + Cin)
    N = 5
    nk = k * k * Cin * Cout
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
    kb = torch.randn(N, nk + Cout, device='cuda', generator=g) / np.sqrt(k * k * Cin)
    pt, pb = k // 2, k // 2 - (1 - k % 2)
    Ho, Wo = H + pt + pb - k + 1, W + pt + pb - k + 1
    wshape = (k, k, Cin, Cout)
    y = ops.grouped_conv2d_fwd(x, kb, wshape, kb[:, nk:], pad_top=pt, pad_left=pt, out_hw=(Ho, Wo), pad_mode=mode, pad_value=0.25, act=act)
    dz = torch.randn(N, Ho, Wo, Cout, device='cuda', generator=g)
    dkb = torch.zeros_like(kb)
    ops.grouped_conv2d_wgrad(x, dz, wshape, dkb, pad_top=pt, pad_left=pt, pad_mode=mode, pad_value=0.25)
    ops.grouped_bias_grad(dz, dkb[:, nk:])
    dpt = (k - 1 - pt) if mode == 'CONSTANT' else k - 1
    ohw = (H, W) if mode == 'CONSTANT' else (H + pt + pb, W + pt + pb)
    gx = ops.grouped_conv2d_fwd(dz, kb, wshape, None, pad_top=dpt, pad_left=dpt, out_hw=ohw, flip_transpose=True)
    ops.set_spectral_mode('off')                  # the per-sample reference: the direct kernels
    try:
        for n in range(N):
            w = kb[n, :nk].view(*wshape).contiguous()
            yn = ops.conv2d_fwd(x[n:n + 1], w, kb[n, nk:].contiguous(), pad_top=pt, pad_left=pt, out_hw=(Ho, Wo), pad_mode=mode, pad_value=0.25, act=act)
            assert rel(y[n:n + 1], yn) < 2e-6
            dwn = ops.conv2d_wgrad(x[n:n + 1], dz[n:n + 1], wshape, pad_top=pt, pad_left=pt, pad_mode=mode, pad_value=0.25)
            assert rel(dkb[n, :nk].view(*wshape), dwn) < 5e-6
            assert rel(dkb[n, nk:], dz[n].sum(dim=(0, 1))) < 5e-6
            gn = ops.conv2d_fwd(dz[n:n + 1], ops.flip_transpose_weights(w), None, pad_top=dpt, pad_left=dpt, out_hw=ohw)
            assert rel(gx[n:n + 1], gn) < 2e-6
    finally:
        ops.set_spectral_mode('auto')


@pytest.mark.parametrize('k,Cin,Cout,H,W', [(3, 6, 40, 20, 30), (5, 40, 6, 17, 33), (7, 32, 32, 40, 40)])
def test_wide_per_sample_layers_leave_the_grouped_kernels(k, Cin, Cout, H, W):
    """channel counts beyond the grouped kernels' limits (more than 32 output channels - in the data-gradient call that is the
    forward layer's INPUT width) must not fail, and wide layers take the ordinary kernels, one launch per sample: forward, data gradient
    and filter gradient against fp64 autograd."""
    import torch.nn.functional as F
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(k + Cin)
    N = 3
    nk = k * k * Cin * Cout
    wshape = (k, k, Cin, Cout)
    pt = k // 2
    x = torch.randn(N, H, W, Cin, device='cuda', generator=g)
    kb = torch.randn(N, nk + Cout, device='cuda', generator=g) / np.sqrt(k * k * Cin)
    dz = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    y = ops.grouped_conv2d_fwd(x, kb, wshape, kb[:, nk:], pad_top=pt, pad_left=pt, out_hw=(H, W), act='leaky_relu')
    dkb = torch.zeros_like(kb)
    ops.grouped_conv2d_wgrad(x, dz, wshape, dkb, pad_top=pt, pad_left=pt)
    gx = ops.grouped_conv2d_fwd(dz, kb, wshape, None, pad_top=k - 1 - pt, pad_left=k - 1 - pt, out_hw=(H, W), flip_transpose=True)
    xt = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    kt = kb.double().cpu().requires_grad_(True)
    z = torch.cat([F.conv2d(xt[n:n + 1], kt[n, :nk].reshape(*wshape).permute(3, 2, 0, 1), kt[n, nk:], padding=pt) for n in range(N)], 0)
    (z * dz.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert rel(y.cpu().permute(0, 3, 1, 2), F.leaky_relu(z.detach(), 0.2)) < 2e-6
    assert rel(gx.cpu().permute(0, 3, 1, 2), xt.grad) < 2e-6
    assert rel(dkb.cpu()[:, :nk], kt.grad[:, :nk]) < 5e-6


def test_grouped_one_dimensional_layer():
    """A 1-D convolution is kh = 1: what layers/metalearning_conv.py does with tf.nn.conv1d (dimensions = 1)."""
    import torch.nn.functional as F
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(9)
    N, L, Cin, Cout, k = 4, 150, 3, 6, 11
    x = torch.randn(N, 1, L, Cin, device='cuda', generator=g)
    kb = torch.randn(N, k * Cin * Cout + Cout, device='cuda', generator=g) * 0.2
    nk = k * Cin * Cout
    y = ops.grouped_conv2d_fwd(x, kb, (1, k, Cin, Cout), kb[:, nk:], pad_top=0, pad_left=k // 2, out_hw=(1, L), pad_mode='CONSTANT', act='linear')
    for n in range(N):
        w = kb[n, :nk].view(k, Cin, Cout).permute(2, 1, 0).double().cpu()          # (Cout, Cin, k)
        ref = F.conv1d(x[n, 0].t()[None].double().cpu(), w, kb[n, nk:].double().cpu(), padding=k // 2)[0].t()
        assert rel(y[n, 0].cpu(), ref) < 2e-6


# the last three shapes have a SAME crop offset (ceil(H / f) f - H) // 2 >= 1 in both axes (every shape used to have offset 0)
DECONV = [(2, 5, 4, 20, 31, 40, 62), (3, 8, 8, 14, 10, 41, 29), (4, 3, 6, 9, 9, 36, 36), (3, 5, 4, 8, 9, 22, 25), (4, 6, 3, 3, 4, 10, 13), (8, 32, 32, 32, 32, 250, 250)]


@pytest.mark.parametrize('f,Cin,Cout,hc,wc,H,W', DECONV)
def test_grouped_deconv_matches_the_oracle(f, Cin, Cout, hc, wc, H, W):
    """pcnn_grouped_deconv_fwd / _bwd_data / _bwd_filter against the fp64 oracle (oracle/torch_twin.conv2d_transpose_same + autograd) directly."""
    from oracle import torch_twin as T
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(f + 17)
    N = 3
    nk = f * f * Cout * Cin
    kshape = (f, f, Cout, Cin)
    x = torch.randn(N, hc, wc, Cin, device='cuda', generator=g)
    kb = torch.randn(N, nk + Cout, device='cuda', generator=g) * 0.3
    dy = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    y = ops.grouped_deconv_fwd(x, kb, kshape, kb[:, nk:], (H, W), f)
    dkb = torch.zeros_like(kb)
    ops.grouped_deconv_bwd_filter(x, dy, f, dkb, dkb[:, nk:])
    dx = ops.grouped_deconv_bwd_data(dy, kb, kshape, (hc, wc), f)
    xt = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    kt = kb.double().cpu().requires_grad_(True)
    yt = torch.cat([T.conv2d_transpose_same(xt[n:n + 1], kt[n, :nk].reshape(*kshape), kt[n, nk:], (H, W), f) for n in range(N)], 0)
    (yt * dy.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    assert rel(y.cpu().permute(0, 3, 1, 2), yt.detach()) < 2e-6
    assert rel(dx.cpu().permute(0, 3, 1, 2), xt.grad) < 2e-6
    assert rel(dkb.cpu()[:, :nk], kt.grad[:, :nk]) < 5e-6 and rel(dkb.cpu()[:, nk:], kt.grad[:, nk:]) < 5e-6


@pytest.mark.parametrize('f,Cin,Cout,hc,wc,H,W', DECONV)
def test_grouped_deconv_matches_per_sample_launches(f, Cin, Cout, hc, wc, H, W):
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(f)
    N = 4
    nk = f * f * Cout * Cin
    kshape = (f, f, Cout, Cin)
    x = torch.randn(N, hc, wc, Cin, device='cuda', generator=g)
    kb = torch.randn(N, nk + Cout, device='cuda', generator=g) * 0.3
    y = ops.grouped_deconv_fwd(x, kb, kshape, kb[:, nk:], (H, W), f)
    dy = torch.randn(N, H, W, Cout, device='cuda', generator=g)
    dkb = torch.zeros_like(kb)
    ops.grouped_deconv_bwd_filter(x, dy, f, dkb, dkb[:, nk:])
    dx = ops.grouped_deconv_bwd_data(dy, kb, kshape, (hc, wc), f)
    for n in range(N):
        kn = kb[n, :nk].view(*kshape).contiguous()
        assert rel(y[n:n + 1], ops.deconv_fwd(x[n:n + 1].contiguous(), kn, kb[n, nk:].contiguous(), (H, W), f)) < 2e-6
        dk = torch.zeros_like(kn)
        db = torch.zeros(Cout, device='cuda')
        ops.deconv_bwd_filter(x[n:n + 1].contiguous(), dy[n:n + 1].contiguous(), f, dk=dk, dbias=db)
        assert rel(dkb[n, :nk].view(*kshape), dk) < 5e-6 and rel(dkb[n, nk:], db) < 5e-6
        assert rel(dx[n:n + 1], ops.deconv_bwd_data(dy[n:n + 1].contiguous(), kn, (hc, wc), f)) < 2e-6


def test_metalearning_layer_launch_count_is_independent_of_batch_size():
    """Forward + backward of metalearning_conv and metalearning_deconvupscale: the number of library calls is the same for 2 and for 7 samples."""
    from poisson_cnn_amd import metalearning as M, _lib

    def count(N):
        conv = M.metalearning_conv(5, 5, padding='same', padding_mode='SYMMETRIC', conv_activation='tf.nn.leaky_relu', use_bias=True, pre_output_dense_units=(6, 8),
                                   previous_layer_filters=3, dense_input_features=4, seed=1)
        dec = M.metalearning_deconvupscale(2, 4, 2, use_bias=True, pre_output_dense_units=(6, 8), previous_layer_filters=5, dense_input_features=4, seed=2)
        g = torch.Generator(device='cuda').manual_seed(N)
        x = torch.randn(N, 24, 20, 3, device='cuda', generator=g)
        d = torch.randn(N, 4, device='cuda', generator=g)
        calls = []
        orig = _lib.Handle.call

        def spy(self, name, *a):
            calls.append(name)
            return orig(self, name, *a)
        _lib.Handle.call = spy
        try:
            y = conv.forward(x, d, training=True)
            dx, dd = conv.backward(torch.ones_like(y))
            z = dec.forward(y, d, (48, 40), training=True)
            dec.backward(torch.ones_like(z))
        finally:
            _lib.Handle.call = orig
        assert any(c.startswith('pcnn_grouped_conv2d_fwd') for c in calls) and 'pcnn_grouped_deconv_fwd' in calls
        return len(calls)
    assert count(2) == count(7)
