"""The narrow-convolution kernels (csrc/conv_small.hip: <= 16 channels, 3x3 / 5x5, exact fp32 FMA on the vector ALUs, built for the HBM
roofline): forward with every epilogue option and ragged channel counts, data gradient and weight gradient vs the fp64 oracle - in BOTH
math modes (the kernels do not depend on the mode) - and that the route is really taken."""
import numpy as np
import pytest
import torch

from oracle import np_ops, torch_twin

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def nhwc(a):
    return torch.tensor(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 3, 1)), dtype=torch.float32, device='cuda')


def nchw(t):
    return t.detach().cpu().numpy().transpose(0, 3, 1, 2)


SHAPES = [(3, 8, 8), (3, 16, 12), (3, 12, 12), (3, 12, 8), (3, 8, 4), (3, 4, 1), (3, 2, 4), (3, 4, 4), (3, 3, 5), (3, 1, 16), (3, 16, 16), (3, 7, 11),
          (5, 16, 16), (5, 8, 12), (5, 3, 1), (5, 12, 16)]


@pytest.mark.parametrize('mode', ['fp32', 'split_f16'])
@pytest.mark.parametrize('k,Cin,Cout', SHAPES)
def test_forward_backward(mode, k, Cin, Cout):
    from poisson_cnn_amd import ops
    prev = ops.get_math_mode()
    ops.set_math_mode(mode)
    try:
        rng = np.random.default_rng(k * 1000 + Cin * 17 + Cout)
        N, H, W = 2, 37, 45
        pad_mode = ['CONSTANT', 'SYMMETRIC', 'REFLECT'][(Cin + Cout) % 3]
        x = rng.standard_normal((N, Cin, H, W)).astype(np.float32).astype(np.float64)
        w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32).astype(np.float64)
        b = rng.standard_normal(Cout).astype(np.float32).astype(np.float64)
        dy = rng.standard_normal((N, Cout, H, W)).astype(np.float32).astype(np.float64)
        xt, wt, bt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True), torch.tensor(b, requires_grad=True)
        yt = torch_twin.padded_conv2d(xt, wt, bt, pad_mode, 0.3, 'linear')
        (yt * torch.tensor(dy)).sum().backward()
        p = k // 2
        xd, wd = nhwc(x), torch.tensor(w, dtype=torch.float32, device='cuda')
        y = ops.conv2d_fwd(xd, wd, torch.tensor(b, dtype=torch.float32, device='cuda'), pad_top=p, pad_left=p, pad_mode=pad_mode, pad_value=0.3)
        assert rel(nchw(y), np_ops.padded_conv2d(x, w, b, pad_mode, 0.3, 'linear')) < 2e-6
        dw = ops.conv2d_wgrad(xd, nhwc(dy), w.shape, pad_top=p, pad_left=p, pad_mode=pad_mode, pad_value=0.3)
        assert rel(dw.cpu().numpy(), wt.grad.numpy()) < 5e-6
        wf = ops.flip_transpose_weights(wd)
        if pad_mode == 'CONSTANT':
            dx = ops.conv2d_fwd(nhwc(dy), wf, None, pad_top=k - 1 - p, pad_left=k - 1 - p)
        else:
            gp = ops.conv2d_fwd(nhwc(dy), wf, None, pad_top=k - 1, pad_left=k - 1, out_hw=(H + k - 1, W + k - 1))
            dx = ops.pad_fold_bwd(gp, (H, W), ((p, p), (p, p)), pad_mode)
        assert rel(nchw(dx), xt.grad.numpy()) < 2e-6
    finally:
        ops.set_math_mode(prev)


def test_epilogue_options_and_channel_slices():
    """bias + leaky-ReLU + BN affine + residual + act_out, input and output as channel slices of wider buffers (vector and scalar paths)."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(3)
    for Cin, Cout, ldx_extra, ldy_extra in ((8, 8, 8, 8), (12, 8, 4, 0), (6, 5, 1, 3)):
        N, H, W = 2, 19, 40
        xb = torch.tensor(rng.standard_normal((N, H, W, Cin + ldx_extra)), dtype=torch.float32, device='cuda')
        x = xb[..., ldx_extra:]
        w = torch.tensor(rng.standard_normal((3, 3, Cin, Cout)) / 8, dtype=torch.float32, device='cuda')
        b = torch.tensor(rng.standard_normal(Cout), dtype=torch.float32, device='cuda')
        sc = torch.tensor(rng.uniform(0.5, 1.5, Cout), dtype=torch.float32, device='cuda')
        sh = torch.tensor(rng.standard_normal(Cout), dtype=torch.float32, device='cuda')
        res = torch.tensor(rng.standard_normal((N, H, W, Cout)), dtype=torch.float32, device='cuda')
        outb = torch.zeros((N, H, W, Cout + ldy_extra), dtype=torch.float32, device='cuda')
        a_out = torch.empty((N, H, W, Cout), dtype=torch.float32, device='cuda')
        amax = torch.zeros(1, device='cuda')
        ops.conv2d_fwd(x, w, b, pad_top=1, pad_left=1, pad_mode='SYMMETRIC', act='leaky_relu', bn_scale=sc, bn_shift=sh, residual=res,
                       out=outb[..., :Cout], act_out=a_out, y_absmax=amax)
        xn = x.cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)
        a = np_ops.padded_conv2d(xn, w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64), 'SYMMETRIC', 0.0, 'leaky_relu')
        ref = a * sc.cpu().numpy()[None, :, None, None] + sh.cpu().numpy()[None, :, None, None] + res.cpu().numpy().transpose(0, 3, 1, 2)
        assert rel(nchw(a_out), a) < 2e-6 and rel(nchw(outb[..., :Cout]), ref) < 2e-6
        if ldy_extra:
            assert float(outb[..., Cout:].abs().max()) == 0.0
        assert abs(float(amax) - np.abs(ref).max()) < 1e-5 * np.abs(ref).max()


def test_small_route_at_full_size():
    """8 x 1024^2, 3x3 8->8: same numbers as the MFMA implicit GEMM (to rounding), and this is the north star's "conv forward vs HBM
    roofline" layer: 537 MB of tensors per launch."""
    import os
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(8, 1024, 1024, 8, device='cuda', generator=g)
    w = torch.randn(3, 3, 8, 8, device='cuda', generator=g) * 0.1
    y = ops.conv2d_fwd(x, w, None, pad_top=1, pad_left=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.conv2d_fwd(x, w, None, pad_top=1, pad_left=1, out=y)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    gbs = 2 * x.numel() * 4 / ms / 1e6
    print('3x3 8->8 at 8x1024^2: %.3f ms, %.0f GB/s algorithmic (%.0f %% of 8 TB/s)' % (ms, gbs, gbs / 80))
    ref = torch.nn.functional.conv2d(x[:1, :64, :64].permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1)
    assert float((y[:1, 1:63, 1:63].permute(0, 3, 1, 2).double() - ref[:, :, 1:63, 1:63]).norm() / ref[:, :, 1:63, 1:63].norm()) < 2e-6
    # no assertion on the rate: a GPU that has idled through the preceding CPU-side reference work starts these ten launches at its idle clock
    # (one run of the suite measured 9 ms per launch here instead of 0.18 ms); the rate is reported by bench.py (roofline.hbm_bound)


@pytest.mark.parametrize('k,Cin,Cout', [(3, 12, 8), (5, 16, 16), (5, 7, 3)])
def test_weight_gradient_with_more_tiles_than_workgroups(k, Cin, Cout):
    """The MFMA weight-gradient kernel is persistent (tile += splits): 1250 tiles over at most 768 workgroups, channel-sliced operands."""
    import torch.nn.functional as F
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(k + Cin)
    N, H, W = 2, 200, 800
    xb = torch.randn(N, H, W, Cin + 4, device='cuda', generator=g)
    zb = torch.randn(N, H, W, Cout + 1, device='cuda', generator=g)
    x, dz = xb[..., 4:], zb[..., :Cout]
    p = k // 2
    dw = ops.conv2d_wgrad(x, dz, (k, k, Cin, Cout), pad_top=p, pad_left=p)
    xt = x.permute(0, 3, 1, 2).double().cpu()
    wt = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xt, wt, padding=p) * dz.permute(0, 3, 1, 2).double().cpu()).sum().backward()
    ref = wt.grad.permute(2, 3, 1, 0).numpy()
    assert rel(dw.cpu().numpy(), ref) < 5e-6


@pytest.mark.parametrize('k,Cl_in,Cl_out,act,res,raw', [
    (3, 12, 16, 'leaky_relu', False, False),     # the data gradient of a 12 -> 16 ... layer: gradient comes in with Cl_out channels, goes out with Cl_in
    (3, 12, 12, 'leaky_relu', True, True),       # inside a resnet: skip-connection gradient added, raw copy kept
    (3, 8, 8, 'tanh', True, False),
    (3, 8, 12, 'relu', False, True),
    (3, 4, 1, 'leaky_relu', False, False),       # final/out1: one gradient channel in, four out
    (3, 16, 12, 'linear', False, False),
    (5, 16, 16, 'leaky_relu', True, True),
    (3, 4, 4, 'leaky_relu', False, False),
])
def test_data_gradient_with_the_producers_activation_backward_fused(k, Cl_in, Cl_out, act, res, raw):
    """pcnn_conv2d_dgrad_post (round 6): dx = conv(dz; flipped filter) [+ skip gradient], raw copy, times act'(producer's activation), producer's bias gradient
    - against the two calls it replaces (conv2d_fwd + epilogue_bwd): dx and the raw copy bit-identical, the bias gradient equal up to its summation order.
    Ragged image (the last tile row and column are partial) and a batch of 3."""
    from poisson_cnn_amd import ops
    prev_mode = ops._spectral_mode
    ops.set_spectral_mode('off')                                  # 5 x 5 at 16 channels would otherwise go to the spectral route: this test is about the narrow kernel
    try:
        g = torch.Generator(device='cuda').manual_seed(100 * k + 10 * Cl_in + Cl_out)
        N, H, W = 3, 61, 75
        dz = torch.randn(N, H, W, Cl_out, device='cuda', generator=g)
        w = torch.randn(k, k, Cl_in, Cl_out, device='cuda', generator=g) / (k * np.sqrt(Cl_in))
        wf = ops.flip_transpose_weights(w)
        a = torch.randn(N, H, W, Cl_in, device='cuda', generator=g)                     # the producer's saved activation output
        if act == 'tanh':
            a = torch.tanh(a)
        add = torch.randn(N, H, W, Cl_in, device='cuda', generator=g) if res else None
        p = k // 2
        # reference: the two launches
        ref_raw = ops.conv2d_fwd(dz, wf, None, pad_top=k - 1 - p, pad_left=k - 1 - p, residual=add)
        ref_db = torch.zeros(Cl_in, device='cuda')
        ref_dx = ops.epilogue_bwd(ref_raw, a if act != 'linear' else None, act=act, dz=torch.empty_like(ref_raw) if act != 'linear' else None, dbias=ref_db)
        if act == 'linear':
            ref_dx = ref_raw
        # fused
        db = torch.full((Cl_in,), 7.0, device='cuda')                                     # must be overwritten, not accumulated into
        post = ops.Post(a, act, db, want_raw=raw)
        dx = ops.conv2d_dgrad_post(dz, wf, pad_top=k - 1 - p, pad_left=k - 1 - p, out_hw=(H, W), residual=add, post=post)
        assert dx is not None and post.applied, 'the narrow route must take the offer for %dx%d %d->%d' % (k, k, Cl_out, Cl_in)
        torch.cuda.synchronize()
        assert torch.equal(dx, ref_dx)
        if raw:
            assert torch.equal(post.raw, ref_raw)
        assert float((db - ref_db).abs().max()) <= 2e-6 * float(ref_dx.abs().sum(dim=(0, 1, 2)).max())
        # an offer the kernel cannot take is declined, nothing written
        post2 = ops.Post(a[..., :Cl_in - 1] if Cl_in > 1 else a, act, None)
        if Cl_in > 1:
            assert ops.conv2d_dgrad_post(dz, wf, pad_top=k - 1 - p, pad_left=k - 1 - p, out_hw=(H, W), residual=None, post=post2) is None and not post2.applied
    finally:
        ops.set_spectral_mode(prev_mode)
