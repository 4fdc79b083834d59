"""GPU parity of every libpcnn primitive (through the C-ABI) against the fp64 oracle; backward kernels are checked
against autograd of the oracle's torch twin."""
import numpy as np
import pytest
import torch

from oracle import np_ops, torch_twin, loss as oloss

pytestmark = pytest.mark.gpu
TOL = 2e-6      # per-op rel-L2 (fp32 kernels vs fp64 oracle)
TOL_RED = 5e-6  # long fp32 reductions (filter gradients over ~1e4 pixels)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def nhwc(a):
    return torch.tensor(np.ascontiguousarray(np.asarray(a).transpose(0, 2, 3, 1)), dtype=torch.float32, device='cuda')


def nchw(t):
    return t.detach().cpu().numpy().transpose(0, 3, 1, 2)


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device='cuda')


def f32(a):
    return np.asarray(a).astype(np.float32).astype(np.float64)


BWD_CASES = [
    (3, 8, 8, 20, 37, 'CONSTANT'), (5, 32, 32, 16, 32, 'CONSTANT'), (7, 64, 32, 19, 45, 'CONSTANT'), (15, 3, 4, 40, 50, 'SYMMETRIC'),
    (13, 4, 16, 35, 33, 'SYMMETRIC'), (11, 16, 32, 17, 64, 'SYMMETRIC'), (13, 32, 28, 30, 41, 'CONSTANT'), (9, 28, 24, 25, 36, 'REFLECT'),
    (3, 4, 1, 33, 31, 'CONSTANT'), (15, 32, 32, 33, 47, 'CONSTANT'), (3, 2, 4, 21, 22, 'CONSTANT'), (5, 32, 32, 2, 3, 'CONSTANT'),
]


@pytest.mark.parametrize('k,Cin,Cout,H,W,mode', BWD_CASES)
def test_conv_backward_matches_autograd(k, Cin, Cout, H, W, mode):
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(k * 100 + Cin + Cout)
    N = 2
    x = f32(rng.standard_normal((N, Cin, H, W)))
    w = f32(rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin))
    dz = f32(rng.standard_normal((N, Cout, H, W)))
    xt = torch.tensor(x, requires_grad=True); wt = torch.tensor(w, requires_grad=True)
    y = torch_twin.padded_conv2d(xt, wt, None, mode, 0.0, 'linear')
    (y * torch.tensor(dz)).sum().backward()
    p = k // 2
    xd, dzd, wd = nhwc(x), nhwc(dz), dev(w)
    dw = ops.conv2d_wgrad(xd, dzd, w.shape, pad_top=p, pad_left=p, pad_mode=mode)
    wflip = ops.flip_transpose_weights(wd)
    if mode == 'CONSTANT':
        dx = ops.conv2d_fwd(dzd, wflip, None, pad_top=k - 1 - p, pad_left=k - 1 - p)
    else:
        gp = ops.conv2d_fwd(dzd, wflip, None, pad_top=k - 1, pad_left=k - 1, out_hw=(H + k - 1, W + k - 1))
        dx = ops.pad_fold_bwd(gp, (H, W), ((p, k - 1 - p), (p, k - 1 - p)), mode)
    torch.cuda.synchronize()
    assert rel(dw.cpu().numpy(), wt.grad.numpy()) < TOL_RED
    assert rel(nchw(dx), xt.grad.numpy()) < TOL


@pytest.mark.parametrize('mode', ['SYMMETRIC', 'REFLECT', 'CONSTANT'])
@pytest.mark.parametrize('C', [8, 6])
def test_pad_fold_and_pool_float4_forms(mode, C):
    """tf.pad adjoint (float4 form for channel counts that are multiples of 4, scalar otherwise; plain and accumulate mode) and the
    float4 pooling forward, against numpy.pad's index map / the numpy oracle."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(C)
    N, H, W, pt, pb, pl, pr = 2, 11, 9, 3, 2, 4, 1
    gp = f32(rng.standard_normal((N, C, H + pt + pb, W + pl + pr)))
    tmode = {'SYMMETRIC': 'symmetric', 'REFLECT': 'reflect', 'CONSTANT': 'constant'}[mode]
    # the adjoint through an index map: padded[i, j] = x[iy[i], ix[j]] (or the constant)
    iy = np.pad(np.arange(H), (pt, pb), mode=tmode, **({'constant_values': -1} if mode == 'CONSTANT' else {}))
    ix = np.pad(np.arange(W), (pl, pr), mode=tmode, **({'constant_values': -1} if mode == 'CONSTANT' else {}))
    ref = np.zeros((N, C, H, W))
    for i, sy in enumerate(iy):
        for j, sx in enumerate(ix):
            if sy >= 0 and sx >= 0:
                ref[:, :, sy, sx] += gp[:, :, i, j]
    dx = ops.pad_fold_bwd(nhwc(gp), (H, W), ((pt, pb), (pl, pr)), mode)
    assert rel(nchw(dx), ref) < TOL
    base = f32(rng.standard_normal(ref.shape))
    acc = nhwc(base)
    ops.pad_fold_bwd(nhwc(gp), (H, W), ((pt, pb), (pl, pr)), mode, out=acc, accumulate=True)
    assert rel(nchw(acc), base + ref) < TOL
    x = f32(rng.standard_normal((N, C, 14, 10)))
    for f in (2, 3):
        for kind in ('average', 'max'):
            assert rel(nchw(ops.pool2d_fwd(nhwc(x), f, kind)), np_ops.pool2d_same(x, f, kind)) < TOL


def test_epilogue_bwd_and_bn_fold():
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(0)
    N, H, W, C = 2, 19, 23, 28
    dy = f32(rng.standard_normal((N, H, W, C))); a = f32(rng.standard_normal((N, H, W, C)))
    g, b, m, v = [f32(t) for t in (rng.uniform(0.5, 1.5, C), rng.standard_normal(C), rng.standard_normal(C), rng.uniform(0.5, 2, C))]
    sc, sh = ops.empty((C,)), ops.empty((C,))
    ops.bn_fold(dev(g), dev(b), dev(m), dev(v), sc, sh)
    assert rel(sc.cpu().numpy(), g / np.sqrt(v + 1e-3)) < 1e-6 and rel(sh.cpu().numpy(), b - m * g / np.sqrt(v + 1e-3)) < 1e-6
    for act, gfun in (('leaky_relu', lambda a: np.where(a > 0, 1.0, 0.2)), ('tanh', lambda a: 1 - a * a), ('linear', lambda a: np.ones_like(a))):
        dz = ops.empty((N, H, W, C)); db, s1, s2 = ops.empty((C,)), ops.empty((C,)), ops.empty((C,))
        ops.epilogue_bwd(dev(dy), dev(a), act=act, bn_scale=sc, dz=dz, dbias=db, s_dy_a=s1, s_dy=s2)
        ref = dy * (g / np.sqrt(v + 1e-3)) * gfun(a)
        assert rel(dz.cpu().numpy(), ref) < TOL
        assert rel(db.cpu().numpy(), ref.sum((0, 1, 2))) < TOL_RED
        assert rel(s1.cpu().numpy(), (dy * a).sum((0, 1, 2))) < TOL_RED and rel(s2.cpu().numpy(), dy.sum((0, 1, 2))) < TOL_RED
    dg, dbt = ops.empty((C,)), ops.empty((C,))
    ops.bn_fold_bwd(s1, s2, dev(m), dev(v), dg, dbt)
    # autograd reference of inference BN
    at = torch.tensor(a.transpose(0, 3, 1, 2)); gt = torch.tensor(g, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
    (torch_twin.batchnorm_inference(at, gt, bt, m, v) * torch.tensor(dy.transpose(0, 3, 1, 2))).sum().backward()
    assert rel(dg.cpu().numpy(), gt.grad.numpy()) < 2e-5 and rel(dbt.cpu().numpy(), bt.grad.numpy()) < TOL_RED


@pytest.mark.parametrize('H,W,f', [(12, 8, 2), (13, 10, 3), (37, 29, 4), (33, 47, 8), (33, 17, 16), (70, 90, 32), (5, 6, 8)])
def test_pool_fwd_bwd(H, W, f):
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(f)
    x = f32(rng.standard_normal((2, 5, H, W)))
    for kind in ('average', 'max'):
        xt = torch.tensor(x, requires_grad=True)
        yt = torch_twin.pool2d_same(xt, f, kind)
        dy = f32(rng.standard_normal(yt.shape))
        (yt * torch.tensor(dy)).sum().backward()
        xd = nhwc(x)
        y = ops.pool2d_fwd(xd, f, kind)
        assert rel(nchw(y), np_ops.pool2d_same(x, f, kind)) < TOL
        dx = ops.pool2d_bwd(xd, nhwc(dy), f, kind)
        assert rel(nchw(dx), xt.grad.numpy()) < TOL
        acc = torch.ones_like(dx)
        ops.pool2d_bwd(xd, nhwc(dy), f, kind, dx=acc, accumulate=True)
        assert rel(nchw(acc), xt.grad.numpy() + 1.0) < TOL


@pytest.mark.parametrize('H,W,f', [(12, 8, 2), (13, 10, 3), (37, 29, 4), (33, 47, 8), (33, 17, 16)])
def test_deconv_fwd_bwd(H, W, f):
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(f)
    hc, wc, Ci, Co = -(-H // f), -(-W // f), 32, 32
    x = f32(rng.standard_normal((2, Ci, hc, wc))); k = f32(rng.standard_normal((f, f, Co, Ci)) / 6); b = f32(rng.standard_normal(Co))
    xt, kt, bt = torch.tensor(x, requires_grad=True), torch.tensor(k, requires_grad=True), torch.tensor(b, requires_grad=True)
    yt = torch_twin.conv2d_transpose_same(xt, kt, bt, (H, W), f)
    dy = f32(rng.standard_normal(yt.shape))
    (yt * torch.tensor(dy)).sum().backward()
    alpha = 1.0 / 256
    base = f32(rng.standard_normal((2, H, W, Co)))
    y = dev(base).clone()
    ops.deconv_fwd(nhwc(x), dev(k), dev(b), (H, W), f, alpha=alpha, beta=1.0, out=y)
    assert rel(nchw(y), np_ops.conv2d_transpose_same(x, k, b, (H, W), f) * alpha + base.transpose(0, 3, 1, 2)) < TOL
    dx = ops.deconv_bwd_data(nhwc(dy), dev(k), (hc, wc), f, alpha=alpha)
    assert rel(nchw(dx), xt.grad.numpy() * alpha) < TOL
    dbias = ops.empty((Co,))
    dk = ops.deconv_bwd_filter(nhwc(x), nhwc(dy), f, alpha=alpha, dbias=dbias)
    assert rel(dk.cpu().numpy(), kt.grad.numpy() * alpha) < TOL_RED
    assert rel(dbias.cpu().numpy(), bt.grad.numpy() * alpha) < TOL_RED


@pytest.mark.parametrize('H,W,f,k,Ci,Co', [(40, 52, 2, 5, 8, 6), (33, 47, 3, 4, 20, 32), (48, 64, 4, 7, 32, 32), (30, 45, 3, 2, 5, 7), (26, 26, 2, 3, 4, 4)])
def test_deconv_kernel_size_differs_from_stride(H, W, f, k, Ci, Co):
    """layers/deconvupscale.py:100-106 with kernel_size != upsample_ratio (the reference's own self-test uses upsample_ratio 2, kernel 5, :111):
    zero insertion + the fused pad+conv kernels, forward with the branch-merge alpha / beta, data gradient, filter and bias gradients."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(100 * f + k)
    hc, wc = -(-H // f), -(-W // f)
    x = f32(rng.standard_normal((2, Ci, hc, wc))); kk = f32(rng.standard_normal((k, k, Co, Ci)) / (k * np.sqrt(Ci))); b = f32(rng.standard_normal(Co))
    xt, kt, bt = torch.tensor(x, requires_grad=True), torch.tensor(kk, requires_grad=True), torch.tensor(b, requires_grad=True)
    yt = torch_twin.conv2d_transpose_same(xt, kt, bt, (H, W), f)
    dy = f32(rng.standard_normal(yt.shape))
    (yt * torch.tensor(dy)).sum().backward()
    assert rel(np_ops.conv2d_transpose_same(x, kk, b, (H, W), f), yt.detach().numpy()) < 1e-12        # the two oracles agree
    alpha = 0.25
    base = f32(rng.standard_normal((2, H, W, Co)))
    y = dev(base).clone()
    ops.deconv_fwd(nhwc(x), dev(kk), dev(b), (H, W), f, alpha=alpha, beta=1.0, out=y)
    assert rel(nchw(y), yt.detach().numpy() * alpha + base.transpose(0, 3, 1, 2)) < TOL
    assert rel(nchw(ops.deconv_fwd(nhwc(x), dev(kk), dev(b), (H, W), f)), yt.detach().numpy()) < TOL
    dx = ops.deconv_bwd_data(nhwc(dy), dev(kk), (hc, wc), f, alpha=alpha)
    assert rel(nchw(dx), xt.grad.numpy() * alpha) < TOL
    dbias = ops.empty((Co,))
    dk = ops.deconv_bwd_filter(nhwc(x), nhwc(dy), f, alpha=alpha, dbias=dbias, kernel_size=(k, k))
    assert rel(dk.cpu().numpy(), kt.grad.numpy() * alpha) < TOL_RED
    assert rel(dbias.cpu().numpy(), bt.grad.numpy() * alpha) < TOL_RED


def test_deconvupscale_layer_with_the_reference_self_test_shape():
    """deconvupscale(upsample_ratio=2, filters=2, kernel_size=5) - the constructor call of the reference's own __main__ block
    (layers/deconvupscale.py:111) - through the Keras-style layer: forward and gradients vs the oracle."""
    from poisson_cnn_amd.keras_layers import deconvupscale
    rng = np.random.default_rng(5)
    x = f32(rng.standard_normal((3, 4, 9, 11)))
    lyr = deconvupscale(upsample_ratio=2, filters=2, kernel_size=5, dimensions=2, seed=1)
    shp = np.array([3, 2, 18, 21], dtype=np.int32)
    y = lyr([dev(x), shp], training=True)
    w = dict(zip(lyr.weight_names, lyr.get_weights()))
    xt = torch.tensor(x, requires_grad=True)
    kt, bt = torch.tensor(f32(w['kernel']), requires_grad=True), torch.tensor(f32(w['bias']), requires_grad=True)
    yt = torch_twin.conv2d_transpose_same(xt, kt, bt, (18, 21), 2)
    assert tuple(y.shape) == (3, 2, 18, 21) and rel(y.cpu().numpy(), yt.detach().numpy()) < TOL
    dy = f32(rng.standard_normal(yt.shape))
    (yt * torch.tensor(dy)).sum().backward()
    dx = lyr.backward(dev(dy))
    g = lyr.gradients
    assert rel(dx.cpu().numpy(), xt.grad.numpy()) < TOL
    assert rel(g['kernel'].cpu().numpy(), kt.grad.numpy()) < TOL_RED and rel(g['bias'].cpu().numpy(), bt.grad.numpy()) < TOL_RED


@pytest.mark.parametrize('method', ['nearest', 'bilinear', 'bicubic'])
@pytest.mark.parametrize('hc,wc,Ho,Wo', [(2, 3, 64, 70), (5, 7, 33, 41), (1, 1, 40, 36), (8, 8, 128, 128)])
def test_resize_fwd_bwd(method, hc, wc, Ho, Wo):
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(hc * 10 + wc)
    x = f32(rng.standard_normal((2, 6, hc, wc)))
    ref = np_ops.resize2d(x, (Ho, Wo), method)
    y = ops.resize_fwd(nhwc(x), (Ho, Wo), method)
    assert rel(nchw(y), ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    xt = torch.tensor(x, requires_grad=True)
    (torch_twin.resize2d(xt, (Ho, Wo), method) * torch.tensor(dy)).sum().backward()
    dx = ops.resize_bwd(nhwc(dy), (hc, wc), method, alpha=0.5)
    assert rel(nchw(dx), 0.5 * xt.grad.numpy()) < TOL_RED


@pytest.mark.parametrize('method', ['nearest', 'bilinear', 'bicubic'])
@pytest.mark.parametrize('hc,wc,Ho,Wo', [(5, 7, 33, 41), (37, 9, 150, 40), (8, 8, 128, 96)])
def test_resize_float4_forms(method, hc, wc, Ho, Wo):
    """Channel counts that are multiples of 4 take the two-pass forward (x pass + row-uniform y pass, two rows per pass: odd Ho) and the
    streaming backward (8 / 4 coarse rows per workgroup: hc not a multiple of either); accumulate mode y = beta*y + alpha*resize(x)."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(hc + Wo)
    x = f32(rng.standard_normal((2, 8, hc, wc)))
    ref = np_ops.resize2d(x, (Ho, Wo), method)
    y = ops.resize_fwd(nhwc(x), (Ho, Wo), method)
    assert rel(nchw(y), ref) < TOL
    y0 = f32(rng.standard_normal(ref.shape))
    out = nhwc(y0)
    ops.resize_fwd(nhwc(x), (Ho, Wo), method, alpha=0.25, beta=1.0, out=out)
    assert rel(nchw(out), y0 + 0.25 * ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    xt = torch.tensor(x, requires_grad=True)
    (torch_twin.resize2d(xt, (Ho, Wo), method) * torch.tensor(dy)).sum().backward()
    dx = ops.resize_bwd(nhwc(dy), (hc, wc), method, alpha=0.5)
    assert rel(nchw(dx), 0.5 * xt.grad.numpy()) < TOL_RED


def test_dense_spp_scales_ring_jacobi():
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(4)
    # dense
    x = f32(rng.standard_normal((5, 38))); w = f32(rng.standard_normal((38, 100)) / 6); b = f32(rng.standard_normal(100))
    for act in ('leaky_relu', 'linear'):
        xt, wt, bt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True), torch.tensor(b, requires_grad=True)
        yt = torch_twin.dense(xt, wt, bt, act); dy = f32(rng.standard_normal(yt.shape)); (yt * torch.tensor(dy)).sum().backward()
        y = ops.dense_fwd(dev(x), dev(w), dev(b), act)
        assert rel(y.cpu().numpy(), yt.detach().numpy()) < TOL
        dw, db = ops.zeros((38, 100)), ops.zeros((100,))
        dx = ops.dense_bwd(dev(x), dev(w), y, dev(dy), act, dw, db)
        assert rel(dx.cpu().numpy(), xt.grad.numpy()) < TOL and rel(dw.cpu().numpy(), wt.grad.numpy()) < TOL and rel(db.cpu().numpy(), bt.grad.numpy()) < TOL
    # spp
    xs = f32(rng.standard_normal((3, 4, 11, 13)))
    levels = [[2, 2], 3, 5]
    bins = []
    for lv in levels:
        lv = [lv, lv] if isinstance(lv, int) else lv
        iy, ix = np_ops.split_indices(11, lv[0]), np_ops.split_indices(13, lv[1])
        bins += [[iy[a], iy[a + 1], ix[c], ix[c + 1]] for a in range(lv[0]) for c in range(lv[1])]
    bins_d = torch.tensor(np.array(bins, dtype=np.int32), device='cuda')
    out, arg = ops.spp_max_fwd(nhwc(xs), bins_d)
    assert rel(out.cpu().numpy(), np_ops.spatial_pyramid_pool(xs, levels, 'max')) == 0
    xt = torch.tensor(xs, requires_grad=True); ft = torch_twin.spatial_pyramid_pool(xt, levels, 'max'); df = f32(rng.standard_normal(ft.shape))
    (ft * torch.tensor(df)).sum().backward()
    assert rel(nchw(ops.spp_max_bwd(arg, dev(df), (3, 11, 13, 4))), xt.grad.numpy()) < TOL
    # a sample that has gone NaN (a diverged training run) must not turn into a write outside the tensor: the maximum of an all-NaN bin is NaN
    # (tf.reduce_max) and its argmax stays an index inside the sample
    xn = nhwc(xs).clone(); xn[1] = float('nan')
    outn, argn = ops.spp_max_fwd(xn, bins_d)
    assert torch.isnan(outn[1]).all() and torch.equal(outn[0], out[0]) and torch.equal(outn[2], out[2])
    assert int(argn.min()) >= 0 and int(argn.max()) < 11 * 13 * 4
    assert torch.isfinite(ops.spp_max_bwd(argn, dev(df), (3, 11, 13, 4))).all()
    # channel scale / sample scale
    xc = f32(rng.standard_normal((3, 9, 10, 32))); s = f32(rng.standard_normal((3, 32))); dyc = f32(rng.standard_normal(xc.shape))
    assert rel(ops.channel_scale_fwd(dev(xc), dev(s)).cpu().numpy(), xc * s[:, None, None, :]) < TOL
    dxc, dsc = ops.channel_scale_bwd(dev(xc), dev(s), dev(dyc))
    assert rel(dxc.cpu().numpy(), dyc * s[:, None, None, :]) < TOL and rel(dsc.cpu().numpy(), (dyc * xc).sum((1, 2))) < TOL_RED
    g = f32(rng.standard_normal(3)); x1 = f32(rng.standard_normal((3, 9, 10, 1))); d1 = f32(rng.standard_normal(x1.shape))
    assert rel(ops.sample_scale_fwd(dev(x1), dev(g)).cpu().numpy(), x1 * (1 + g)[:, None, None, None]) < TOL
    dx1, dg = ops.sample_scale_bwd(dev(x1), dev(g), dev(d1))
    assert rel(dx1.cpu().numpy(), d1 * (1 + g)[:, None, None, None]) < TOL and rel(dg.cpu().numpy(), (d1 * x1).sum((1, 2, 3))) < TOL_RED
    # bc ring fwd/bwd, jacobi fwd/bwd (adjoint via autograd of the twin)
    u = f32(rng.standard_normal((2, 1, 9, 7))); r = f32(rng.standard_normal(u.shape)); dxx = f32(rng.uniform(0.05, 0.1, (2, 2)))
    for neumann, mode in ((False, 'CONSTANT'), (True, 'SYMMETRIC')):
        ut = torch.tensor(u, requires_grad=True); (torch_twin.bc_ring(ut, mode) * torch.tensor(r)).sum().backward()
        assert rel(ops.bc_ring_fwd(nhwc(u), neumann).cpu().numpy()[..., 0], np_ops.bc_ring(u, mode)[:, 0]) == 0
        assert rel(ops.bc_ring_bwd(nhwc(r), neumann).cpu().numpy()[..., 0], ut.grad.numpy()[:, 0]) < TOL
    ut = torch.tensor(u, requires_grad=True); jt = torch_twin.jacobi_iterations(ut, r, dxx, 1); dj = f32(rng.standard_normal(u.shape))
    (jt * torch.tensor(dj)).sum().backward()
    assert rel(ops.jacobi_sweep(nhwc(u), nhwc(r), dev(dxx)).cpu().numpy()[..., 0], np_ops.jacobi_iterations(u, r, dxx, 1)[:, 0]) < TOL
    assert rel(ops.jacobi_sweep_bwd(nhwc(dj), dev(dxx)).cpu().numpy()[..., 0], ut.grad.numpy()[:, 0]) < TOL


def test_assemble_axpby_adam():
    from poisson_cnn_amd import ops
    from oracle import hpnn
    rng = np.random.default_rng(8)
    rhs = f32(rng.standard_normal((2, 17, 23)))
    got = ops.assemble_input(dev(rhs)).cpu().numpy()
    ref = np.concatenate([rhs[:, None], hpnn.position_embeddings(np_ops, 2, 17, 23)], 1).transpose(0, 2, 3, 1)
    assert rel(got, ref) < TOL
    x = f32(rng.standard_normal((2, 5, 6, 8))); buf = f32(rng.standard_normal((2, 5, 6, 16)))
    yb = dev(buf)
    ops.axpby(0.5, dev(x), 2.0, yb[..., 8:])
    ref = buf.copy(); ref[..., 8:] = 0.5 * x + 2.0 * buf[..., 8:]
    assert rel(yb.cpu().numpy(), ref) < TOL
    n = 1000
    w, g = f32(rng.standard_normal(n)), f32(rng.standard_normal(n))
    wd, md, vd = dev(w), ops.zeros((n,)), ops.zeros((n,))
    m = np.zeros(n); v = np.zeros(n); wr = w.copy()
    for t in range(1, 4):
        ops.adam_step(wd, dev(g), md, vd, 1e-3, 0.9, 0.999, 1e-7, t)
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        wr = wr - 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
    assert rel(wd.cpu().numpy(), wr) < TOL
    # AMSGrad (tf.keras Adam(amsgrad=True)): the denominator keeps the running maximum of v; gradients shrink so that it matters
    wd, md, vd, vh = dev(w), ops.zeros((n,)), ops.zeros((n,)), ops.zeros((n,))
    m = np.zeros(n); v = np.zeros(n); vhat = np.zeros(n); wr = w.copy()
    for t in range(1, 5):
        gt = g / t ** 2
        ops.adam_step(wd, dev(gt), md, vd, 1e-3, 0.9, 0.999, 1e-7, t, vhat=vh)
        m = 0.9 * m + 0.1 * gt; v = 0.999 * v + 0.001 * gt * gt; vhat = np.maximum(vhat, v)
        wr = wr - 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(vhat) + 1e-7)
    assert rel(wd.cpu().numpy(), wr) < TOL and rel(vh.cpu().numpy(), vhat) < 1e-4      # (1 - beta_2) is formed in fp32, as in Keras


@pytest.mark.parametrize('nesterov', [False, True])
def test_sgd_momentum(nesterov):
    """tf.keras.optimizers.SGD(momentum, nesterov) (train/utils.py:7-8): three steps vs the update rule in float64."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(7)
    w0 = f32(rng.standard_normal(1000)); wr = w0.copy(); vr = np.zeros_like(w0)
    w, v = dev(w0), dev(np.zeros(1000))
    lr, mom, gs = 0.05, 0.9, 0.5
    for t in range(3):
        g = f32(rng.standard_normal(1000))
        ops.sgd_momentum_step(w, dev(g), v, lr, mom, nesterov, grad_scale=gs)
        vr = mom * vr - lr * gs * g
        wr = wr + (mom * vr - lr * gs * g if nesterov else vr)
    assert rel(w.cpu().numpy(), wr) < TOL and rel(v.cpu().numpy(), vr) < TOL
    from poisson_cnn_amd.train import SGD
    assert SGD(learning_rate=0.1, momentum=0.9, nesterov=True).momentum == 0.9


def test_loss_partials_and_bwd():
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(12)
    N, H, W = 3, 40, 52
    yt, yp = f32(rng.standard_normal((N, 1, H, W))), f32(rng.standard_normal((N, 1, H, W)))
    G = oloss.integral_weight_map((H, W), 47)
    part = ops.loss_partials(dev(yp), dev(yt), dev(G)).cpu().numpy()
    d = yp - yt
    ref = np.stack([np.abs(d).sum((1, 2, 3)), (d * d).sum((1, 2, 3)), (G * d * d).sum((1, 2, 3)), np.abs(yt).max((1, 2, 3))], 1)
    assert rel(part, ref) < TOL_RED
    cm, cs, ci = f32(rng.uniform(0.1, 1, N)), f32(rng.uniform(0.1, 1, N)), f32(rng.uniform(0.1, 1, N))
    g = ops.loss_bwd(dev(yp), dev(yt), dev(G), dev(cm), dev(cs), dev(ci)).cpu().numpy()
    refg = cm[:, None, None, None] * np.sign(d) + 2 * d * (cs[:, None, None, None] + ci[:, None, None, None] * G)
    assert rel(g, refg) < TOL


@pytest.mark.parametrize('p,scale', [(2, True), (3, True), (4, False), (1, True)])
def test_loss_wrapper_with_other_integral_exponents(p, scale):
    """losses/integral_loss.py:88,153 Lp_norm_power != 2 (odd p keeps the sign of y_true - y_pred, as the reference does) together with the
    1 / peak^p sample weights of losses/loss_wrapper.py:68: loss value and d loss / d prediction vs autograd of the oracle."""
    from poisson_cnn_amd.losses import loss_wrapper
    rng = np.random.default_rng(40 + p)
    N, H, W = 3, 36, 44
    yt, yp = f32(rng.standard_normal((N, 1, H, W))), f32(rng.standard_normal((N, 1, H, W)))
    rhs = f32(rng.standard_normal((N, 1, H, W)))
    dx = f32(rng.uniform(0.01, 0.05, (N, 1)))
    cfg = dict(ndims=2, integral_loss_weight=0.7, integral_loss_config={'n_quadpts': 31, 'Lp_norm_power': p}, physics_informed_loss_weight=0.0,
               physics_informed_loss_config={'stencil_sizes': [5, 5], 'orders': [2, 2], 'normalize': False}, mse_loss_weight=0.2, mae_loss_weight=0.1,
               scale_sample_loss_by_target_peak_magnitude=scale, global_batch_size=6)
    ypt = torch.tensor(yp, requires_grad=True)
    ref = oloss.loss_wrapper(**cfg)(yt, ypt, torch.tensor(rhs), np.concatenate([dx, dx], 1))
    ref.backward()
    L = loss_wrapper(**cfg)
    loss, dpred = L._evaluate(dev(yt), dev(yp), dev(rhs), dev(np.concatenate([dx, dx], 1)), True)
    assert abs(float(loss) - float(ref.detach())) < 2e-5 * abs(float(ref.detach()))
    assert rel(dpred.cpu().numpy(), ypt.grad.numpy()) < 5e-6


@pytest.mark.parametrize('padding,mode', [('same', 'SYMMETRIC'), ('same', 'CONSTANT'), ('valid', 'constant')])
def test_metalearning_conv_forward_backward(padding, mode):
    """Per-sample hyper-network filters (layers/metalearning_conv.py): forward and all gradients vs autograd of the oracle twin."""
    from poisson_cnn_amd.metalearning import metalearning_conv
    rng = np.random.default_rng(21)
    N, H, W, Cin, Cout, k, F = 3, 21, 19, 8, 12, 5, 3
    lay = metalearning_conv(Cout, k, Cin, F, padding=padding, padding_mode=mode, constant_padding_value=0.2, conv_activation='tf.nn.leaky_relu',
                            dense_activations='tf.nn.tanh', pre_output_dense_units=[8, 16], seed=4)
    x = f32(rng.standard_normal((N, Cin, H, W))); di = f32(rng.standard_normal((N, F)))
    names = lay.store.names
    wts = {n: lay.store.w[n].cpu().numpy().astype(np.float64) for n in names}
    wt = {n: torch.tensor(v, requires_grad=True) for n, v in wts.items()}
    xt, dit = torch.tensor(x, requires_grad=True), torch.tensor(di, requires_grad=True)
    kb = dit
    for i in range(3):
        kb = torch_twin.dense(kb, wt['metalearning_conv/dense%d/kernel' % i], wt['metalearning_conv/dense%d/bias' % i], 'tanh')
    nk = k * k * Cin * Cout
    outs = []
    for n in range(N):
        kern = kb[n, :nk].reshape(k, k, Cin, Cout); bias = kb[n, nk:]
        if padding == 'same':
            outs.append(torch_twin.padded_conv2d(xt[n:n + 1], kern, bias, mode, 0.2, 'leaky_relu'))
        else:
            outs.append(torch_twin.activation(torch_twin.conv2d_valid(xt[n:n + 1], kern, bias), 'leaky_relu'))
    yt = torch.cat(outs, 0)
    dy = f32(rng.standard_normal(tuple(yt.shape)))
    (yt * torch.tensor(dy)).sum().backward()
    y = lay.forward(nhwc(x), dev(di))
    assert rel(nchw(y), yt.detach().numpy()) < TOL
    dx, ddi = lay.backward(nhwc(dy))
    assert rel(nchw(dx), xt.grad.numpy()) < 5e-6 and rel(ddi.cpu().numpy(), dit.grad.numpy()) < 2e-5
    for n in names:
        assert rel(lay.store.g[n].cpu().numpy(), wt[n].grad.numpy()) < 2e-5, n


@pytest.mark.parametrize('mode,use_bias', [('CONSTANT', True), ('SYMMETRIC', False)])
def test_metalearning_conv_one_dimensional(mode, use_bias):
    """dimensions = 1 (tf.nn.conv1d per sample, layers/metalearning_conv.py:115-116 - the boundary convolutions of
    models/Dirichlet_BC_NN_Metalearning.py:43-55): forward and all gradients vs torch autograd in fp64; reference call convention (N, C, L)."""
    import torch.nn.functional as F
    from poisson_cnn_amd.metalearning import metalearning_conv
    rng = np.random.default_rng(31)
    N, L, Cin, Cout, k, Fd = 3, 57, 3, 5, 7, 4
    lay = metalearning_conv(Cout, [k], Cin, Fd, padding='same', padding_mode=mode, conv_activation='tf.nn.tanh', dense_activations='tf.nn.tanh',
                            pre_output_dense_units=[6, 8], use_bias=use_bias, dimensions=1, seed=5)
    x = f32(rng.standard_normal((N, Cin, L))); di = f32(rng.standard_normal((N, Fd)))
    names = lay.store.names
    wt = {n: torch.tensor(lay.store.w[n].cpu().numpy().astype(np.float64), requires_grad=True) for n in names}
    xt, dit = torch.tensor(x, requires_grad=True), torch.tensor(di, requires_grad=True)
    kb = dit
    for i in range(3):
        kb = torch_twin.dense(kb, wt['metalearning_conv/dense%d/kernel' % i], wt.get('metalearning_conv/dense%d/bias' % i), 'tanh')
    nk = k * Cin * Cout
    p = k // 2
    outs = []
    for n in range(N):
        kern = kb[n, :nk].reshape(k, Cin, Cout).permute(2, 1, 0)                   # (Cout, Cin, k)
        xin = xt[n:n + 1]
        xin = F.pad(xin, (p, p)) if mode == 'CONSTANT' else torch.cat([xin[:, :, :p].flip(2), xin, xin[:, :, -p:].flip(2)], 2)     # tf.pad SYMMETRIC
        outs.append(torch.tanh(F.conv1d(xin, kern, kb[n, nk:] if use_bias else None)))
    yt = torch.cat(outs, 0)
    dy = f32(rng.standard_normal(tuple(yt.shape)))
    (yt * torch.tensor(dy)).sum().backward()
    y = lay([dev(x), dev(di)], training=True)
    assert tuple(y.shape) == (N, Cout, L) and rel(y.cpu().numpy(), yt.detach().numpy()) < TOL
    dx, ddi = lay.backward(dev(dy).permute(0, 2, 1).contiguous())
    assert rel(dx.permute(0, 2, 1).cpu().numpy(), xt.grad.numpy()) < 5e-6 and rel(ddi.cpu().numpy(), dit.grad.numpy()) < 2e-5
    for n in names:
        assert rel(lay.store.g[n].cpu().numpy(), wt[n].grad.numpy()) < 2e-5, n


def _ml_setup(layer, rng):
    """Random hyper-network weights (biases, layer-norm and BN parameters too) -> fp64 dict, loaded into the layer's store."""
    w = {}
    for n in layer.store.names:
        t = layer.store.w[n].cpu().numpy()
        if n.endswith(('moving_variance', 'gamma')):
            v = rng.uniform(0.6, 1.4, t.shape)
        elif n.endswith('kernel'):
            v = t * 1.3
        else:
            v = rng.standard_normal(t.shape) * 0.2
        w[n] = f32(v)
        layer.store.w[n].copy_(torch.tensor(w[n], dtype=torch.float32))
    return w


def _ml_check(layer, w, ref_fn, x, di, run_fwd):
    pt = {k: torch.tensor(v, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in w.items()}
    xt, dit = torch.tensor(x, requires_grad=True), torch.tensor(di, requires_grad=True)
    yt = ref_fn(pt, xt, dit)
    y = run_fwd()
    assert tuple(y.shape) == tuple(yt.shape) and rel(y.cpu().numpy(), yt.detach().numpy()) < 5e-6
    dy = f32(np.random.default_rng(5).standard_normal(tuple(yt.shape)))
    (yt * torch.tensor(dy)).sum().backward()
    dx, ddi = layer.backward(nhwc(dy))
    assert rel(nchw(dx), xt.grad.numpy()) < 1e-5 and rel(ddi.cpu().numpy(), dit.grad.numpy()) < 3e-5
    for n in layer.store.trainable_names():
        assert rel(layer.store.g[n].cpu().numpy(), pt[n].grad.numpy()) < 3e-5, n


def test_metalearning_conv_layernorm_and_stride():
    """use_layernorm (layers/metalearning_conv.py:128-129) and the strided 'same' form (metalearning_bottleneck_block.py:60-62), lazy build."""
    from oracle import metalearning as oml
    from poisson_cnn_amd.metalearning import metalearning_conv
    rng = np.random.default_rng(31)
    N, H, W, Cin, Cout, k, F = 2, 20, 23, 6, 5, 3, 4
    x, di = f32(rng.standard_normal((N, Cin, H, W))), f32(rng.standard_normal((N, F)))
    lay = metalearning_conv(Cout, k, strides=2, padding='same', padding_mode='SYMMETRIC', conv_activation='tf.nn.leaky_relu', dense_activations='tf.nn.tanh',
                            pre_output_dense_units=[8, 16], use_layernorm=True, seed=2)
    lay([dev(x), dev(di)])                                   # reference call convention, builds lazily
    w = _ml_setup(lay, rng)
    fn = lambda p, xt, dt: oml.mconv(p, 'metalearning_conv', xt, dt, k, Cin, Cout, ['tanh'] * 3, same=True, mode='SYMMETRIC', act='leaky_relu', stride=2, use_layernorm=True)
    _ml_check(lay, w, fn, x, di, lambda: lay([dev(x), dev(di)], training=True))


# (f, coarse grid, output grid): the last three have a TensorFlow 'SAME' crop offset p = (ceil(H / f) f - H) // 2 >= 1 in BOTH axes - round 3's grouped
# kernels dropped it and every test shape had p = 0
DECONV_SHAPES = [(3, (7, 9), (20, 26)), (3, (8, 9), (22, 25)), (4, (3, 4), (10, 13)), (8, (32, 32), (250, 250))]


@pytest.mark.parametrize('f,chw,ohw', DECONV_SHAPES)
def test_metalearning_deconvupscale(f, chw, ohw):
    """layers/metalearning_deconvupscale.py:13-16,28-30: conv2d_transpose(..., padding='SAME') per sample - forward, dK, dbias (through the
    hyper-network's gradients), dx, d(dense input) against oracle/metalearning.mdeconv."""
    from oracle import metalearning as oml
    from poisson_cnn_amd.metalearning import metalearning_deconvupscale
    rng = np.random.default_rng(32)
    N, Cin, F = 2, 5, 3
    x, di = f32(rng.standard_normal((N, Cin) + chw)), f32(rng.standard_normal((N, F)))
    up = metalearning_deconvupscale(f, 4, f, dense_activations='tf.nn.tanh', pre_output_dense_units=[6, 8], seed=1)
    shp = np.array([N, 4, ohw[0], ohw[1]], dtype=np.int32)
    up([dev(x), dev(di), shp])
    w = _ml_setup(up, rng)
    _ml_check(up, w, lambda p, xt, dt: oml.mdeconv(p, 'metalearning_deconvupscale', xt, dt, f, Cin, 4, ['tanh'] * 3, ohw), x, di,
              lambda: up([dev(x), dev(di), shp], training=True))


def test_metalearning_deconvupscale_rejects_a_coarse_grid_that_is_not_same_padding():
    """TensorFlow's conv2d_transpose raises when ceil(output / stride) != input (padding='SAME'); so does the library."""
    from poisson_cnn_amd.metalearning import metalearning_deconvupscale
    up = metalearning_deconvupscale(3, 4, 3, pre_output_dense_units=[6, 8], seed=1)
    x, di = dev(f32(np.zeros((1, 5, 7, 9)))), dev(f32(np.zeros((1, 3))))
    with pytest.raises(RuntimeError, match="SAME"):
        up([x, di, np.array([1, 4, 22, 26], dtype=np.int32)])


def test_metalearning_resnet():
    from oracle import metalearning as oml
    from poisson_cnn_amd.metalearning import metalearning_resnet
    rng = np.random.default_rng(32)
    N, F = 2, 3
    di = f32(rng.standard_normal((N, F)))
    for use_bn in (False, True):
        x2 = f32(rng.standard_normal((N, 6, 15, 17)))
        blk = metalearning_resnet(6, 3, use_batchnorm=use_bn, padding_mode='SYMMETRIC', conv_activation='tf.nn.leaky_relu', dense_activations='tf.nn.tanh',
                                  pre_output_dense_units=[6, 8], seed=3)
        blk([dev(x2), dev(di)])
        w = _ml_setup(blk, rng)
        fn = lambda p, xt, dt: oml.mresnet(p, 'metalearning_resnet', xt, dt, 3, 6, ['tanh'] * 3, use_bn, same=True, mode='SYMMETRIC', act='leaky_relu')
        _ml_check(blk, w, fn, x2, di, lambda: blk([dev(x2), dev(di)], training=True))


@pytest.mark.parametrize('hw', [(18, 24), (22, 25)])       # (22, 25) / 3: coarse 8 x 9, SAME crop offset 1 in both axes of the transposed convolution
@pytest.mark.parametrize('kind,method,use_resnet,use_bn', [('deconv', 'pool', True, True), ('deconv', 'conv', False, True), ('multilinear', 'conv', True, False),
                                                            ('multilinear', 'pool', False, True)])
def test_metalearning_bottleneck_blocks(kind, method, use_resnet, use_bn, hw):
    from oracle import metalearning as oml
    from poisson_cnn_amd.metalearning import metalearning_bottleneck_block_deconvupsample, metalearning_bottleneck_block_multilinearupsample
    rng = np.random.default_rng(33)
    N, Cin, F = 2, 4, 3
    x, di = f32(rng.standard_normal((N, Cin) + hw)), f32(rng.standard_normal((N, F)))
    common = dict(ndims=2, downsampling_factor=3, filters=5, conv_kernel_size=3, n_convs=3, conv_padding_mode='SYMMETRIC', conv_conv_activation='tf.nn.leaky_relu',
                  conv_dense_activation='tf.nn.tanh', conv_pre_output_dense_units=[6, 8], use_resnet=use_resnet, downsampling_method=method,
                  conv_downsampling_kernel_size=5, pool_downsampling_method='average', use_batchnorm=use_bn, seed=4)
    if kind == 'deconv':
        blk = metalearning_bottleneck_block_deconvupsample(deconv_kernel_size=3, deconv_dense_activation='tf.nn.tanh', deconv_pre_output_dense_units=[6, 8], **common)
        call = lambda training=False: blk([dev(x), dev(di)], training=training)
        name = 'metalearning_bottleneck_deconv'
    else:
        blk = metalearning_bottleneck_block_multilinearupsample(**common)
        call = lambda training=False: blk([dev(x), dev(di), None], training=training)
        name = 'metalearning_bottleneck_multilinear'
    call()
    w = _ml_setup(blk, rng)
    okw = dict(kind=kind, f=3, up=3, filters=5, k=3, n_convs=3, acts=['tanh'] * 3, mode='SYMMETRIC', value=0.0, act='leaky_relu', method=method, pool='average',
               use_resnet=use_resnet, use_bn=use_bn, kdown=5, kdeconv=3, dacts=['tanh'] * 3)
    _ml_check(blk, w, lambda p, xt, dt: oml.mbottleneck(p, name, xt, dt, **okw), x, di, lambda: call(training=True))


@pytest.mark.parametrize('mode', ['SYMMETRIC', 'REFLECT'])
@pytest.mark.parametrize('act', ['leaky_relu', 'tanh'])
def test_pad_fold_with_the_producers_activation_backward_in_one_pass(mode, act):
    """pcnn_pad_fold_bwd_post == pad_fold_bwd (+ skip gradient) followed by epilogue_bwd: dz and the raw copy bit for bit (same sums in the same order),
    the bias gradient to summation-order rounding."""
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cpu').manual_seed(11)
    N, H, W, C, pads = 2, 37, 29, 12, ((3, 3), (2, 4))
    gp = torch.randn(N, H + 6, W + 6, C, generator=g).cuda()
    a = torch.tanh(torch.randn(N, H, W, C, generator=g)).cuda()
    skip = torch.randn(N, H, W, C, generator=g).cuda()
    for add_to, want_raw in ((None, False), (skip, True)):
        dx = ops.pad_fold_bwd(gp, (H, W), pads, mode)
        if add_to is not None:
            dx = ops.axpby(1.0, add_to, 1.0, dx)
        db_ref = ops.zeros((C,))
        dz_ref = ops.epilogue_bwd(dx, a, act=act, dz=ops.empty((N, H, W, C), gp.device), dbias=db_ref)
        db = ops.zeros((C,))
        post = ops.Post(a, act, db, want_raw=want_raw)
        dz = ops.pad_fold_bwd_post(gp, (H, W), pads, mode, post, add_to=add_to)
        assert dz is not None and post.applied
        assert torch.equal(dz, dz_ref)
        assert (post.raw is not None) == want_raw and (not want_raw or torch.equal(post.raw, dx))
        assert float((db - db_ref).abs().max()) <= 1e-5 * float(db_ref.abs().max()) + 1e-6
    # an inference-mode BatchNormalization behind the activation: bn_scale in dz, and the two sums of its gamma / beta gradients
    sc = (torch.rand(C, generator=g) + 0.5).cuda()
    dx = ops.pad_fold_bwd(gp, (H, W), pads, mode)
    ref = [ops.zeros((C,)) for _ in range(3)]
    dz_ref = ops.epilogue_bwd(dx, a, act=act, bn_scale=sc, dz=ops.empty((N, H, W, C), gp.device), dbias=ref[0], s_dy_a=ref[1], s_dy=ref[2])
    got = [ops.zeros((C,)) for _ in range(3)]
    post = ops.Post(a, act, got[0], bn_scale=sc, s_dy_a=got[1], s_dy=got[2])
    dz = ops.pad_fold_bwd_post(gp, (H, W), pads, mode, post)
    assert post.applied and torch.equal(dz, dz_ref)
    for u, v in zip(got, ref):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-6
    # not eligible (3 channels): nothing is done, the caller keeps the two-pass route
    post = ops.Post(a[..., :3].contiguous(), act, None)
    assert ops.pad_fold_bwd_post(gp[..., :3].contiguous(), (H, W), pads, mode, post) is None and not post.applied


@pytest.mark.parametrize('act', ['leaky_relu', 'tanh', 'relu'])
def test_channel_scale_backward_with_the_producers_activation_backward(act):
    """pcnn_channel_scale_bwd_post (round 6): the einsum's adjoint also applies the activation backward of the layer whose activation output is the einsum's
    input - dz and ds bit-identical to channel_scale_bwd followed by epilogue_bwd, the bias gradient equal up to summation order; an offer about another
    tensor is declined."""
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(31)
    N, H, W, C = 3, 45, 53, 32
    x = torch.randn(N, H, W, C, device='cuda', generator=g)
    if act == 'tanh':
        x = torch.tanh(x)
    s = torch.randn(N, C, device='cuda', generator=g)
    dy = torch.randn(N, H, W, C, device='cuda', generator=g)
    dx_ref, ds_ref = ops.channel_scale_bwd(x, s, dy)
    db_ref = torch.zeros(C, device='cuda')
    dz_ref = ops.epilogue_bwd(dx_ref, x, act=act, dz=torch.empty_like(dx_ref), dbias=db_ref)
    db = torch.full((C,), -3.0, device='cuda')
    post = ops.Post(x, act, db)
    out = ops.channel_scale_bwd_post(x, s, dy, post)
    assert out is not None and post.applied
    dz, ds = out
    torch.cuda.synchronize()
    assert torch.equal(dz, dz_ref) and torch.equal(ds, ds_ref)
    assert float((db - db_ref).abs().max()) <= 2e-6 * float(dz_ref.abs().sum(dim=(0, 1, 2)).max())
    other = ops.Post(x.clone(), act, db)
    assert ops.channel_scale_bwd_post(x, s, dy, other) is None and not other.applied


@pytest.mark.parametrize('nsrc,beta', [(3, 1.0), (3, 0.0), (2, 1.0)])
def test_resize_branches_summed_in_one_pass_equal_their_separate_calls(nsrc, beta):
    """pcnn_resize_fwd_multi (round 6): bicubic + bilinear + nearest up-sampling of three coarse images accumulated into one destination slice in ONE pass -
    bit-identical to the separate resize_fwd calls (beta, then 1, 1); odd output sizes, a destination that is a channel slice of a wider buffer."""
    from poisson_cnn_amd import ops
    g = torch.Generator(device='cuda').manual_seed(17 + nsrc)
    N, C, Ho, Wo = 3, 32, 67, 93
    shapes = [(3, 4), (5, 6), (9, 12)][:nsrc]
    methods = ['bicubic', 'bilinear', 'nearest'][:nsrc]
    xs = [torch.randn(N, h, w, C, device='cuda', generator=g) for h, w in shapes]
    base = torch.randn(N, Ho, Wo, 2 * C, device='cuda', generator=g)
    ref = base.clone()
    for k, (x, m) in enumerate(zip(xs, methods)):
        ops.resize_fwd(x, (Ho, Wo), m, alpha=0.125, beta=beta if k == 0 else 1.0, out=ref[..., C:])
    got = base.clone()
    assert ops.resize_fwd_multi(xs, (Ho, Wo), methods, alpha=0.125, beta=beta, out=got[..., C:]) is not None
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    assert ops.resize_fwd_multi(xs[:1], (Ho, Wo), methods[:1], alpha=1.0, beta=0.0, out=got[..., C:]) is None      # one source: not this entry point's business


@pytest.mark.parametrize('method', ['nearest', 'bilinear', 'bicubic'])
@pytest.mark.parametrize('C,hc,wc,Ho,Wo', [(32, 2, 3, 256, 384), (32, 9, 5, 288, 207), (16, 3, 3, 97, 131), (64, 4, 2, 130, 67)])
def test_resize_backward_at_the_merge_branches_factors(method, C, hc, wc, Ho, Wo):
    """The adjoint of tf.image.resize at the up-sampling factors of the multilinear bottleneck branches (32 ... 128), 16 / 32 / 64 channels, against the fp64
    autograd twin.  (Round 6 tried its column pass with one wave per coarse pixel - the X window spread over the lanes, butterfly sum: correct, and the train step
    did not move: 187.7 / 187.5 / 186.2 vs 185.8 / 186.4 / 186.2 ms A/B on one box - the 1.1 ms of this latency-bound kernel already runs beside other streams' work.)"""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(C + hc + Wo)
    x = f32(rng.standard_normal((2, C, hc, wc)))
    dy = f32(rng.standard_normal((2, C, Ho, Wo)))
    xt = torch.tensor(x, requires_grad=True)
    (torch_twin.resize2d(xt, (Ho, Wo), method) * torch.tensor(dy)).sum().backward()
    dx = ops.resize_bwd(nhwc(dy), (hc, wc), method, alpha=0.5)
    assert rel(nchw(dx), 0.5 * xt.grad.numpy()) < TOL_RED
