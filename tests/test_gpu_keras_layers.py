"""Stand-alone layer / block classes with the reference's constructor kwargs and `layer(inputs)` call conventions
(poisson_cnn_amd.keras_layers <- poisson_CNN/layers, /blocks, utils/apply_advanced_padding_and_call_conv_layer.py): forward against the fp64
numpy oracle, the input gradient and EVERY parameter gradient against autograd of the oracle's torch twin."""
import numpy as np
import pytest
import torch

from oracle import hpnn as ohpnn, np_ops, torch_twin

pytestmark = pytest.mark.gpu
TOL = 5e-6


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def randomize(layer, rng, scale=1.0):
    """Random weights (biases / BN statistics too) -> dict name -> fp64 array, also loaded into the layer."""
    w = {}
    for n, t in zip(layer.weight_names, layer.get_weights()):
        if n.endswith('moving_variance') or n.endswith('gamma'):
            v = rng.uniform(0.6, 1.4, t.shape)
        elif n.endswith('kernel'):
            v = t * scale * 1.5
        else:
            v = rng.standard_normal(t.shape) * 0.2
        w[n] = f32(v)
    layer.set_weights(w)
    return w


def check_grads(layer, w, ref_fn, inputs_t, dy, dx_got, in_names=('x',)):
    """ref_fn(params as fp64 torch leaves, inputs as fp64 torch leaves) -> output; compares dL/dinputs and dL/dparams for L = <y, dy>."""
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in w.items()}
    xs = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in inputs_t]
    y = ref_fn(pt, *xs)
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    for got, x in zip(dx_got if isinstance(dx_got, (list, tuple)) else [dx_got], xs):
        assert rel(got, x.grad) < TOL, 'input gradient'
    for n, g in layer.gradients.items():
        ref = pt[n].grad
        assert ref is not None and rel(g, ref) < TOL, n
    return y.detach().numpy()


def test_advanced_padding_conv_and_strided_form():
    from poisson_cnn_amd.keras_layers import Conv2D, apply_advanced_padding_and_call_conv_layer
    rng = np.random.default_rng(0)
    x = f32(rng.standard_normal((2, 5, 21, 26)))
    for stride, mode in ((1, 'SYMMETRIC'), (2, 'CONSTANT'), (3, 'REFLECT')):
        conv = Conv2D(filters=7, kernel_size=5, strides=stride, activation='tf.nn.leaky_relu', padding='same')
        call = apply_advanced_padding_and_call_conv_layer(mode, conv, constant_padding_value=0.25)
        y = call(x, training=True)
        w = randomize(conv, rng)
        y = call(x, training=True)
        ref = np_ops.padded_conv2d(x, w['conv/kernel'], w['conv/bias'], mode, 0.25, 'leaky_relu', stride=stride)
        assert tuple(y.shape) == ref.shape and rel(y, ref) < TOL
        dy = f32(rng.standard_normal(ref.shape))
        dx = call.backward(dy)
        check_grads(conv, w, lambda p, xx: torch_twin.padded_conv2d(xx, p['conv/kernel'], p['conv/bias'], mode, 0.25, 'leaky_relu', stride=stride), [x], dy, dx)
    same = Conv2D(4, 3, padding='same', activation='tanh')
    w = None
    y = same(x)
    w = randomize(same, rng)
    assert rel(same(x), np_ops.same_conv2d(x, w['conv/kernel'], w['conv/bias'], 'tanh')) < TOL


@pytest.mark.parametrize('use_bn', [False, True])
def test_resnet_block(use_bn):
    from poisson_cnn_amd.keras_layers import resnet
    rng = np.random.default_rng(1)
    x = f32(rng.standard_normal((2, 6, 19, 23)))
    blk = resnet(2, use_batchnorm=use_bn, padding_mode='symmetric', filters=6, kernel_size=5, activation='tf.nn.leaky_relu', data_format='channels_first')
    blk(x)
    w = randomize(blk, rng)
    y = blk(x, training=True)
    fn = lambda p, xx: ohpnn.resnet_forward(torch_twin, p, 'resnet', xx, 'SYMMETRIC', 0.0, 'leaky_relu', use_bn)
    ref = ohpnn.resnet_forward(np_ops, w, 'resnet', x, 'SYMMETRIC', 0.0, 'leaky_relu', use_bn)
    assert rel(y, ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    check_grads(blk, w, fn, [x], dy, blk.backward(dy))
    with pytest.raises(ValueError, match='as many input channels'):
        resnet(2, filters=4, kernel_size=3)(x)


@pytest.mark.parametrize('filters,k,act,hw,tile', [(32, 7, 'tf.nn.leaky_relu', (70, 83), 0), (12, 5, 'tf.nn.tanh', (64, 64), 0), (20, 9, 'tf.nn.leaky_relu', (50, 41), 0),
                                                 (32, 15, 'tf.nn.leaky_relu', (70, 83), 64), (28, 13, 'tf.nn.tanh', (101, 60), 64), (24, 11, 'tf.nn.tanh', (66, 120), 64)])
def test_resnet_block_fuses_the_activation_backward_into_the_data_gradient(filters, k, act, hw, tile):
    """Round 4: with zero padding and the data gradients on the 32-point spectral route, each convolution's data-gradient
    kernel applies the activation backward of the convolution BEFORE it (dz = dx act'(a), bias gradient from per-lane partial sums; conv2's also
    hands back the raw gradient for the skip connection).  The block's gradients must match the fp64 oracle exactly as without the fusion,
    the fused entry point must really have been taken (three times per block), and dx / dK must agree with the unfused path to rounding."""
    from poisson_cnn_amd import _lib, ops
    from poisson_cnn_amd.keras_layers import resnet
    rng = np.random.default_rng(3)
    x = f32(rng.standard_normal((2, filters) + hw))
    dy = f32(rng.standard_normal((2, filters) + hw))
    oact = 'leaky_relu' if 'leaky' in act else 'tanh'
    res = {}
    ops.set_spectral_mode('force')
    ops.set_spectral_tile(tile)                                # 64: the 64-point inverse kernel's POST epilogue (second session of round 4), ragged 16-channel halves
    try:
        for fused in (True, False):
            ops.set_post_fusion(fused)
            blk = resnet(2, padding_mode='constant', filters=filters, kernel_size=k, activation=act, data_format='channels_first', seed=5)
            blk(x)
            w = randomize(blk, np.random.default_rng(4))
            y = blk(x, training=True)
            calls = []
            orig = _lib.Handle.call

            def spy(self, name, *a):
                calls.append(name)
                return orig(self, name, *a)
            _lib.Handle.call = spy
            try:
                dx = blk.backward(dy)
            finally:
                _lib.Handle.call = orig
            n_post = calls.count('pcnn_conv2d_bwd_spectral_post')
            # conv2's and conv1's data gradients carry the activation backward of conv1 / conv0 (conv0's has no producer inside the block)
            assert n_post == (2 if fused else 0), calls
            assert calls.count('pcnn_conv2d_epilogue_bwd_absmax') + calls.count('pcnn_conv2d_epilogue_bwd') == (1 if fused else 3)
            fn = lambda p, xx: ohpnn.resnet_forward(torch_twin, p, 'resnet', xx, 'CONSTANT', 0.0, oact, False)
            check_grads(blk, w, fn, [x], dy, dx)
            to_np = lambda t: t.detach().cpu().numpy().copy() if hasattr(t, 'detach') else np.array(t)
            res[fused] = (to_np(dx), {n: to_np(g) for n, g in blk.gradients.items()})
    finally:
        ops.set_post_fusion(True)
        ops.set_spectral_tile(0)
        ops.set_spectral_mode('auto')
    assert rel(res[True][0], res[False][0]) < 1e-6
    for n, g in res[True][1].items():
        assert rel(g, res[False][1][n]) < 2e-6, n


@pytest.mark.parametrize('filters,act,hw', [(8, 'tf.nn.leaky_relu', (37, 45)), (4, 'tf.nn.leaky_relu', (16, 32)), (4, 'relu', (9, 70)), (8, 'linear', (7, 5)),
                                            (8, 'tf.nn.leaky_relu', (50, 33))])
def test_narrow_resnet_stage_runs_as_one_launch(filters, act, hw):
    """Round 4: a 3 x 3 resnet stage of 4 / 8 channels with zero padding is ONE kernel (pcnn_resnet3_fwd: the tile's halo is
    recomputed, conv0's and conv1's outputs live in LDS; an intermediate outside the image is the next convolution's zero padding).  Against the
    fp64 oracle (blocks/resnet.py:29-39 restated in oracle/hpnn.resnet_forward): inference output, training output, every gradient through the
    UNCHANGED backward pass (which consumes the three intermediates the fused launch leaves behind), and agreement with the three-launch chain.
    Shapes: ragged tiles in both axes, images smaller than one tile (8-row tiles) and larger (16-row tiles)."""
    from poisson_cnn_amd import _lib, ops
    from poisson_cnn_amd.keras_layers import resnet
    rng = np.random.default_rng(11)
    x = f32(rng.standard_normal((2, filters) + hw))
    dy = f32(rng.standard_normal((2, filters) + hw))
    oact = {'tf.nn.leaky_relu': 'leaky_relu', 'relu': 'relu', 'linear': 'linear'}[act]
    to_np = lambda t: t.detach().cpu().numpy().copy() if hasattr(t, 'detach') else np.array(t)
    res = {}
    try:
        for fused in (True, False):
            ops.set_stage_fusion(fused)
            blk = resnet(2, padding_mode='constant', filters=filters, kernel_size=3, activation=act, data_format='channels_first', seed=5)
            blk(x)
            w = randomize(blk, np.random.default_rng(4))
            calls = []
            orig = _lib.Handle.call

            def spy(self, name, *a):
                calls.append(name)
                return orig(self, name, *a)
            _lib.Handle.call = spy
            try:
                y_inf = blk(x)
                y = blk(x, training=True)
            finally:
                _lib.Handle.call = orig
            if fused:
                assert calls.count('pcnn_resnet3_fwd') == 2 and not any('conv2d_fwd' in c for c in calls), calls
            else:
                assert 'pcnn_resnet3_fwd' not in calls and sum('conv2d_fwd' in c for c in calls) == 6, calls
            ref = ohpnn.resnet_forward(np_ops, w, 'resnet', x, 'CONSTANT', 0.0, oact, False)
            assert rel(y_inf, ref) < 2e-6 and rel(y, ref) < 2e-6
            dx = blk.backward(dy)
            fn = lambda p, xx: ohpnn.resnet_forward(torch_twin, p, 'resnet', xx, 'CONSTANT', 0.0, oact, False)
            check_grads(blk, w, fn, [x], dy, dx)
            res[fused] = (to_np(y), to_np(dx), {n: to_np(g) for n, g in blk.gradients.items()})
    finally:
        ops.set_stage_fusion(True)
    # same FMA chain per output value: the fused launch and the three-launch chain agree to the last bit where no sum is reordered
    assert np.array_equal(res[True][0], res[False][0])
    assert rel(res[True][1], res[False][1]) < 1e-6
    for n, g in res[True][2].items():
        assert rel(g, res[False][2][n]) < 2e-6, n


def test_fused_stage_refuses_what_it_does_not_cover():
    """pcnn_resnet3_fwd is an optimisation with a narrow domain: outside it the library says so (and layers.resnet never asks: 12 channels, tanh, 5 x 5
    filters, SYMMETRIC padding and BatchNormalization all take the three-launch chain)."""
    from poisson_cnn_amd import _lib, ops
    from poisson_cnn_amd.keras_layers import resnet
    assert ops.resnet3_eligible(8, 'leaky_relu') and ops.resnet3_eligible(4, 'relu')
    assert not ops.resnet3_eligible(12, 'leaky_relu') and not ops.resnet3_eligible(8, 'tanh') and not ops.resnet3_eligible(6, 'linear')
    x = torch.randn(1, 9, 9, 12, device='cuda')
    w = torch.randn(3, 3, 12, 12, device='cuda')
    with pytest.raises(RuntimeError, match='eligible'):
        ops.resnet3_fwd(x, w, None, w, None, w, None, act='leaky_relu', training=False)
    rng = np.random.default_rng(0)
    for kw in (dict(filters=12, kernel_size=3, activation='tf.nn.leaky_relu'), dict(filters=8, kernel_size=3, activation='tf.nn.tanh'),
               dict(filters=8, kernel_size=5, activation='tf.nn.leaky_relu'), dict(filters=8, kernel_size=3, activation='tf.nn.leaky_relu', padding_mode='symmetric'),
               dict(filters=8, kernel_size=3, activation='tf.nn.leaky_relu', use_batchnorm=True)):
        kw.setdefault('padding_mode', 'constant')
        blk = resnet(2, data_format='channels_first', seed=1, **kw)
        calls = []
        orig = _lib.Handle.call

        def spy(self, name, *a):
            calls.append(name)
            return orig(self, name, *a)
        _lib.Handle.call = spy
        try:
            blk(f32(rng.standard_normal((1, kw['filters'], 20, 33))))
        finally:
            _lib.Handle.call = orig
        assert 'pcnn_resnet3_fwd' not in calls, kw


def _oracle_bottleneck(ops_ns, p, x, *, f, up, k, n_convs, mode, val, act, method, pool, use_resnet, use_bn, kdown, kind, resize='bilinear'):
    """blocks/bottleneck_block.py:9-118 restated for every constructor path (conv / pool down-sampling, plain / resnet stages)."""
    H, W = x.shape[2], x.shape[3]
    n_layers, i = 0, 0
    if method == 'conv':
        o = ops_ns.padded_conv2d(x, p['block/downsample/kernel'], p['block/downsample/bias'], mode, val, act, stride=f)
    else:
        o = ops_ns.pool2d_same(x, f, pool)
        if use_resnet:
            o = ops_ns.padded_conv2d(o, p['block/conv0/kernel'], p['block/conv0/bias'], mode, val, act)
            n_layers = 1
    while n_layers < n_convs:
        if use_resnet:
            o = ohpnn.resnet_forward(ops_ns, p, 'block/res%d' % i, o, mode, val, act, use_bn)
            n_layers += 1
        else:
            o = ops_ns.padded_conv2d(o, p['block/conv%d/kernel' % i], p['block/conv%d/bias' % i], mode, val, act)
            if use_bn:
                o = ops_ns.batchnorm_inference(o, p['block/bn%d/gamma' % i], p['block/bn%d/beta' % i], p['block/bn%d/moving_mean' % i], p['block/bn%d/moving_variance' % i])
            n_layers += 2 if use_bn else 1
        i += 1
    out_hw = (int((H / f) * up), int((W / f) * up))
    if kind == 'deconv':
        return ops_ns.conv2d_transpose_same(o, p['block/deconv/kernel'], p['block/deconv/bias'], out_hw, up, 'linear')
    return ops_ns.resize2d(o, out_hw, resize)


@pytest.mark.parametrize('kind,method,use_resnet,use_bn', [('deconv', 'pool', True, True), ('deconv', 'conv', False, True), ('multilinear', 'conv', True, False),
                                                            ('multilinear', 'pool', False, False), ('multilinear', 'pool', False, True)])
def test_bottleneck_blocks(kind, method, use_resnet, use_bn):
    from poisson_cnn_amd.keras_layers import bottleneck_block_deconvupsample, bottleneck_block_multilinearupsample
    rng = np.random.default_rng(2)
    x = f32(rng.standard_normal((2, 5, 24, 30)))
    common = dict(ndims=2, downsampling_factor=3, filters=6, conv_kernel_size=3, conv_activation='tf.nn.leaky_relu', use_resnet=use_resnet, padding_mode='SYMMETRIC',
                  n_convs=3, downsampling_method=method, conv_downsampling_kernel_size=5, pool_downsampling_method='average', use_batchnorm=use_bn)
    if kind == 'deconv':
        blk = bottleneck_block_deconvupsample(deconv_kernel_size=3, **common)
        call = lambda t, training=False: blk(t, training=training)
    else:
        blk = bottleneck_block_multilinearupsample(resize_method='bicubic', **common)
        call = lambda t, training=False: blk([t, np.ones((2, 2))], training=training)
    call(x)
    w = randomize(blk, rng)
    okw = dict(f=3, up=3, k=3, n_convs=3, mode='SYMMETRIC', val=0.0, act='leaky_relu', method=method, pool='average', use_resnet=use_resnet, use_bn=use_bn, kdown=5,
               kind=kind, resize='bicubic')
    ref = _oracle_bottleneck(np_ops, w, x, **okw)
    y = call(x, training=True)
    assert tuple(y.shape) == ref.shape and rel(y, ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    check_grads(blk, w, lambda p, xx: _oracle_bottleneck(torch_twin, p, xx, **okw), [x], dy, blk.backward(dy))


def test_deconvupscale_upsample_spp_jacobi_scaling():
    from poisson_cnn_amd.keras_layers import JacobiIterationLayer, Scaling, SpatialPyramidPool, Upsample, deconvupscale
    rng = np.random.default_rng(3)
    # deconvupscale([x, output_shape])
    x = f32(rng.standard_normal((2, 5, 7, 9)))
    lay = deconvupscale(upsample_ratio=4, filters=6, kernel_size=4)
    out_shape = np.array([2, 6, 27, 34], dtype=np.int32)
    lay([x, out_shape])
    w = randomize(lay, rng)
    y = lay([x, out_shape], training=True)
    ref = np_ops.conv2d_transpose_same(x, w['kernel'], w['bias'], (27, 34), 4, 'linear')
    assert rel(y, ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    check_grads(lay, w, lambda p, xx: torch_twin.conv2d_transpose_same(xx, p['kernel'], p['bias'], (27, 34), 4, 'linear'), [x], dy, lay.backward(dy))
    # kernel_size != upsample_ratio (zero insertion + the fused pad+conv kernels)
    lay = deconvupscale(upsample_ratio=2, filters=4, kernel_size=3)
    out_shape = np.array([2, 4, 13, 18], dtype=np.int32)
    lay([x, out_shape])
    w = randomize(lay, rng)
    y = lay([x, out_shape], training=True)
    ref = np_ops.conv2d_transpose_same(x, w['kernel'], w['bias'], (13, 18), 2, 'linear')
    assert rel(y, ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    check_grads(lay, w, lambda p, xx: torch_twin.conv2d_transpose_same(xx, p['kernel'], p['bias'], (13, 18), 2, 'linear'), [x], dy, lay.backward(dy))
    # Upsample([x, domain_sizes, out_hw])
    for method in ('bilinear', 'bicubic', 'nearest'):
        up = Upsample(2, resize_method=method)
        y = up([x, np.ones((2, 2)), np.array([20, 31])])
        assert rel(y, np_ops.resize2d(x, (20, 31), method)) < TOL
        dy = f32(rng.standard_normal((2, 5, 20, 31)))
        xt = torch.tensor(x, requires_grad=True)
        (torch_twin.resize2d(xt, (20, 31), method) * torch.tensor(dy)).sum().backward()
        assert rel(up.backward(dy), xt.grad) < TOL
    # SpatialPyramidPool(x)
    xs = f32(rng.standard_normal((3, 4, 17, 23)))
    for kind in ('max', 'average'):
        spp = SpatialPyramidPool([[2, 2], 3, [5]], ndims=2, pooling_type=kind)
        y = spp(xs)
        ref = np_ops.spatial_pyramid_pool(xs, [[2, 2], [3, 3], [5, 5]], 'max' if kind == 'max' else 'average')
        assert tuple(y.shape) == (3, 38) and rel(y, ref) < TOL
        dy = f32(rng.standard_normal(ref.shape))
        xt = torch.tensor(xs, requires_grad=True)
        (torch_twin.spatial_pyramid_pool(xt, [[2, 2], [3, 3], [5, 5]], 'max' if kind == 'max' else 'average') * torch.tensor(dy)).sum().backward()
        assert rel(spp.backward(dy), xt.grad) < TOL
    # JacobiIterationLayer([guess, rhs, dx])
    g0, rhs, dx = f32(rng.standard_normal((2, 1, 20, 26))), f32(rng.standard_normal((2, 1, 20, 26))), f32(rng.uniform(0.01, 0.05, (2, 2)))
    dx[:, 1] = dx[:, 0]
    jac = JacobiIterationLayer([3, 3], [2, 2], ndims=2, n_iterations=3)
    y = jac([g0, rhs, dx], training=True)
    ref = np_ops.jacobi_iterations(g0, rhs, dx, 3)
    assert rel(y, ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    gt = torch.tensor(g0, requires_grad=True)
    (torch_twin.jacobi_iterations(gt, torch.tensor(rhs), dx, 3) * torch.tensor(dy)).sum().backward()
    assert rel(jac.backward(dy), gt.grad) < TOL
    # Scaling([x_to_scale, other])
    a, b = f32(rng.standard_normal((2, 1, 40, 44))), f32(rng.standard_normal((2, 1, 40, 44)))
    sc = Scaling(2, stages=2, downsampling_ratio_per_stage=2, spp_levels=[[2, 2], 3], filters=4, kernel_size=3, activation='tf.nn.leaky_relu')
    sc([a, b])
    w = randomize(sc, rng)
    meta = {'stages': 2, 'ratio': 2, 'activation': 'leaky_relu', 'spp_levels': [[2, 2], [3, 3]]}
    y = sc([a, b], training=True)
    ref = ohpnn.scaling_forward(np_ops, w, meta, a, b)
    assert rel(y, ref) < TOL
    dy = f32(rng.standard_normal(ref.shape))
    check_grads(sc, w, lambda p, xx: ohpnn.scaling_forward(torch_twin, p, meta, xx, torch.tensor(b)), [a], dy, sc.backward(dy))


def test_channels_last_layers_match_channels_first():
    """data_format='channels_last' (the kernels' own NHWC layout: nothing is permuted) gives the same numbers as channels_first: a padded
    convolution with its gradients, a resnet block, and deconvupscale with an (N, H, W, C) output_shape."""
    from poisson_cnn_amd.keras_layers import Conv2D, apply_advanced_padding_and_call_conv_layer, deconvupscale, resnet
    rng = np.random.default_rng(77)
    x = f32(rng.standard_normal((2, 6, 21, 17)))
    xl = np.ascontiguousarray(x.transpose(0, 2, 3, 1))
    dy = f32(rng.standard_normal((2, 5, 21, 17)))
    outs = {}
    for fmt in ('channels_first', 'channels_last'):
        conv = Conv2D(filters=5, kernel_size=7, activation='tf.nn.leaky_relu', padding='valid', data_format=fmt, seed=3)
        op = apply_advanced_padding_and_call_conv_layer('SYMMETRIC', conv)
        y = op(x if fmt == 'channels_first' else xl, training=True)
        dx = conv.backward(dy if fmt == 'channels_first' else np.ascontiguousarray(dy.transpose(0, 2, 3, 1)))
        y, dx = y.cpu().numpy(), dx.cpu().numpy()
        if fmt == 'channels_last':
            y, dx = y.transpose(0, 3, 1, 2), dx.transpose(0, 3, 1, 2)
        outs[fmt] = (y, dx, conv.gradients['conv/kernel'].cpu().numpy())
    for a, b in zip(outs['channels_first'], outs['channels_last']):
        np.testing.assert_array_equal(a, b)
    r1, r2 = resnet(ndims=2, filters=6, kernel_size=3, seed=4), resnet(ndims=2, filters=6, kernel_size=3, data_format='channels_last', seed=4)
    np.testing.assert_array_equal(r1(x).cpu().numpy(), r2(xl).cpu().numpy().transpose(0, 3, 1, 2))
    d1 = deconvupscale(upsample_ratio=2, filters=4, kernel_size=2, seed=5)
    d2 = deconvupscale(upsample_ratio=2, filters=4, kernel_size=2, data_format='channels_last', seed=5)
    y1 = d1([x, np.array([2, 4, 41, 34], dtype=np.int32)])
    y2 = d2([xl, np.array([2, 41, 34, 4], dtype=np.int32)])
    assert tuple(y2.shape) == (2, 41, 34, 4)
    np.testing.assert_array_equal(y1.cpu().numpy(), y2.cpu().numpy().transpose(0, 3, 1, 2))
    with pytest.raises(ValueError):
        Conv2D(filters=1, kernel_size=3, data_format='NCHW')


def test_merge_with_attention():
    """layers/MergeWithAttention.py:30-33: softmax over the whole (n, C) weight table, then a per-channel weighted sum of the inputs."""
    from poisson_cnn_amd.keras_layers import MergeWithAttention
    rng = np.random.default_rng(11)
    xs = [f32(rng.standard_normal((2, 6, 19, 23))) for _ in range(3)]
    with pytest.raises(ValueError):
        MergeWithAttention(n_channels=6)
    for prebuilt in (False, True):
        lay = MergeWithAttention(n_channels=6, n_inputs=3) if prebuilt else MergeWithAttention()
        if not prebuilt:
            lay(xs)
        assert lay.weight_names == ['attention_weights'] and lay.count_params() == 18
        assert np.abs(lay.get_weights()[0]).max() <= 0.05           # Keras 'uniform'
        w = {'attention_weights': f32(rng.standard_normal((3, 6)))}
        lay.set_weights(w)
        y = lay(xs, training=True)

        def ref_fn(p, *xx):
            sm = torch.exp(p['attention_weights']) / torch.exp(p['attention_weights']).sum()
            return torch.einsum('bchwn,nc->bchw', torch.stack(xx, -1), sm)

        dy = f32(rng.standard_normal(xs[0].shape))
        ref = check_grads(lay, w, ref_fn, xs, dy, lay.backward(dy))
        assert rel(y, ref) < TOL
    last = MergeWithAttention(data_format='channels_last')
    yl = last([np.ascontiguousarray(x.transpose(0, 2, 3, 1)) for x in xs])
    last.set_weights(w)
    yl = last([np.ascontiguousarray(x.transpose(0, 2, 3, 1)) for x in xs])
    assert rel(yl.permute(0, 3, 1, 2), ref) < TOL
