"""torch.autograd over the libpcnn layers (poisson_cnn_amd/autograd.py; VERDICT r4 missing #3): a model composed from the mirrored layers and ordinary
torch glue trains with loss.backward().  The tape calls the very pair a hand-chained backward calls, so everything is compared BIT FOR BIT:

  * every layer class of keras_layers: output, input gradient and flat parameter gradient of the autograd route == the hand-called pair;
  * a composition with torch glue (add, scale, concatenate, a torch loss) == the same chain written out by hand;
  * the whole reference model as one module: Differentiable(Homogeneous_Poisson_NN_Legacy) under a torch loss == model.backward(dL/dpred), and the
    layer-level gradients against the fp64 autograd twin of the oracle (tolerances of tests/test_gpu_keras_layers.py);
  * torch.optim steps the kernels' own parameter bucket in place; a module applied twice before backward() refuses.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(make, inputs_of, dy_seed=0, n_grad_inputs=1):
    """(manual, autograd) results of one layer: two identically seeded instances, the same inputs and output gradient."""
    from poisson_cnn_amd.autograd import Differentiable
    a, b = make(), make()
    inp = inputs_of()
    ya = a(inp, training=True)
    g = torch.Generator(device='cuda').manual_seed(100 + dy_seed)
    dy = torch.randn(ya.shape, device='cuda', generator=g)
    da = a.backward(dy)
    lay_a = getattr(a, 'layer', a)
    ga = lay_a.store.flat_g.clone() if getattr(lay_a, 'store', None) is not None else None
    mod = Differentiable(b)
    inp_b = inputs_of()
    if isinstance(inp_b, (list, tuple)):
        inp_b = list(inp_b)
        leaves = []
        for i in range(n_grad_inputs):
            inp_b[i] = inp_b[i].clone().requires_grad_(True)
            leaves.append(inp_b[i])
    else:
        inp_b = inp_b.clone().requires_grad_(True)
        leaves = [inp_b]
    yb = mod(inp_b)
    yb.backward(dy)
    assert torch.equal(ya, yb)
    das = da if isinstance(da, (list, tuple)) else [da]
    for leaf, d in zip(leaves, das):
        assert torch.equal(leaf.grad, d)
    if ga is not None and ga.numel():
        assert mod.weight is not None and torch.equal(mod.weight.grad, ga)
        assert mod.weight.data_ptr() == getattr(b, 'layer', b).store.flat_w.data_ptr()        # the parameter IS the kernels' bucket
    return mod


def _x(shape, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randn(shape, device='cuda', generator=g)


def test_every_layer_class_on_the_tape_equals_its_hand_called_pair():
    from poisson_cnn_amd import keras_layers as K
    N, C, H, W = 2, 6, 40, 44
    _pair(lambda: K.Conv2D(8, 3, padding='same', activation='tanh', seed=3), lambda: _x((N, C, H, W), 1))
    _pair(lambda: K.apply_advanced_padding_and_call_conv_layer('SYMMETRIC', K.Conv2D(8, 5, activation='tf.nn.leaky_relu', seed=4)), lambda: _x((N, C, H, W), 2))
    _pair(lambda: K.resnet(2, filters=C, kernel_size=5, activation='tf.nn.leaky_relu', padding_mode='symmetric', use_batchnorm=True, seed=5), lambda: _x((N, C, H, W), 3))
    common = dict(ndims=2, downsampling_factor=4, filters=C, conv_kernel_size=3, conv_activation='tanh', padding_mode='SYMMETRIC', n_convs=2,
                  downsampling_method='pool', pool_downsampling_method='average')
    _pair(lambda: K.bottleneck_block_deconvupsample(deconv_kernel_size=4, use_resnet=True, use_batchnorm=True, seed=6, **common), lambda: _x((N, C, H, W), 4))
    ds = torch.tensor([[1.0, 1.1], [0.7, 1.3]], device='cuda')
    _pair(lambda: K.bottleneck_block_multilinearupsample(resize_method='bicubic', seed=7, **common), lambda: [_x((N, C, H, W), 5), ds])
    _pair(lambda: K.deconvupscale(upsample_ratio=4, filters=C, kernel_size=4, seed=8), lambda: [_x((N, C, 10, 11), 6), np.array([N, C, 38, 43], dtype=np.int32)])
    _pair(lambda: K.Upsample(2, resize_method='bilinear'), lambda: [_x((N, C, 9, 7), 7), np.ones((N, 2)), np.array([36, 29])])
    _pair(lambda: K.Scaling(2, stages=2, downsampling_ratio_per_stage=2, spp_levels=((2, 2), 3), filters=4, kernel_size=3, activation='tanh', seed=9),
          lambda: [_x((N, 1, H, W), 8), _x((N, 1, H, W), 9)])
    _pair(lambda: K.SpatialPyramidPool([[2, 2], 3], ndims=2, pooling_type='max'), lambda: _x((N, C, 21, 26), 10))
    _pair(lambda: K.JacobiIterationLayer([3, 3], [2, 2], ndims=2, n_iterations=3), lambda: [_x((N, 1, H, W), 11), _x((N, 1, H, W), 12), torch.full((N, 1), 0.03, device='cuda')])
    _pair(lambda: K.MergeWithAttention(seed=11), lambda: [_x((N, C, 12, 13), 13), _x((N, C, 12, 13), 14), _x((N, C, 12, 13), 15)], n_grad_inputs=3)


def test_a_composition_with_torch_glue_equals_the_hand_written_chain():
    """h = conv(x); r = resnet(h); y = cat([0.5 r + h, up(pool(h))]); L = mean((y - t)^2): the tape's gradients == the chain rule written out with
    the layers' backward calls and the same torch glue."""
    import torch.nn.functional as F
    from poisson_cnn_amd import keras_layers as K
    from poisson_cnn_amd.autograd import Differentiable
    N, C, H, W = 2, 8, 48, 40

    def build():
        conv = K.apply_advanced_padding_and_call_conv_layer('REFLECT', K.Conv2D(C, 7, activation='tanh', seed=21))
        res = K.resnet(2, filters=C, kernel_size=3, activation='tf.nn.leaky_relu', padding_mode='constant', seed=22)
        dec = K.deconvupscale(upsample_ratio=4, filters=C, kernel_size=4, seed=23)
        return conv, res, dec
    x, t = _x((N, 3, H, W), 31), _x((N, 2 * C, H, W), 32)
    # ---- by hand
    conv, res, dec = build()
    h = conv(x, training=True)
    r = res(h, training=True)
    pooled = F.avg_pool2d(h, 4)
    u = dec([pooled, np.array([N, C, H, W], dtype=np.int32)], training=True)
    y = torch.cat([0.5 * r + h, u], 1)
    dy = 2.0 * (y - t) / y.numel()
    d_a, d_u = dy[:, :C].contiguous(), dy[:, C:].contiguous()
    d_pooled = dec.backward(d_u)
    d_h = d_a + res.backward(0.5 * d_a) + F.interpolate(d_pooled, scale_factor=4, mode='nearest') / 16.0
    d_x = conv.backward(d_h)
    hand = [conv.layer.store.flat_g.clone(), res.store.flat_g.clone(), dec.store.flat_g.clone(), d_x.clone(), y.clone()]
    # ---- on the tape
    mods = [Differentiable(m) for m in build()]
    xt = x.clone().requires_grad_(True)
    h = mods[0](xt)
    r = mods[1](h)
    u = mods[2]([F.avg_pool2d(h, 4), np.array([N, C, H, W], dtype=np.int32)])
    y2 = torch.cat([0.5 * r + h, u], 1)
    loss = (y2 - t).square().mean()
    loss.backward()
    assert torch.equal(y2, hand[4])
    # the glue's own adjoints (mean, cat, avg_pool) run in torch on both sides; where the hand-written chain sums three contributions into d_h in
    # another order than the tape does, the sums differ by rounding: compare at that rounding, the leaves that see no such sum bit for bit
    assert torch.equal(mods[2].weight.grad, hand[2])
    for got, want in ((mods[0].weight.grad, hand[0]), (mods[1].weight.grad, hand[1]), (xt.grad, hand[3])):
        assert float((got - want).norm() / want.norm()) < 1e-6


def test_a_model_composed_from_layers_against_the_fp64_oracle_twin():
    """conv -> resnet -> Scaling, trained through loss.backward(): output and every gradient against oracle/torch_twin (fp64 autograd of the restated ops)."""
    from oracle import torch_twin as T
    from poisson_cnn_amd import keras_layers as K
    from poisson_cnn_amd.autograd import Differentiable
    N, C, H, W = 2, 6, 36, 40
    conv = Differentiable(K.apply_advanced_padding_and_call_conv_layer('SYMMETRIC', K.Conv2D(C, 5, activation='tanh', seed=41)))
    res = Differentiable(K.resnet(2, filters=C, kernel_size=3, activation='tanh', padding_mode='symmetric', seed=42))
    x, t = _x((N, 2, H, W), 51), _x((N, C, H, W), 52)
    xt = x.clone().requires_grad_(True)
    y = res(conv(xt))
    loss = (y - t).square().sum()
    loss.backward()
    cw = {n: torch.tensor(v, dtype=torch.float64, requires_grad=True) for n, v in zip(conv.layer.weight_names, conv.layer.get_weights())}
    rw = {n: torch.tensor(v, dtype=torch.float64, requires_grad=True) for n, v in zip(res.layer.weight_names, res.layer.get_weights())}
    rkey = {k: [n for n in rw if n.endswith(k)][0] for k in ('conv0/kernel', 'conv0/bias', 'conv1/kernel', 'conv1/bias', 'conv2/kernel', 'conv2/bias')}
    x64 = x.double().cpu().requires_grad_(True)
    h = T.padded_conv2d(x64, cw['conv/kernel'], cw['conv/bias'], 'SYMMETRIC', 0.0, 'tf.nn.tanh')
    o = T.padded_conv2d(h, rw[rkey['conv0/kernel']], rw[rkey['conv0/bias']], 'SYMMETRIC', 0.0, 'tf.nn.tanh')
    o = T.padded_conv2d(o, rw[rkey['conv1/kernel']], rw[rkey['conv1/bias']], 'SYMMETRIC', 0.0, 'tf.nn.tanh')
    o = T.padded_conv2d(h + o, rw[rkey['conv2/kernel']], rw[rkey['conv2/bias']], 'SYMMETRIC', 0.0, 'tf.nn.tanh')
    ((o - t.double().cpu()) ** 2).sum().backward()

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm())
    assert rel(y.detach(), o.detach()) < 2e-6
    assert rel(xt.grad, x64.grad) < 2e-5
    ref_c = torch.cat([cw[n].grad.reshape(-1) for n in conv.layer.weight_names])
    ref_r = torch.cat([rw[n].grad.reshape(-1) for n in res.layer.weight_names])
    assert rel(conv.weight.grad, ref_c) < 2e-5 and rel(res.weight.grad, ref_r) < 2e-5


def test_the_reference_model_as_one_module_under_a_torch_loss():
    """Differentiable(Homogeneous_Poisson_NN_Legacy)([rhs, dx]) with the loss written in torch: the parameter gradient autograd returns is, bit for bit,
    what model.backward(dL/dpred) leaves in the model's gradient bucket - and a torch.optim step moves the weights the kernels read."""
    from oracle import hpnn as ohpnn
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.autograd import Differentiable
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    cfg = configs.hpnn_tiny()['model']
    rng = np.random.default_rng(3)
    rhs = torch.tensor(rng.uniform(-1, 1, (2, 1, 40, 44)).astype(np.float32), device='cuda')
    dx = torch.tensor(rng.uniform(5e-3, 5e-2, (2, 1)).astype(np.float32), device='cuda')
    tgt = torch.tensor(rng.standard_normal((2, 1, 40, 44)).astype(np.float32) * 0.1, device='cuda')
    p = ohpnn.init_params(cfg, seed=5, gain=1.4, randomize_all=True)
    a, b = Homogeneous_Poisson_NN_Legacy(**cfg), Homogeneous_Poisson_NN_Legacy(**cfg)
    a.set_weights(p); b.set_weights(p)
    pred_a = a.call([rhs, dx], training=True)
    dpred = 2.0 * (pred_a - tgt) / pred_a.numel()
    a.backward(dpred)
    mod = Differentiable(b)
    pred_b = mod([rhs, dx])
    loss = (pred_b - tgt).square().mean()
    loss.backward()
    assert torch.equal(pred_a, pred_b)
    assert mod.weight.numel() == b.store.flat_w.numel() and torch.equal(mod.weight.grad, a.store.flat_g)
    w0 = b.store.w['pre/conv0/kernel'].clone()
    torch.optim.SGD(mod.parameters(), lr=1e-3).step()
    assert not torch.equal(b.store.w['pre/conv0/kernel'], w0)                       # the optimizer stepped the bucket the kernels read
    assert torch.equal(b.store.w['pre/conv0/kernel'], w0 - 1e-3 * a.store.g['pre/conv0/kernel'])
    with torch.no_grad():
        assert torch.equal(mod([rhs, dx]), b([rhs, dx]))                             # inference path, new weights


def test_a_module_applied_twice_before_backward_refuses():
    from poisson_cnn_amd import keras_layers as K
    from poisson_cnn_amd.autograd import Differentiable
    conv = Differentiable(K.Conv2D(4, 3, padding='same', seed=1))
    x = _x((1, 3, 16, 16), 1).requires_grad_(True)
    y = conv(x)
    with pytest.raises(RuntimeError, match='applied twice'):
        conv(y.detach()[:, :3].contiguous().requires_grad_(True))
    y.sum().backward()
    conv(x).sum().backward()                                                         # after backward() the layer is free again
    conv.eval()
    assert conv(x).requires_grad is False                                            # eval(): the inference path, nothing on the tape
