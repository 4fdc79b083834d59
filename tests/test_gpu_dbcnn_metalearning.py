"""GPU parity of Dirichlet_BC_NN_Metalearning (models/Dirichlet_BC_NN_Metalearning.py:13-208): forward and EVERY parameter
gradient vs the fp64 autograd restatement (oracle/dbcnn_metalearning.py), the reference's own __main__ configuration at its own size, a short
training run, and the pieces the model added to the library (row softmax, the Dense 'softmax' activation, LayerNormalization between Dense layers,
1-D metalearning_resnet)."""
import copy

import numpy as np
import pytest
import torch

from oracle import dbcnn_metalearning as odm, metalearning as oml

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def f32(a):
    return np.asarray(a).astype(np.float32).astype(np.float64)


def main_config():
    """The configuration of the reference's __main__ block (:219-250)."""
    bccfg = {'filters': [4, 8, 16, 22], 'kernel_sizes': [19, 17, 15, 13, 11], 'padding_mode': 'SYMMETRIC', 'conv_activation': 'tf.nn.tanh',
             'dense_activations': ['linear', 'linear', 'linear'], 'pre_output_dense_units': [16, 32], 'use_layernorm': True, 'use_bias': True}
    sppcfg = {'levels': [[2], 3, 5, 8], 'pooling_type': 'average'}
    mlpcfg = {'units': [250, 125, 22], 'activations': ['tf.nn.leaky_relu', 'tf.nn.leaky_relu', 'softmax']}
    fccfg = {'filters': [32, 16, 8, 4, 2, 1], 'kernel_sizes': [7, 5, 3, 3, 3, 3], 'padding_mode': 'CONSTANT', 'constant_padding_value': 0.0,
             'final_regular_conv_stages': 3, 'use_bias': True}
    return dict(ndims=2, data_format='channels_first', boundary_conv_config=bccfg, spp_config=sppcfg, domain_info_mlp_config=mlpcfg,
                final_convolutions_config=fccfg, postsmoother_iterations=0, use_batchnorm=True)


def small_config(use_bn=True, pooling='average', jacobi=0, variant=0):
    if variant == 0:      # the shape of the __main__ configs: layer-normed tanh hyper-networks, symmetric padding on the boundary
        bcc = {'filters': [4, 6], 'kernel_sizes': [7, 5], 'padding_mode': 'SYMMETRIC', 'conv_activation': 'tf.nn.tanh', 'dense_activations': ['tf.nn.tanh', 'tf.nn.tanh', 'linear'],
               'pre_output_dense_units': [6, 10], 'use_layernorm': True, 'use_bias': True}
        fcc = {'filters': [5, 4, 2, 1], 'kernel_sizes': [5, 3, 3, 3], 'padding_mode': 'CONSTANT', 'constant_padding_value': 0.0, 'final_regular_conv_stages': 2, 'use_bias': True}
    else:                 # metalearning_conv's defaults on the boundary (linear, zero padding), no biases + a padded, activated 2-D stack
        bcc = {'filters': [4, 6], 'kernel_sizes': [7, 5], 'use_bias': False}
        fcc = {'filters': [5, 4, 2, 1], 'kernel_sizes': [5, 3, 3, 3], 'padding_mode': 'REFLECT', 'conv_activation': 'tf.nn.tanh', 'dense_activations': 'tf.nn.tanh',
               'use_layernorm': True, 'final_regular_conv_stages': 2, 'use_bias': True}
    return dict(ndims=2, data_format='channels_first', use_batchnorm=use_bn, postsmoother_iterations=jacobi, boundary_conv_config=bcc,
                spp_config={'levels': [[2], 3], 'pooling_type': pooling},
                domain_info_mlp_config={'units': [14, 9, 6], 'activations': ['tf.nn.leaky_relu', 'tf.nn.tanh', 'softmax']}, final_convolutions_config=fcc)


def randomize(model, rng, bias_std=0.05):
    """Random biases / BN statistics / layer-norm parameters so that no gradient term is trivially zero, at scales where the activations stay O(1)
    and the closing tanh convolutions are not saturated."""
    w = {}
    for n in model.store.names:
        t = model.store.w[n].cpu().numpy()
        if n.endswith('layernorm/gamma'):                       # a layer-normed hyper-network emits O(gamma) filter taps: keep tanh(conv) off saturation
            v = rng.uniform(0.04, 0.12, t.shape)
        elif n.endswith(('moving_variance', 'gamma')):
            v = rng.uniform(0.6, 1.4, t.shape)
        elif n.endswith('kernel'):
            v = t
        else:
            v = rng.standard_normal(t.shape) * bias_std
        w[n] = f32(v)
    model.set_weights(w)
    return w


def make_inputs(N, L, seed):
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 1, L)
    bc = np.stack([np.sin(2 * np.pi * (k + 1) * t + rng.uniform(0, 6)) * rng.uniform(0.3, 1.0) + 0.3 * rng.standard_normal(L) for k in range(N)])[:, None, :]
    dx = np.repeat(rng.uniform(5e-3, 5e-2, (N, 1)), 2, axis=1)
    return f32(bc), f32(dx)


def test_constructor_follows_the_reference():
    from poisson_cnn_amd.dbcnn_models import Dirichlet_BC_NN_Metalearning
    kw = main_config()
    for key, msg in (('boundary_conv_config', 'boundary convolutions'), ('spp_config', 'Spatial Pyramid Pooling'), ('final_convolutions_config', 'domain convolutions'),
                     ('domain_info_mlp_config', 'domain info MLP')):
        bad = copy.deepcopy(kw)
        bad[key] = None
        with pytest.raises(ValueError, match=msg):
            Dirichlet_BC_NN_Metalearning(**bad)
    model = Dirichlet_BC_NN_Metalearning(**copy.deepcopy(kw), seed=1)
    assert model.x_dir_nmodes == 22 and model.dense_features == 4 + 18
    # every non-list key of a stage config reaches the stage's layers (get_init_arguments_from_config)
    c0 = model.boundary[0]
    assert (c0.ndims, c0.k, c0.kh, c0.cin, c0.cout, c0.mode, c0.act, c0.use_bias, c0.use_layernorm) == (1, 19, 1, 3, 4, 'SYMMETRIC', 'tanh', True, True)
    assert [tuple(model.store.w['bc/stage0/conv/dense%d/kernel' % i].shape) for i in range(3)] == [(4, 16), (16, 32), (32, 19 * 3 * 4 + 4)]
    r0 = model.boundary[1].convs[2]
    assert (r0.ndims, r0.k, r0.mode, r0.act, r0.use_layernorm) == (1, 19, 'SYMMETRIC', 'tanh', True)
    f0 = model.final_meta[0]
    assert (f0.ndims, f0.k, f0.cin, f0.cout, f0.mode, f0.act, f0.use_layernorm) == (2, 7, 24, 32, 'CONSTANT', 'linear', False)
    assert tuple(model.store.w['final/stage0/conv/dense0/kernel'].shape) == (22, 8)
    # 3 metalearning stages (conv + resnet each), then the 3 plain tanh convolutions
    assert len(model.final_meta) == 6 and [c.cout for c in model.final_regular] == [4, 2, 1]
    names = model.weight_names
    assert 'mlp/ln1/gamma' in names and 'mlp/ln2/beta' in names and 'mlp/ln0/gamma' not in names
    assert sum(n.endswith('moving_mean') for n in names) == 2 * (4 + 3)            # two BatchNormalization layers per metalearning_resnet


@pytest.mark.parametrize('use_bn,pooling,jacobi,variant', [(True, 'average', 0, 0), (False, 'max', 2, 1)])
def test_forward_and_every_gradient(use_bn, pooling, jacobi, variant):
    from poisson_cnn_amd.dbcnn_models import Dirichlet_BC_NN_Metalearning
    kw = small_config(use_bn, pooling, jacobi, variant)
    model = Dirichlet_BC_NN_Metalearning(**copy.deepcopy(kw), seed=3)
    rng = np.random.default_rng(5)
    w = randomize(model, rng)
    bc, dx = make_inputs(3, 37, 2)
    X = 29
    pt = {k: torch.tensor(v, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in w.items()}
    taps = {}
    yt = odm.forward(kw, pt, bc, dx, X, taps=taps)
    y_inf = model([bc, dx, X])
    assert tuple(y_inf.shape) == (3, 1, X, 37)
    assert rel(y_inf, yt) < 2e-5
    assert torch.equal(y_inf[:, :, 0, :].cpu(), torch.tensor(bc, dtype=torch.float32))              # the first row is the boundary condition itself
    y = model.call([bc, dx, X], training=True)
    assert torch.equal(y, y_inf)
    dy = f32(rng.standard_normal(tuple(yt.shape)))
    (yt * torch.tensor(dy)).sum().backward()
    model.backward(torch.tensor(dy, dtype=torch.float32, device=y.device))
    worst = ('', 0.0)
    for n in model.store.trainable_names():
        ref = pt[n].grad.numpy()
        assert np.abs(ref).max() > 0, n
        e = rel(model.store.g[n], ref)
        worst = max(worst, (n, e), key=lambda t: t[1])
    assert worst[1] < 2e-4, worst


def test_reference_main_configuration_at_its_own_size():
    """bsize 10, nx 101, ny 75 (:211-213) with freshly initialised weights: output against the fp64 restatement."""
    from poisson_cnn_amd.dbcnn_models import Dirichlet_BC_NN_Metalearning
    kw = main_config()
    model = Dirichlet_BC_NN_Metalearning(**copy.deepcopy(kw), seed=4)
    w = randomize(model, np.random.default_rng(8), bias_std=0.03)
    bc, dx = make_inputs(10, 75, 6)
    y = model([bc, dx, 101])
    assert tuple(y.shape) == (10, 1, 101, 75) and bool(torch.isfinite(y).all())
    with torch.no_grad():
        yt = odm.forward(kw, {k: torch.tensor(v) for k, v in w.items()}, bc, dx, 101)
    assert rel(y, yt) < 5e-5
    assert model.count_params() == sum(int(np.prod(v.shape)) for n, v in w.items() if not n.endswith(('moving_mean', 'moving_variance')))


def test_train_step_reduces_the_loss():
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.dbcnn_models import Dirichlet_BC_NN_Metalearning
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    model = Dirichlet_BC_NN_Metalearning(**small_config(False), seed=7)
    lossp = dict(configs.dbcnn_tiny()['training']['loss_parameters'])
    model.compile(loss=loss_wrapper(global_batch_size=4, **lossp), optimizer=Adam(learning_rate=2e-3))
    bc, dx = make_inputs(4, 33, 9)
    X = 27
    # a harmonic target with the boundary on its first row: sinh-decaying modes of the boundary data
    xbar = np.linspace(0, 1, X)[None, :, None]
    target = f32(bc[:, :, None, :][:, 0] * np.sinh(3 * (1 - xbar)) / np.sinh(3.0))[:, None]
    losses = [float(model.train_step(((bc, dx[:, :1]), target))['loss']) for _ in range(15)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses
    w = model.get_weights()
    model.set_weights(w)


def test_softmax_dense_and_layernorm_layers():
    from poisson_cnn_amd import layers as L, ops
    rng = np.random.default_rng(2)
    x = f32(rng.standard_normal((5, 37)) * 3)
    xt = torch.tensor(x, dtype=torch.float32, device='cuda')
    y = ops.softmax_fwd(xt)
    ref = torch.softmax(torch.tensor(x), -1)
    assert rel(y, ref) < 1e-6 and np.allclose(y.sum(1).cpu().numpy(), 1.0, atol=1e-6)
    dy = f32(rng.standard_normal((5, 37)))
    xr = torch.tensor(x, requires_grad=True)
    (torch.softmax(xr, -1) * torch.tensor(dy)).sum().backward()
    assert rel(ops.softmax_bwd(y, torch.tensor(dy, dtype=torch.float32, device='cuda')), xr.grad) < 1e-5
    # Dense('softmax') after a LayerNormalization, as the domain-info chain composes them
    S = L.ParamStore()
    ln = L.LayerNormalization(S, 'ln', 37)
    dn = L.Dense(S, 'dense', 37, 11, 'softmax')
    S.finalize(torch.device('cuda'))
    S.initialize(3)
    S.w['ln/gamma'].copy_(torch.tensor(rng.uniform(0.5, 1.5, 37), dtype=torch.float32))
    S.w['ln/beta'].copy_(torch.tensor(rng.standard_normal(37) * 0.2, dtype=torch.float32))
    S.w['dense/bias'].copy_(torch.tensor(rng.standard_normal(11) * 0.2, dtype=torch.float32))
    out = dn.forward(ln.forward(xt))
    pt = {n: S.w[n].detach().cpu().double().requires_grad_(True) for n in S.names}
    xr = torch.tensor(x, requires_grad=True)
    mu = xr.mean(-1, keepdim=True)
    var = ((xr - mu) ** 2).mean(-1, keepdim=True)
    h = (xr - mu) / torch.sqrt(var + 1e-3) * pt['ln/gamma'] + pt['ln/beta']
    ref = torch.softmax(h @ pt['dense/kernel'] + pt['dense/bias'], -1)
    assert rel(out, ref) < 1e-5
    dy = f32(rng.standard_normal((5, 11)))
    (ref * torch.tensor(dy)).sum().backward()
    dx = ln.backward(dn.backward(torch.tensor(dy, dtype=torch.float32, device='cuda')))
    assert rel(dx, xr.grad) < 1e-4
    for n in S.names:
        assert rel(S.g[n], pt[n].grad) < 1e-4, n
    with pytest.raises(ValueError):
        L.ConvUnit(L.ParamStore(), L.Context(), 'c', 3, 2, 2, activation='softmax')


def test_one_dimensional_metalearning_resnet():
    """blocks/metalearning_resnet.py:27-37 with dimensions = 1 (the boundary stages, models/Dirichlet_BC_NN_Metalearning.py:55-57), call convention
    [x (N,C,L), dense_input]."""
    from poisson_cnn_amd.metalearning import metalearning_resnet
    rng = np.random.default_rng(4)
    blk = metalearning_resnet(filters=5, kernel_size=7, dimensions=1, use_batchnorm=True, seed=6)
    x, d = f32(rng.standard_normal((3, 5, 41))), f32(rng.standard_normal((3, 4)))
    xt, dt = torch.tensor(x, dtype=torch.float32, device='cuda'), torch.tensor(d, dtype=torch.float32, device='cuda')
    blk([xt, dt])
    w = {}
    for n in blk.store.names:
        t = blk.store.w[n].cpu().numpy()
        w[n] = f32(rng.uniform(0.6, 1.4, t.shape) if n.endswith(('moving_variance', 'gamma')) else (t * 1.3 if n.endswith('kernel') else rng.standard_normal(t.shape) * 0.2))
        blk.store.w[n].copy_(torch.tensor(w[n], dtype=torch.float32))
    y = blk([xt, dt], training=True)
    pt = {k: torch.tensor(v, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in w.items()}
    xr, dr = torch.tensor(x, requires_grad=True), torch.tensor(d, requires_grad=True)
    ref = oml.mresnet(pt, 'metalearning_resnet', xr[:, :, None, :], dr, 7, 5, ['linear'] * 3, True, kh=1)[:, :, 0, :]
    assert tuple(y.shape) == (3, 5, 41) and rel(y, ref) < 1e-5
    dy = f32(rng.standard_normal((3, 5, 41)))
    (ref * torch.tensor(dy)).sum().backward()
    dyt = torch.tensor(dy, dtype=torch.float32, device='cuda').permute(0, 2, 1).unsqueeze(1).contiguous()
    dx, dd = blk.backward(dyt)
    assert rel(dx.squeeze(1).permute(0, 2, 1), xr.grad) < 1e-4 and rel(dd, dr.grad) < 1e-4
    for n in blk.store.trainable_names():
        assert rel(blk.store.g[n], pt[n].grad) < 1e-4, n
