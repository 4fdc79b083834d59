"""GPU parity of the chained-bottleneck models: Homogeneous_Poisson_NN_Metalearning and Homogeneous_Poisson_NN -
forward and EVERY parameter gradient vs the fp64 autograd restatement (oracle/hpnn_chain.py), and a short training run."""
import copy

import numpy as np
import pytest
import torch

from oracle import hpnn_chain as och

pytestmark = pytest.mark.gpu

TANH3 = ['tf.nn.tanh', 'tf.nn.tanh', 'linear']


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def f32(a):
    return np.asarray(a).astype(np.float32).astype(np.float64)


def meta_config(upsampling, use_bn, use_bias, act='tf.nn.leaky_relu'):
    """The shape of the reference's own example (models/Homogeneous_Poisson_NN_Metalearning.py:334-375), shrunk."""
    pre = {'filters': [4, 6], 'kernel_sizes': [5, 3], 'padding_mode': 'CONSTANT', 'conv_activation': act, 'dense_activations': TANH3,
           'use_bias': use_bias, 'bias_initializer': 'zeros', 'pre_output_dense_units': [6, 8]}
    bott = {'downsampling_factors': [2, 1, 3], 'upsampling_factors': [2, 1, 3], 'filters': 5, 'conv_kernel_sizes': [3, 3, 3], 'n_convs': [2, 2, 2],
            'conv_padding_mode': 'SYMMETRIC', 'conv_conv_activation': act, 'conv_dense_activation': TANH3, 'conv_pre_output_dense_units': [6, 8],
            'conv_use_bias': use_bias, 'use_resnet': True, 'conv_downsampling_kernel_sizes': [2, 3, 3]}
    if upsampling == 'deconv':
        bott.update({'downsampling_factors': [2, 3], 'upsampling_factors': [2, 3], 'conv_kernel_sizes': [3, 3], 'n_convs': [2, 1], 'conv_downsampling_kernel_sizes': [2, 3],
                     'deconv_kernel_sizes': [2, 3], 'deconv_dense_activation': TANH3, 'deconv_pre_output_dense_units': [6, 8], 'use_resnet': False})
    fin = {'filters': [6, 4, 2, 1], 'kernel_sizes': [3, 3, 3, 3], 'padding_mode': 'CONSTANT', 'conv_activation': 'tf.nn.tanh', 'dense_activations': TANH3,
           'use_bias': use_bias, 'pre_output_dense_units': [6, 8], 'bias_initializer': 'zeros', 'final_regular_conv_stages': 2}
    return dict(ndims=2, use_batchnorm=use_bn, output_scaling={'max_domain_size_squared': True, 'rhs_max_magnitude': True},
                pre_bottleneck_convolutions_config=pre, bottleneck_config=bott, final_convolutions_config=fin, bottleneck_upsampling=upsampling)


def plain_config(upsampling, use_bn, method, use_resnet):
    pre = {'filters': [4, 6], 'kernel_sizes': [5, 3], 'padding_mode': 'SYMMETRIC', 'activation': 'tf.nn.leaky_relu', 'use_bias': True}
    bott = {'downsampling_factors': [2, 1, 3], 'upsampling_factors': [2, 1, 3], 'filters': 5, 'conv_kernel_sizes': [3, 5, 3], 'n_convs': [2, 1, 3],
            'padding_mode': 'SYMMETRIC', 'conv_activation': 'tf.nn.leaky_relu', 'conv_use_bias': True, 'use_resnet': use_resnet, 'downsampling_method': method,
            'pool_downsampling_method': 'average', 'conv_downsampling_kernel_sizes': [2, 3, 3]}
    if upsampling == 'deconv':
        bott['deconv_kernel_sizes'] = [2, 1, 3]
    fin = {'filters': [6, 3, 1], 'kernel_sizes': [3, 3, 3], 'padding_mode': 'CONSTANT', 'activation': 'tf.nn.tanh', 'use_bias': True}
    return dict(ndims=2, use_batchnorm=use_bn, output_scaling={'max_domain_size_squared': True}, pre_bottleneck_convolutions_config=pre, bottleneck_config=bott,
                final_convolutions_config=fin, bottleneck_upsampling=upsampling)


def randomize(model, rng):
    """Random weights everywhere (biases, BN statistics too) so that no term of a gradient is trivially zero; returns the fp64 dict."""
    w = {}
    for n in model.store.names:
        t = model.store.w[n].cpu().numpy()
        if n.endswith(('moving_variance', 'gamma')):
            v = rng.uniform(0.6, 1.4, t.shape)
        elif n.endswith('kernel'):
            v = t * 1.3
        else:
            v = rng.standard_normal(t.shape) * 0.2
        w[n] = f32(v)
    model.set_weights(w)
    return w


def inputs(N, H, W, seed):
    rng = np.random.default_rng(seed)
    return f32(rng.uniform(-1, 1, (N, 1, H, W))), f32(rng.uniform(5e-3, 5e-2, (N, 2)))


def check_model(model, kw, ref_fn, tol_fwd, tol_grad):
    rng = np.random.default_rng(11)
    w = randomize(model, rng)
    rhs, dx = inputs(2, 24, 18, 3)
    pt = {k: torch.tensor(v, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in w.items()}
    yt = ref_fn(pt, kw, torch.tensor(rhs), torch.tensor(dx))
    y_inf = model([rhs, dx])
    assert tuple(y_inf.shape) == (2, 1, 24, 18)
    assert rel(y_inf.cpu().numpy(), yt.detach().numpy()) < tol_fwd
    y = model.call([rhs, dx], training=True)
    assert torch.equal(y, y_inf)
    dy = f32(rng.standard_normal(tuple(yt.shape)))
    (yt * torch.tensor(dy)).sum().backward()
    model.backward(torch.tensor(dy, dtype=torch.float32, device=y.device))
    worst = 0.0
    for n in model.store.trainable_names():
        ref = pt[n].grad.numpy()
        assert np.abs(ref).max() > 0, n
        worst = max(worst, rel(model.store.g[n].cpu().numpy(), ref))
        assert worst < tol_grad, (n, worst)


@pytest.mark.parametrize('upsampling,use_bn,use_bias,act,tol', [('multilinear', True, False, 'tf.nn.tanh', 2e-4), ('multilinear', True, False, 'tf.nn.leaky_relu', 2e-3),
                                                                ('deconv', False, True, 'tf.nn.leaky_relu', 2e-3)])
def test_hpnn_metalearning_forward_and_gradients(upsampling, use_bn, use_bias, act, tol):
    """With the smooth activation the gradients hold 2e-4 through ~25 per-sample layers; with leaky_relu a pre-activation within fp32 rounding
    of zero takes the other slope than in the fp64 twin (a finite, not a rounding-size, change of that term), hence the looser bound."""
    from poisson_cnn_amd.hpnn_models import Homogeneous_Poisson_NN_Metalearning
    kw = meta_config(upsampling, use_bn, use_bias, act)
    model = Homogeneous_Poisson_NN_Metalearning(**copy.deepcopy(kw), seed=2)
    assert [b.f for b in model.bottleneck_blocks] == sorted([b.f for b in model.bottleneck_blocks], reverse=True)
    check_model(model, kw, och.metalearning_forward, 2e-5, tol)


@pytest.mark.parametrize('upsampling,use_bn,method,use_resnet', [('deconv', True, 'pool', True), ('multilinear', False, 'conv', False), ('deconv', True, 'conv', False)])
def test_hpnn_plain_forward_and_gradients(upsampling, use_bn, method, use_resnet):
    from poisson_cnn_amd.hpnn_models import Homogeneous_Poisson_NN
    kw = plain_config(upsampling, use_bn, method, use_resnet)
    model = Homogeneous_Poisson_NN(**copy.deepcopy(kw), seed=2)
    check_model(model, kw, och.plain_forward, 1e-5, 1e-4)


@pytest.mark.parametrize('which', ['metalearning', 'plain'])
def test_chain_models_train(which):
    """compile + train_step on a fixed batch: the loss falls, the weights API round-trips, soln_max_magnitude scaling runs."""
    from poisson_cnn_amd.hpnn_models import Homogeneous_Poisson_NN, Homogeneous_Poisson_NN_Metalearning
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    if which == 'metalearning':
        model = Homogeneous_Poisson_NN_Metalearning(**meta_config('multilinear', False, True), seed=5)
    else:
        model = Homogeneous_Poisson_NN(**plain_config('deconv', False, 'pool', True), seed=5)
    from poisson_cnn_amd import configs
    lossp = dict(configs.hpnn()['training']['loss_parameters'])                   # MAE + integral loss, as the homogeneous-Poisson trainings use
    model.compile(loss=loss_wrapper(global_batch_size=3, **lossp), optimizer=Adam(learning_rate=2e-3))
    rhs, dx = inputs(3, 24, 18, 7)
    target = f32(np.random.default_rng(1).uniform(-1, 1, (3, 1, 24, 18)) * 1e-3)
    losses = [float(model.train_step(((rhs, dx), target))['loss']) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses
    w = model.get_weights()
    model.set_weights(w)
    assert model.count_params() == sum(int(np.prod(v.shape)) for n, v in zip(model.weight_names, w) if not n.endswith(('moving_mean', 'moving_variance')))
    model.output_scaling['soln_max_magnitude'] = True
    y = model([rhs, dx])
    assert np.allclose(y.abs().amax(dim=(1, 2, 3)).cpu().numpy(), 1.0, atol=1e-6)


@pytest.mark.parametrize('which', ['cnn_metalearning', 'cnn'])
def test_hpnn_train_cli(which, tmp_path):
    """python -m poisson_cnn_amd.train <json> with a model_type in the model section = train/hpnn_train.py: analytic dataset generated on the
    device, two optimizer steps, a checkpoint that loads back into a fresh model."""
    import json, os
    from poisson_cnn_amd import configs, train
    from poisson_cnn_amd.hpnn_models import Homogeneous_Poisson_NN, Homogeneous_Poisson_NN_Metalearning
    cfg = configs.hpnn_metalearning_tiny() if which == 'cnn_metalearning' else configs.hpnn_plain_tiny()
    assert cfg['model']['model_type'] == which
    path = tmp_path / 'cfg.json'
    path.write_text(json.dumps(cfg))
    train.main([str(path), '--checkpoint_dir', str(tmp_path), '--epochs', '1'])
    assert any(f.startswith('chkpt') for f in os.listdir(tmp_path))
    mcfg = dict(cfg['model'])
    mcfg.pop('model_type')
    fresh = (Homogeneous_Poisson_NN_Metalearning if which == 'cnn_metalearning' else Homogeneous_Poisson_NN)(**mcfg, seed=99)
    before = [w.copy() for w in fresh.get_weights()]
    train.load_model_checkpoint(fresh, str(tmp_path))
    assert any(not np.array_equal(a, b) for a, b in zip(before, fresh.get_weights()))
