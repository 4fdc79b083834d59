"""GPU parity of the whole Homogeneous_Poisson_NN_Legacy forward pass, the loss and the full training step
(gradients of every parameter, Adam update) against the fp64 oracle / its autograd twin."""
import os

import numpy as np
import pytest
import torch

from oracle import hpnn as ohpnn, np_ops, torch_twin, loss as oloss
from poisson_cnn_amd import configs

pytestmark = pytest.mark.gpu
TOL_FWD = 1e-5    # north-star bound: relative L2 of the solution vs the CPU reference
TOL_GRAD = 2e-4   # flat-gradient rel-L2 (fp32 chain of ~100 layers fwd + bwd vs fp64 autograd)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def make_inputs(N, H, W, seed):
    rng = np.random.default_rng(seed)
    rhs = rng.uniform(-1, 1, (N, 1, H, W))
    rhs /= np.abs(rhs).max(axis=(1, 2, 3), keepdims=True)
    dx = rng.uniform(5e-3, 5e-2, (N, 1))
    return rhs.astype(np.float32).astype(np.float64), dx.astype(np.float32).astype(np.float64)


def build(cfg, seed, gain=1.6):
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    model = Homogeneous_Poisson_NN_Legacy(**cfg)
    p = ohpnn.init_params(cfg, seed=seed, gain=gain, randomize_all=True)
    model.set_weights(p)
    return model, p


@pytest.mark.parametrize('bc', ['dirichlet', 'neumann'])
@pytest.mark.parametrize('H,W', [(36, 40), (53, 47)])
def test_tiny_model_forward(bc, H, W):
    cfg = configs.hpnn_tiny()['model']
    cfg['bc_type'] = bc
    cfg['postsmoother_iterations'] = 2
    model, p = build(cfg, 5)
    rhs, dx = make_inputs(2, H, W, 7)
    ref = ohpnn.forward(np_ops, cfg, p, rhs, dx)
    y = model([rhs, dx]).cpu().numpy()
    assert y.shape == ref.shape
    assert rel(y, ref) < TOL_FWD


@pytest.mark.parametrize('bc', ['dirichlet', 'neumann'])
def test_hpnn_forward_matches_oracle(bc):
    """The shipped hpnn.json model (97 convs + 5 transposed convs, 5.56 M parameters) on a 112 x 120 grid."""
    cfg = configs.hpnn()['model']
    cfg['bc_type'] = bc
    model, p = build(cfg, 11)
    rhs, dx = make_inputs(1, 112, 120, 13)
    taps = {}
    ref = ohpnn.forward(np_ops, cfg, p, rhs, dx, taps=taps)
    y = model([rhs, dx]).cpu().numpy()
    assert np.isfinite(y).all() and np.abs(ref).max() > 0
    assert rel(y, ref) < TOL_FWD


def _train_reference(cfg, p, rhs, dx, target, lossp, gbs, dtype=torch.float64):
    """fp64 (default): the oracle.  fp32: the same graph in PyTorch-CPU single precision - an independent fp32 implementation whose distance
    from the fp64 oracle shows what fp32 arithmetic alone costs."""
    torch_twin.set_dtype(dtype)
    try:
        pt = {k: torch.tensor(v, dtype=dtype, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in p.items()}
        pred = ohpnn.forward(torch_twin, cfg, pt, torch.tensor(rhs, dtype=dtype), torch.tensor(dx, dtype=dtype))
        L = oloss.loss_wrapper(global_batch_size=gbs, **lossp)
        loss = L(target.astype(np.float32 if dtype == torch.float32 else np.float64), pred, torch.tensor(rhs, dtype=dtype), np.concatenate([dx, dx], 1))
        loss.backward()
        grads = {k: v.grad.double().numpy() for k, v in pt.items() if v.requires_grad}
        return float(loss.detach()), pred.detach().double().numpy(), grads
    finally:
        torch_twin.set_dtype(torch.float64)


@pytest.mark.parametrize('bc,pi_w', [('dirichlet', 0.0), ('neumann', 6e-4)])
def test_tiny_model_train_step(bc, pi_w):
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    full = configs.hpnn_tiny()
    cfg = full['model']
    cfg['bc_type'] = bc
    cfg['postsmoother_iterations'] = 1
    lossp = dict(full['training']['loss_parameters'])
    lossp.update(physics_informed_loss_weight=pi_w, mse_loss_weight=0.2)
    lossp['physics_informed_loss_config'] = dict(lossp['physics_informed_loss_config'], inputs_have_max_domain_size_squared_normalization=True)
    model, p = build(cfg, 21)
    rhs, dx = make_inputs(3, 44, 38, 23)
    target = np.random.default_rng(3).standard_normal(rhs.shape).astype(np.float32).astype(np.float64) * 0.1
    ref_loss, ref_pred, ref_g = _train_reference(cfg, p, rhs, dx, target, lossp, 6)
    model.compile(loss=loss_wrapper(global_batch_size=6, **lossp), optimizer=Adam(learning_rate=1e-3))
    w0 = dict(zip(model.weight_names, model.get_weights()))
    logs = model.train_step(((rhs, dx), target))
    assert abs(float(logs['loss']) - ref_loss) < 2e-5 * abs(ref_loss)
    assert abs(float(logs['mse']) - np.mean((ref_pred - target) ** 2)) < 1e-4 * np.mean((ref_pred - target) ** 2)
    g = {n: model.store.g[n].cpu().numpy() for n in model.store.trainable_names()}
    flat = np.concatenate([g[n].ravel() for n in g]); flat_ref = np.concatenate([ref_g[n].ravel() for n in g])
    assert rel(flat, flat_ref) < TOL_GRAD
    for n in g:   # every parameter tensor individually (looser: small tensors carry more relative rounding)
        assert rel(g[n], ref_g[n]) < 2e-3, n
    # Adam: first step moves every weight by lr * sign(g) (up to eps)
    w1 = dict(zip(model.weight_names, model.get_weights()))
    for n in g:
        big = np.abs(ref_g[n]) > 1e-6 * np.abs(ref_g[n]).max()
        step = (w1[n] - w0[n])[big]
        assert np.allclose(step, -1e-3 * np.sign(ref_g[n][big]), rtol=2e-2, atol=1e-6), n
    for n in ('pre/bn0/moving_mean', 'pre/bn0/moving_variance'):
        assert np.array_equal(w0[n], w1[n])


def test_gradient_accumulation_equals_full_batch():
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import SGD
    full = configs.hpnn_tiny()
    cfg = full['model']
    lossp = dict(full['training']['loss_parameters'])
    rhs, dx = make_inputs(4, 40, 36, 2)
    target = np.random.default_rng(5).standard_normal(rhs.shape) * 0.1
    outs = []
    for steps in (None, 2):
        c = dict(cfg, gradient_accumulation_steps=steps)
        model, p = build(c, 31)
        # the reference divides each micro-batch loss by the global batch size and then the summed grads by `steps`
        # (models/Homogeneous_Poisson_NN_Legacy.py:272-287), so accumulated grads = full-batch grads / steps
        model.compile(loss=loss_wrapper(global_batch_size=4, **lossp), optimizer=SGD(learning_rate=0.0))
        model.train_step(((rhs, dx), target))
        outs.append(model.store.flat_g.cpu().numpy().copy())
    assert rel(outs[1] * 2, outs[0]) < 1e-4


@pytest.mark.parametrize('activation,tol', [('tf.nn.leaky_relu', 8e-3), ('tf.nn.tanh', 3e-4)])
def test_hpnn_train_step_gradients(activation, tol):
    """Full hpnn.json model, 2 x 112 x 112 grids: loss and the flat 5.56 M-element gradient vs fp64 autograd.
    With leaky-ReLU the fp32 and fp64 forward passes pick different slopes at the few activations that round to opposite
    signs, which perturbs the gradient by O(1e-3) after ~45 layers (error grows towards the first layers, last layers agree
    to 1e-6); with the smooth tanh activation in the configurable stages the same kernels agree an order of magnitude tighter."""
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    full = configs.hpnn()
    cfg = full['model']
    cfg['pre_bottleneck_convolutions_config']['activation'] = activation
    cfg['bottleneck_deconv_config']['conv_activation'] = activation
    cfg['bottleneck_multilinear_config']['conv_activation'] = activation
    cfg['final_convolutions_config']['activation'] = activation
    lossp = full['training']['loss_parameters']
    model, p = build(cfg, 41, gain=1.6 if 'leaky' in activation else 1.0)
    rhs, dx = make_inputs(2, 112, 112, 43)
    target = np.random.default_rng(9).standard_normal(rhs.shape).astype(np.float32).astype(np.float64) * 0.1
    ref_loss, ref_pred, ref_g = _train_reference(cfg, p, rhs, dx, target, lossp, 2)
    model.compile(loss=loss_wrapper(global_batch_size=2, **lossp), optimizer=Adam(learning_rate=1e-5))
    logs = model.train_step(((rhs, dx), target))
    assert abs(float(logs['loss']) - ref_loss) < 2e-5 * abs(ref_loss)
    names = model.store.trainable_names()
    flat = np.concatenate([model.store.g[n].cpu().numpy().ravel() for n in names])
    flat_ref = np.concatenate([ref_g[n].ravel() for n in names])
    assert flat.size == 5556956
    errs = {n: rel(model.store.g[n].cpu().numpy(), ref_g[n]) for n in names}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print('flat rel', rel(flat, flat_ref), 'worst tensors', worst, 'last layer', errs['final/out1/kernel'])
    assert errs['final/out1/kernel'] < 2e-5 and errs['scaling/dense2/kernel'] < 2e-5
    assert rel(flat, flat_ref) < tol
    if 'leaky' in activation:
        # The 8e-3 is not kernel error: the SAME graph evaluated by PyTorch-CPU in fp32 (oneDNN convolutions, no code shared with
        # libpcnn) sits as far from the fp64 oracle, because fp32 and fp64 pick different leaky-ReLU slopes at activations that
        # round to opposite signs.  The HIP gradient must be no farther from the oracle than ~2x that independent fp32 run, and the
        # effect must be visible in it (otherwise the tolerance above would be hiding something else).
        _, _, g32 = _train_reference(cfg, p, rhs, dx, target, lossp, 2, dtype=torch.float32)
        flat32 = np.concatenate([g32[n].ravel() for n in names])
        d_cpu32, d_hip = rel(flat32, flat_ref), rel(flat, flat_ref)
        print('fp32 torch-CPU twin vs fp64 oracle', d_cpu32, 'HIP vs fp64 oracle', d_hip)
        assert d_cpu32 > 1e-4
        assert d_hip < 2.5 * d_cpu32 + 1e-4


def test_forward_matches_committed_golden_vectors():
    """HIP forward vs the committed oracle fixtures (tests/golden/make_model_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'hpnn_forward_golden.npz'))
    cfg = configs.hpnn()['model']
    model, _ = build(cfg, 11)
    y = model([g['hpnn_rhs'], g['hpnn_dx']]).cpu().numpy()
    assert rel(y, g['hpnn_out']) < TOL_FWD
    cfg = configs.hpnn_tiny()['model']
    cfg['bc_type'] = 'neumann'
    model, _ = build(cfg, 5)
    y = model([g['tiny_rhs'], g['tiny_dx']]).cpu().numpy()
    assert rel(y, g['tiny_out']) < TOL_FWD


def test_training_mode_batchnorm_train_step():
    """batchnorm_training=True: batch statistics in the forward pass, their gradient in the backward pass, moving-average update."""
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    full = configs.hpnn_tiny()
    cfg = full['model']
    for st in ('pre_bottleneck_convolutions_config',):
        cfg[st]['activation'] = 'tf.nn.tanh'
    cfg['bottleneck_deconv_config']['conv_activation'] = 'tf.nn.tanh'
    cfg['bottleneck_multilinear_config']['conv_activation'] = 'tf.nn.tanh'
    lossp = dict(full['training']['loss_parameters'])
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    model = Homogeneous_Poisson_NN_Legacy(batchnorm_training=True, **cfg)
    p = ohpnn.init_params(cfg, seed=51, gain=1.3, randomize_all=True)
    model.set_weights(p)
    rhs, dx = make_inputs(3, 48, 40, 53)
    target = np.random.default_rng(4).standard_normal(rhs.shape).astype(np.float32).astype(np.float64) * 0.1
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in p.items()}
    pred = ohpnn.forward(torch_twin, cfg, pt, torch.tensor(rhs), torch.tensor(dx), bn_training=True)
    L = oloss.loss_wrapper(global_batch_size=3, **lossp)
    loss = L(target, pred, torch.tensor(rhs), np.concatenate([dx, dx], 1))
    loss.backward()
    model.compile(loss=loss_wrapper(global_batch_size=3, **lossp), optimizer=Adam(learning_rate=0.0))
    w0 = dict(zip(model.weight_names, model.get_weights()))
    logs = model.train_step(((rhs, dx), target))
    assert abs(float(logs['loss']) - float(loss.detach())) < 5e-5 * abs(float(loss.detach()))
    names = model.store.trainable_names()
    flat = np.concatenate([model.store.g[n].cpu().numpy().ravel() for n in names])
    flat_ref = np.concatenate([pt[n].grad.numpy().ravel() for n in names])
    assert rel(flat, flat_ref) < 5e-4
    for n in ('pre/bn0/gamma', 'pre/bn0/beta', 'deconv_f3/res0/bn1/gamma'):
        assert rel(model.store.g[n].cpu().numpy(), pt[n].grad.numpy()) < 2e-3, n
    # moving statistics moved towards the batch statistics with momentum 0.99
    w1 = dict(zip(model.weight_names, model.get_weights()))
    x0 = ohpnn.forward(np_ops, cfg, p, rhs, dx, bn_training=True, taps={})   # oracle batch stats of the first BN input
    a0 = np_ops.padded_conv2d(np.concatenate([rhs, ohpnn.position_embeddings(np_ops, 3, 48, 40)], 1), p['pre/conv0/kernel'], p['pre/conv0/bias'],
                              cfg['pre_bottleneck_convolutions_config']['padding_mode'], 0.0, 'tf.nn.tanh')
    bm, bv = a0.mean((0, 2, 3)), a0.var((0, 2, 3))
    n = a0.size / a0.shape[1]
    assert np.allclose(w1['pre/bn0/moving_mean'], 0.99 * w0['pre/bn0/moving_mean'] + 0.01 * bm, rtol=1e-4, atol=1e-6)
    assert np.allclose(w1['pre/bn0/moving_variance'], 0.99 * w0['pre/bn0/moving_variance'] + 0.01 * bv * n / (n - 1), rtol=1e-4, atol=1e-6)


def test_fit_on_device_generated_data_reduces_loss():
    """End to end (BASELINE configs[4] in miniature): analytic pairs generated on the GPU feed model.fit(); the loss must fall."""
    from poisson_cnn_amd.dataset import reverse_poisson_dataset_generator
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    from poisson_cnn_amd.train import Adam, TerminateOnNaN
    full = configs.hpnn_tiny()
    dcfg = dict(full['dataset'])
    dcfg.update(batch_size=4, batches_per_epoch=12, random_output_shape_range=[[48, 64], [48, 64]], fourier_coeff_grid_size_range=[[1, 3], [1, 3]])
    gen = reverse_poisson_dataset_generator(seed=7, **dcfg)
    model = Homogeneous_Poisson_NN_Legacy(seed=3, **full['model'])
    model.compile(loss=loss_wrapper(global_batch_size=4, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=2e-3))
    hist = model.fit(gen, epochs=5, callbacks=[TerminateOnNaN()], verbose=0)
    assert np.isfinite(hist['loss']).all() and np.isfinite(hist['loss_epoch_mean']).all()
    # fresh random batches every step: compare epoch means (hist['loss'] is the LAST batch's value, as in Keras)
    assert np.mean(hist['loss_epoch_mean'][-2:]) < 0.8 * hist['loss_epoch_mean'][0], hist['loss_epoch_mean']


def test_two_stream_backward_is_bitwise_identical_to_one_stream():
    """The weight gradients run on a second HIP stream (layers.Context): same kernels, same order per buffer, so the gradient bucket must be
    bit-identical to the single-stream run - any difference would be a race."""
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import SGD
    full = configs.hpnn_tiny()
    cfg = full['model']
    rhs, dx = make_inputs(3, 52, 44, 17)
    target = np.random.default_rng(4).standard_normal(rhs.shape) * 0.1
    grads = []
    for side in (True, False, True):
        model, _ = build(cfg, 41)
        model.ctx.use_side = side
        model.compile(loss=loss_wrapper(global_batch_size=3, **full['training']['loss_parameters']), optimizer=SGD(learning_rate=0.0))
        for _ in range(2):
            model.train_step(((rhs, dx), target))
        grads.append(model.store.flat_g.cpu().numpy().copy())
    assert np.array_equal(grads[0], grads[1]) and np.array_equal(grads[0], grads[2])


def test_branch_streams_are_bitwise_identical_to_one_stream_on_the_shipped_model():
    """Round 4: every bottleneck branch of hpnn.json runs its convolution stages on a stream of its own (models.Homogeneous_Poisson_NN_Legacy.call /
    backward: eight branches, pyramid-fed ones and the factor-3 branch that pools the full-resolution tensor itself and accumulates its input gradient).
    Same kernels, same accumulation order per buffer: predictions and the whole gradient bucket must be bit-identical to the single-stream run, step
    after step - any difference would be a missing event or a shared scratch buffer."""
    import os
    if os.environ.get('PCNN_BRANCH_STREAMS', '1') == '0':
        pytest.skip('the developer switch PCNN_BRANCH_STREAMS=0 turns the streams under test off')
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import SGD
    full = configs.hpnn()
    cfg = full['model']
    rhs, dx = make_inputs(2, 256, 256, 23)
    target = np.random.default_rng(5).standard_normal(rhs.shape) * 0.1
    runs = []
    for side in (True, False, True):
        model, _ = build(cfg, 43)
        model.ctx.use_side = side
        model.compile(loss=loss_wrapper(global_batch_size=2, **full['training']['loss_parameters']), optimizer=SGD(learning_rate=0.0))
        preds = []
        for _ in range(3):
            model.train_step(((rhs, dx), target))
            preds.append(model([rhs, dx]).cpu().numpy().copy())
        assert (len(model.ctx.branch_streams) > 0) == side
        runs.append((preds, model.store.flat_g.cpu().numpy().copy()))
    for other in runs[1:]:
        assert np.array_equal(runs[0][1], other[1])
        for a, b in zip(runs[0][0], other[0]):
            assert np.array_equal(a, b)


def test_branch_streams_with_main_stream_accumulators_on_a_non_divisible_grid():
    """with PCNN_COARSE_FACTOR > 2 the small-factor branches stay on the MAIN stream, and on a grid their factors do not divide they pool
    the full-resolution tensor themselves, i.e. their backward ADDS into the shared d_initial - as the stream-0 branches do.  Every accumulator must be
    ordered after the ones before it whatever stream it runs on: the gradient bucket must be bit-identical to the single-stream run."""
    import os
    if os.environ.get('PCNN_BRANCH_STREAMS', '1') == '0':
        pytest.skip('the developer switch PCNN_BRANCH_STREAMS=0 turns the streams under test off')
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import SGD
    full = configs.hpnn()
    cfg = full['model']
    rhs, dx = make_inputs(2, 250, 246, 29)                       # 250 x 246: factors 3, 4, 8, 16 ... do not divide both extents
    target = np.random.default_rng(6).standard_normal(rhs.shape) * 0.1
    for factor in (8, 4):                                        # per factor: streams on / off / on again (the order in which the branches' pooled gradients are
        runs = []                                                # summed depends on which branches run on streams, i.e. on the factor - not on the streams themselves)
        for side in (True, False, True):
            model, _ = build(cfg, 47)
            model.COARSE_FACTOR = factor
            model.ctx.use_side = side
            model.compile(loss=loss_wrapper(global_batch_size=2, **full['training']['loss_parameters']), optimizer=SGD(learning_rate=0.0))
            for _ in range(3):
                model.train_step(((rhs, dx), target))
            runs.append(model.store.flat_g.cpu().numpy().copy())
        for other in runs[1:]:
            assert np.array_equal(runs[0], other), factor


def test_channels_last_model_api():
    """Homogeneous_Poisson_NN_Legacy(data_format='channels_last'): (N,H,W,1) in, (N,H,W,1) out, identical numbers and an identical training
    step (the boundary tensors have one channel, so the two formats are the same memory)."""
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    from poisson_cnn_amd.train import Adam
    full = configs.hpnn_tiny()
    rhs, dx = make_inputs(2, 40, 44, 8)
    tgt = np.random.default_rng(1).standard_normal(rhs.shape).astype(np.float32) * 0.1
    res = {}
    for fmt in ('channels_first', 'channels_last'):
        m = Homogeneous_Poisson_NN_Legacy(**dict(full['model'], data_format=fmt), seed=5)
        m.compile(loss=loss_wrapper(global_batch_size=2, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-3))
        r, t = (rhs, tgt) if fmt == 'channels_first' else (rhs.transpose(0, 2, 3, 1), tgt.transpose(0, 2, 3, 1))
        y = m([r, dx])
        assert tuple(y.shape) == ((2, 1, 40, 44) if fmt == 'channels_first' else (2, 40, 44, 1))
        logs = m.train_step(((r, dx), t))
        res[fmt] = (y.cpu().numpy().reshape(2, 40, 44), float(logs['loss']), m.store.flat_w.cpu().numpy().copy())
    np.testing.assert_array_equal(res['channels_first'][0], res['channels_last'][0])
    assert res['channels_first'][1] == res['channels_last'][1]
    np.testing.assert_array_equal(res['channels_first'][2], res['channels_last'][2])


_SHAPE_STEP_SNIPPET = """
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from poisson_cnn_amd import configs
from poisson_cnn_amd.dataset import reverse_poisson_dataset_generator
from poisson_cnn_amd.losses import loss_wrapper
from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
from poisson_cnn_amd.train import Adam
state_in, state_out, step, H, W = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cfg = configs.hpnn()
model = Homogeneous_Poisson_NN_Legacy(**cfg['model'])
model.compile(loss=loss_wrapper(global_batch_size=50, **cfg['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-4))
if state_in != '-':
    st = np.load(state_in)
    model.store.flat_w.copy_(torch.from_numpy(st['w'])); model.store.flat_stats.copy_(torch.from_numpy(st['stats']))
    model.optimizer.m.copy_(torch.from_numpy(st['m'])); model.optimizer.v.copy_(torch.from_numpy(st['v'])); model.optimizer.iterations = int(st['it'])
    from poisson_cnn_amd import ops
    ops.weights_changed()
d = dict(cfg['dataset'], batches_per_epoch=1)
gen = reverse_poisson_dataset_generator(seed=100 + step, **d)
gen.fixed_output_shape = (H, W)
inp, tar = gen[0]
logs = model.train_step((tuple(inp), tar))
torch.cuda.synchronize()
np.savez(state_out, w=model.store.flat_w.cpu().numpy(), stats=model.store.flat_stats.cpu().numpy(), m=model.optimizer.m.cpu().numpy(),
         v=model.optimizer.v.cpu().numpy(), it=model.optimizer.iterations, loss=float(logs['loss']))
"""


def test_shipped_batch_over_changing_shapes_equals_one_step_per_fresh_process(tmp_path):
    """the shipped training configuration (experiments/hpnn.json: batch 50, a new grid shape every batch) through fit() - with the
    workspaces pre-sized at compile(), the next batch generated on its own stream while a step runs, one host round trip per step - for 6 steps over
    3 distinct shapes, against the SAME six steps run one per fresh process (nothing cached, nothing pre-sized, nothing prefetched; weights and Adam
    state handed over through files): the weights must be BIT-equal.  Per-shape state (tile tables, workspace growth, kept filter spectra, allocator
    reuse, stream hand-over) must never leak into the arithmetic."""
    import subprocess
    import sys
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.dataset import reverse_poisson_dataset_generator
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    from poisson_cnn_amd.train import Adam
    shapes = [(200, 232), (192, 301), (260, 196)] * 2
    cfg = configs.hpnn()

    class Seq:                                                            # the Keras Sequence fit() walks: step i = the generator seeded 100 + i at shapes[i]
        def __len__(self):
            return len(shapes)

        def __getitem__(self, i):
            gen = reverse_poisson_dataset_generator(seed=100 + i, **dict(cfg['dataset'], batches_per_epoch=1))
            gen.fixed_output_shape = shapes[i]
            return gen[0]
    model = Homogeneous_Poisson_NN_Legacy(**cfg['model'])
    model.compile(loss=loss_wrapper(global_batch_size=50, **cfg['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-4), max_input_shape=(50, 260, 301))
    w0 = model.store.flat_w.clone()
    hist = model.fit(Seq(), epochs=1, verbose=0)
    torch.cuda.synchronize()
    w_fit = model.store.flat_w.cpu().numpy()
    assert np.isfinite(w_fit).all() and not np.array_equal(w_fit, w0.cpu().numpy())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prev = '-'
    for i, (H, W) in enumerate(shapes):
        out = str(tmp_path / ('state%d.npz' % i))
        r = subprocess.run([sys.executable, '-c', _SHAPE_STEP_SNIPPET % root, prev, out, str(i), str(H), str(W)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        prev = out
    last = np.load(prev)
    assert int(last['it']) == 6
    assert np.array_equal(last['w'], w_fit), float(np.abs(last['w'] - w_fit).max())
    assert abs(float(last['loss']) - hist['loss'][-1]) == 0.0
