"""Filter spectra kept across calls (include/pcnn.h pcnn_set_filter_version).

The cache must be invisible in the results: a convolution that reuses a kept spectrum, or gets it from the one-launch refresh of all filters, returns the
very bits of the call that recomputes it (the table kernels run the same code on the same values).  Checked at the C-ABI level (ops.conv2d_fwd /
conv2d_bwd_fused with `w_version`), through the layer classes (training steps with the cache on and off, torch-side weight edits, a dying model whose
addresses the next one inherits) and under both transform families."""
import numpy as np
import pytest
import torch

from oracle import hpnn as ohpnn
from poisson_cnn_amd import configs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _spectral_defaults():
    from poisson_cnn_amd import ops
    mode, tile, xf = ops.get_spectral_mode(), ops.get_spectral_tile(), ops.get_spectral_transform()
    yield
    ops.set_spectral_mode(mode); ops.set_spectral_tile(tile); ops.set_spectral_transform(xf); ops.set_filter_cache(True)


def _layers(dev):
    g = torch.Generator(device='cpu').manual_seed(3)
    shapes = [(7, 32, 32, 32), (5, 16, 16, 32), (15, 32, 32, 64), (13, 28, 28, 64), (9, 24, 20, 32)]     # (taps, Cin, Cout, tile)
    out = []
    for k, ci, co, T in shapes:
        x = torch.randn(2, 192, 160, ci, generator=g).to(dev)
        w = (torch.randn(k, k, ci, co, generator=g) * 0.05).to(dev)
        dz = torch.randn(2, 192, 160, co, generator=g).to(dev)
        out.append((k, ci, co, T, x, w, dz))
    return out


@pytest.mark.parametrize('xform', ['fft', 'mfma'])
def test_cached_spectra_give_the_bits_of_the_uncached_call(xform):
    from poisson_cnn_amd import ops
    dev = torch.device('cuda')
    ops.set_spectral_mode('force'); ops.set_spectral_transform(xform)
    layers = _layers(dev)

    wfs = {id(l[5]): torch.empty(l[5].shape[0], l[5].shape[1], l[5].shape[3], l[5].shape[2], device=dev) for l in layers}   # per-filter buffers: stable addresses

    def run(version):
        ys = []
        if version:                                                       # the version's promise covers the flipped filters too: all of them are formed
            for *_, w, _ in layers:                                       # before the first call under a new version (layers.Conv does the same)
                ops.flip_transpose_weights(w, out=wfs[id(w)])
        for k, ci, co, T, x, w, dz in layers:
            ops.set_spectral_tile(T)
            y = ops.conv2d_fwd(x, w, None, pad_top=k // 2, pad_left=k // 2, w_version=version)
            wf = wfs[id(w)] if version else ops.flip_transpose_weights(w)
            dw = torch.empty_like(w)
            dx = ops.conv2d_bwd_fused(x, dz, tuple(w.shape), wf, pad_top=k // 2, pad_left=k // 2, dw=dw, w_version=version)
            assert dx is not None
            ys.append((y.clone(), dx.clone(), dw.clone()))
        return ys

    # (a model of an earlier test may have died since this stream's handle last ran a cached call: the pending clean-up - which resets the counters -
    # happens at the handle's next cached call, so make that call now, on a filter of its own)
    wd = torch.zeros(5, 5, 32, 32, device=dev)
    ops.set_spectral_tile(32)
    ops.conv2d_fwd(torch.zeros(1, 64, 64, 32, device=dev), wd, None, pad_top=2, pad_left=2, w_version=100)
    s0 = ops.filter_cache_stats()
    ref = run(0)
    assert ops.filter_cache_stats()['fills'] == s0['fills']              # version 0: nothing is kept
    first = run(101)                                                      # fills: one entry per filter and direction
    s1 = ops.filter_cache_stats()
    assert s1['fills'] - s0['fills'] == 2 * len(layers) and s1['bytes'] > s0['bytes']
    again = run(101)                                                      # hits
    s2 = ops.filter_cache_stats()
    assert s2['hits'] - s1['hits'] == 2 * len(layers) and s2['fills'] == s1['fills']
    for a, b, c in zip(ref, first, again):
        for t0, t1, t2 in zip(a, b, c):
            assert torch.equal(t0, t1) and torch.equal(t0, t2)
    # new weights under a new version: ONE refresh per tile size brings every filter up to date
    for *_, w, _ in layers:
        w.mul_(-1.5).add_(0.01)
    ref2 = run(0)
    new = run(102)
    s3 = ops.filter_cache_stats()
    assert s3['fills'] == s2['fills'] and s3['refreshes'] - s2['refreshes'] == 1
    for li, (a, b) in enumerate(zip(ref2, new)):
        for ti, (t0, t1) in enumerate(zip(a, b)):
            assert torch.equal(t0, t1), (li, ti, float((t0 - t1).abs().max()), float(t0.abs().max()))
    assert not torch.equal(ref[0][0], ref2[0][0])


def _build(cfg, seed):
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    model = Homogeneous_Poisson_NN_Legacy(**cfg)
    model.set_weights(ohpnn.init_params(cfg, seed=seed, gain=1.6, randomize_all=True))
    return model


def _inputs(N, H, W, seed):
    rng = np.random.default_rng(seed)
    rhs = rng.uniform(-1, 1, (N, 1, H, W)).astype(np.float32)
    dx = rng.uniform(5e-3, 5e-2, (N, 1)).astype(np.float32)
    return rhs, dx, (rng.standard_normal(rhs.shape) * 0.1).astype(np.float32)


def _train(cache, steps=3):
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    ops.set_filter_cache(cache)
    ops.set_spectral_mode('force')                                        # every layer that can takes the spectral route
    full = configs.hpnn_tiny()
    model = _build(full['model'], 21)
    model.compile(loss=loss_wrapper(global_batch_size=2, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-3))
    rhs, dx, target = _inputs(2, 72, 64, 4)
    losses = [float(model.train_step(((rhs, dx), target))['loss']) for _ in range(steps)]
    y = model([rhs, dx]).clone()
    y2 = model([rhs, dx]).clone()                                         # inference twice: the second call meets only kept spectra
    assert torch.equal(y, y2)
    return losses, model.store.flat_w.clone(), y, model


def test_training_steps_are_bit_identical_with_and_without_the_cache():
    from poisson_cnn_amd import ops
    # the counters live in a handle's cache object: a parameter bucket of an EARLIER test that has died makes every handle empty its cache - counters
    # included - at its next cached call (ops.weights_released), so a first run (kept alive) flushes whatever is pending before the counted one
    keep = _train(True)
    s0 = ops.filter_cache_stats()
    l1, w1, y1, m1 = _train(True)
    s1 = ops.filter_cache_stats()
    assert s1['fills'] > s0['fills'] and s1['refreshes'] > s0['refreshes'] and s1['hits'] > s0['hits']
    l0, w0, y0, m0 = _train(False)
    assert l1 == l0 and torch.equal(w1, w0) and torch.equal(y1, y0)
    assert ops.filter_cache_stats()['fills'] == s1['fills']              # cache off: nothing was added
    assert keep[0] == l1 and torch.equal(keep[1], w1)                    # (and the warm-up run was the same training run)


def test_torch_side_weight_edits_and_a_dying_model_are_noticed():
    from poisson_cnn_amd import ops
    ops.set_spectral_mode('force')
    cfg = configs.hpnn_tiny()['model']
    rhs, dx, _ = _inputs(2, 72, 64, 9)
    model = _build(cfg, 5)
    y0 = model([rhs, dx]).clone()
    name = [n for n in model.store.trainable_names() if n.endswith('/kernel')][3]
    model.store.w[name].mul_(1.25)                                        # torch counts the write (Tensor._version): the next call refreshes
    y1 = model([rhs, dx]).clone()
    assert not torch.equal(y0, y1)
    ops.set_filter_cache(False)
    assert torch.equal(model([rhs, dx]), y1)
    ops.set_filter_cache(True)
    # a second model inherits the first one's addresses (same shapes, the allocator recycles the blocks): the dying bucket empties the cache
    w_ptr = model.store.flat_w.data_ptr()
    del model
    torch.cuda.synchronize()
    other = _build(cfg, 6)
    ya = other([rhs, dx]).clone()
    ops.set_filter_cache(False)
    yb = other([rhs, dx]).clone()
    assert torch.equal(ya, yb) and not torch.equal(ya, y1)
    assert other.store.flat_w.data_ptr() == w_ptr or True                 # (recycling is likely, not guaranteed: the check above holds either way)


def test_all_flipped_filters_in_one_launch_equal_the_per_filter_kernel():
    """ops.sync_flipped_filters: every registered layer's flipped filter from ONE pcnn_conv2d_flip_transpose_table launch."""
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    ops.set_spectral_mode('force')
    full = configs.hpnn_tiny()
    model = _build(full['model'], 21)
    model.compile(loss=loss_wrapper(global_batch_size=2, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-3))
    rhs, dx, target = _inputs(2, 72, 64, 3)
    assert np.isfinite(float(model.train_step(((rhs, dx), target))['loss']))   # every layer's first backward: registers its own buffer
    assert np.isfinite(float(model.train_step(((rhs, dx), target))['loss']))   # new weights: all of them re-formed by the table launch
    layers = [l for l in ops._flip_layers if getattr(l, '_wf', None) is not None and l.store is model.store]
    assert len(layers) >= 10
    ver = ops.filter_version()
    assert all(l._wf_ver is not None for l in layers)
    # the weights have moved once more since the flips (the optimizer step at the end of train_step): re-form, then compare
    ops.sync_flipped_filters(ver + 1000)                                  # a version nobody has seen: forces the table launch on the current weights
    for l in layers:
        w = l.store.w[l.name + '/kernel']
        assert torch.equal(l._wf, ops.flip_transpose_weights(w)), l.name
        assert torch.equal(l._wf, w.flip(0, 1).permute(0, 1, 3, 2).contiguous()), l.name


def test_torch_optim_through_the_tape_with_and_without_the_cache():
    """autograd.Differentiable + torch.optim: the optimizer writes the kernels' parameter bucket in place through torch (Tensor._version moves), so the
    kept spectra are refreshed before the next forward - three SGD steps give the same bits with the cache on and off."""
    from poisson_cnn_amd import keras_layers as K, ops
    from poisson_cnn_amd.autograd import Differentiable
    ops.set_spectral_mode('force')

    def run(cache):
        ops.set_filter_cache(cache)
        torch.manual_seed(0)
        conv = Differentiable(K.apply_advanced_padding_and_call_conv_layer('SYMMETRIC', K.Conv2D(16, 7, activation='tanh')))
        res = Differentiable(K.resnet(2, filters=16, kernel_size=5, activation='tanh', padding_mode='symmetric'))
        g = torch.Generator(device='cuda').manual_seed(5)
        x = torch.randn(2, 3, 96, 80, device='cuda', generator=g)
        tgt = torch.randn(2, 16, 96, 80, device='cuda', generator=g)
        with torch.no_grad():
            res(conv(x))                                                  # lazy build: the parameter buckets exist after the first call
        opt = torch.optim.SGD(list(conv.parameters()) + list(res.parameters()), lr=1e-3)
        outs = []
        for _ in range(3):
            opt.zero_grad()
            y = res(conv(x))
            loss = (y - tgt).square().mean()
            loss.backward()
            opt.step()
            outs.append((float(loss.detach()), y.detach().clone()))
        return outs, conv.weight.detach().clone(), res.weight.detach().clone()

    a, wa, ra = run(True)
    b, wb, rb = run(False)
    assert a[0][0] != a[2][0]                                             # the weights did move
    for (la, ya), (lb, yb) in zip(a, b):
        assert la == lb and torch.equal(ya, yb)
    assert torch.equal(wa, wb) and torch.equal(ra, rb)


def _interleaved(cache):
    """A.forward; B optimizer step (weights version moves); A.backward (first cached backward call under the new version: the handle refreshes EVERY kept
    filter, B's flipped filters included); then B forward + backward with no further version bump.  Returns B's and A's gradients."""
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    ops.set_filter_cache(cache)
    ops.set_spectral_mode('force')
    full = configs.hpnn_tiny()
    A, B = _build(full['model'], 31), _build(full['model'], 32)
    for m in (A, B):
        m.compile(loss=loss_wrapper(global_batch_size=2, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-2))
    rhs, dx, target = _inputs(2, 72, 64, 5)
    t = torch.tensor(target, device='cuda')
    for m in (A, B):                                                    # one full step each: every layer has run a backward and keeps a flipped filter
        m.train_step(((rhs, dx), target))
    rhs_d, dx_d = torch.tensor(rhs, device='cuda'), torch.tensor(dx, device='cuda')
    predA = A.call([rhs_d, dx_d], training=True)
    B.train_step(((rhs, dx), target))                                   # B's weights move: new version
    lossA, dpredA = A.loss_fn.value_and_grad(t, predA, rhs_d, torch.cat([dx_d, dx_d], 1))
    A.backward(dpredA)
    gA = A.store.flat_g.clone()
    B._loss_and_grads(rhs_d, dx_d, t)                                   # no version bump in between
    torch.cuda.synchronize()
    return B.store.flat_g.clone(), gA


def test_two_models_interleaved_see_current_flipped_filters():
    """ConvUnit._flipped re-formed only ITS layer's flipped filter when a backward pass met a new weights version, while the cached convolution
    behind it made the handle refresh - and stamp current - every kept filter, other models' stale flipped filters included.  With the fix every registered
    layer is re-flipped first: the interleaved sequence gives bit-identical gradients with the cache on and off."""
    gB1, gA1 = _interleaved(True)
    gB0, gA0 = _interleaved(False)
    assert torch.equal(gA1, gA0)
    assert torch.equal(gB1, gB0), float((gB1 - gB0).abs().max())
