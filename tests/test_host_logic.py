"""CPU tests of host-side logic: parameter structure vs the oracle's, config handling, C-ABI symbol export."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import hpnn as ohpnn
from poisson_cnn_amd import configs, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('cfgname', ['hpnn', 'hpnn_tiny', 'hpnn_neumann'])
def test_parameter_structure_matches_oracle(cfgname):
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    cfg = getattr(configs, cfgname)()['model']
    model = Homogeneous_Poisson_NN_Legacy(device='cpu', **cfg)
    _, spec = ohpnn.build_structure(cfg)
    assert model.weight_names == [s[0] for s in spec]
    for (name, shape, _), w in zip(spec, model.get_weights()):
        assert tuple(shape) == w.shape, name
    if cfgname != 'hpnn_tiny':
        assert model.count_params() == 5556956      # 5 556 956 trainable parameters
    # Keras-default initialisation statistics
    w = dict(zip(model.weight_names, model.get_weights()))
    k = w['final/stage0/conv/kernel']
    lim = utils.glorot_limit(k.shape)
    assert np.abs(k).max() <= lim and np.abs(k).max() > 0.9 * lim
    assert np.all(w['final/stage0/conv/bias'] == 0) and np.all(w['pre/bn0/gamma'] == 1) and np.all(w['pre/bn0/moving_variance'] == 1)
    assert np.abs(w['deconv_f2/deconv/bias']).max() > 0     # deconv bias is Glorot-initialised (layers/deconvupscale.py:37-38)


def test_set_get_weights_roundtrip_and_errors(tmp_path):
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    cfg = configs.hpnn_tiny()['model']
    m = Homogeneous_Poisson_NN_Legacy(device='cpu', **cfg)
    p = ohpnn.init_params(cfg, seed=3, randomize_all=True)
    m.set_weights(p)
    for n, w in zip(m.weight_names, m.get_weights()):
        assert np.array_equal(w, p[n].astype(np.float32))
    m.save_weights(str(tmp_path / 'w.npz'))
    m2 = Homogeneous_Poisson_NN_Legacy(device='cpu', seed=9, **cfg)
    m2.load_weights(str(tmp_path / 'w.npz'))
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), m2.get_weights()))
    with pytest.raises(ValueError):
        m.set_weights(m.get_weights()[:-1])
    with pytest.raises(ValueError):
        Homogeneous_Poisson_NN_Legacy(device='cpu', **dict(cfg, bc_type='robin'))
    with pytest.raises(ValueError):
        Homogeneous_Poisson_NN_Legacy(device='cpu', **{k: v for k, v in cfg.items() if k != 'final_convolutions_config'})


def test_config_helpers():
    cfg = {'key1': 3, 'key2': [0, 1, 2, 3, 4], 'key3': [6, 7, 8, 9, 10]}
    assert utils.get_init_arguments_from_config(cfg, 2, ['key2', 'key3'], ['key2p', 'key3p']) == {'key1': 3, 'key2p': 2, 'key3p': 8}
    assert list(utils.split_indices(229, 4)) == [0, 58, 115, 172, 229]
    assert utils.canonical_activation('tf.nn.leaky_relu') == 'leaky_relu' and utils.canonical_activation('tf.keras.activations.linear') == 'linear'
    with pytest.raises(ValueError):
        utils.convert_tf_object_names({'a': 'tf.nn.swish'})
    c = utils.convert_tf_object_names(configs.hpnn())
    assert c['model']['pre_bottleneck_convolutions_config']['activation'] == 'tf.nn.leaky_relu'
    assert utils.advanced_pad_amounts(15) == (7, 7) and utils.advanced_pad_amounts(4) == (2, 1) and utils.same_pad_amounts(4) == (1, 2)


def test_libpcnn_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'pcnn.h')).read()
    declared = sorted(set(re.findall(r'\b(pcnn_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) > 40
    lib = ctypes.CDLL(os.path.join(ROOT, 'poisson_cnn_amd', 'libpcnn.so'))
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, 'declared in include/pcnn.h but not exported: %s' % missing
    assert lib.pcnn_version() >= 100


def test_every_entry_point_is_called_through_its_declared_prototype():
    """_lib.load() binds argtypes / restype of EVERY function declared in include/pcnn.h from the header text, so a Python int
    passed for an int64_t / size_t argument arrives whole, a value that does not fit raises, and so does a float where an integer is declared."""
    from poisson_cnn_amd import _lib
    protos = _lib.header_prototypes()
    hdr = open(os.path.join(ROOT, 'include', 'pcnn.h')).read()
    declared = sorted(set(re.findall(r'\b(pcnn_[a-z0-9_]+)\s*\(', re.sub(r'/\*.*?\*/', ' ', hdr, flags=re.S))))
    assert sorted(protos) == declared                                   # the prototype parser sees every declaration the export test sees
    lib = _lib.load()
    for name, (ret, types) in protos.items():
        fn = getattr(lib, name)
        assert fn.argtypes is not None and len(fn.argtypes) == len(types), name
    assert protos['pcnn_allreduce'] == ('int', ['pcnn_handle', 'float*', 'size_t'])
    assert protos['pcnn_conv2d_wgrad_workspace'] == ('size_t', ['const pcnn_conv_desc*'])
    # no GPU needed for the workspace-size queries: the same value however the integer is spelled, and loud failures instead of truncation
    assert lib.pcnn_colsum_workspace(32) == lib.pcnn_colsum_workspace(ctypes.c_int(32)) == lib.pcnn_colsum_workspace(ctypes.c_int64(32)) > 0
    assert lib.pcnn_channel_scale_workspace(2, 2 ** 33, 8) == lib.pcnn_channel_scale_workspace(2, 5, 8)   # an int64_t argument beyond 32 bits is accepted whole
    with pytest.raises(ctypes.ArgumentError):
        lib.pcnn_colsum_workspace(2 ** 40)
    with pytest.raises(ctypes.ArgumentError):
        lib.pcnn_colsum_workspace(3.5)
    assert lib.pcnn_crc32c(b'123456789', 9, 0) == 0xE3069283


def test_every_python_call_site_matches_the_header():
    """tools/check_ctypes_calls.py: argument count and the kind (integer / float / pointer) of every wrapped argument of every libpcnn call in
    poisson_cnn_amd/ against include/pcnn.h - static, so a call site that only a GPU run reaches is checked here too."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_ctypes_calls
    sites, problems = check_ctypes_calls.check()
    assert sites >= 100 and not problems, problems


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'poisson_cnn_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


def test_integral_weight_map_matches_oracle():
    import torch
    from oracle import loss as oloss
    from poisson_cnn_amd.losses import integral_loss, build_fd_coefficients
    from oracle import np_ops
    il = integral_loss(47, ndims=2)
    for H, W in [(40, 52), (128, 96), (513, 300)]:
        G = il.weight_map(H, W, torch.device('cpu')).numpy()
        ref = oloss.integral_weight_map((H, W), 47)
        assert np.allclose(G, ref, rtol=1e-6, atol=1e-9)
        assert abs(G.sum() - 4.0) < 1e-5          # integrates the constant 1 over [-1,1]^2
    assert np.allclose(build_fd_coefficients([5, 5], 2, 2), np_ops.build_fd_coefficients([5, 5], 2, 2))
    assert np.allclose(build_fd_coefficients(3, [2, 2], 2), np_ops.build_fd_coefficients([3, 3], [2, 2], 2))


def test_dataset_host_samplers():
    from poisson_cnn_amd import dataset as ds
    rng = np.random.default_rng(0)
    osr = [[192, 384], [192, 384]]
    ar = ds.generate_uniformly_distributed_aspect_ratios(osr, None, samples=2000, rng=rng)
    assert ar.shape == (2000, 1) and ar.min() >= 191 / 383 - 1e-9 and ar.max() <= 383 / 191 + 1e-9
    # roughly half of the aspect ratios fall under 1 for a symmetric range
    assert 0.4 < (ar < 1).mean() < 0.6
    for _ in range(20):
        a = ds.generate_uniformly_distributed_aspect_ratios(osr, None, samples=4, rng=rng)
        shape, dx = ds.generate_output_shapes_and_grid_spacings_from_aspect_ratios(a, osr, [[5e-3, 5e-2], [5e-3, 5e-2]], constant_dx=True, samples=4, rng=rng)
        assert shape.shape == (2,) and np.all(shape <= 384) and np.all(shape >= 96)
        assert dx.shape == (4, 2) and np.all(dx[:, 0] == dx[:, 1]) and dx.min() >= 5e-3 and dx.max() <= 5e-2
    n = ds._process_normalizations({'rhs_max_magnitude': True, 'max_domain_size_squared': True})
    assert n == {'rhs_max_magnitude': 1.0, 'max_domain_size_squared': True, 'soln_max_magnitude': False}
    with pytest.raises(AssertionError):
        ds._range2([3, 1], 2)


def test_channels_last_views_of_one_channel_tensors():
    """data_format='channels_last' at the model API is a reshape: (N,H,W,1) <-> (N,1,H,W), (N,L,1) <-> (N,1,L); dx arrays pass through."""
    import numpy as np
    import torch
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    m = Homogeneous_Poisson_NN_Legacy(**dict(configs.hpnn_tiny()['model'], data_format='channels_last'), device='cpu')
    x = torch.arange(2 * 5 * 7, dtype=torch.float32).reshape(2, 5, 7, 1)
    cf = m._cf(x)
    assert tuple(cf.shape) == (2, 1, 5, 7) and cf.data_ptr() == x.data_ptr()           # the same memory
    assert torch.equal(m._cl(cf), x)
    assert tuple(m._cf(np.zeros((3, 9, 1), dtype=np.float32)).shape) == (3, 1, 9)
    dx = torch.ones(2, 1)
    assert m._cf(dx) is dx
    with pytest.raises(ValueError):
        m._cf(torch.zeros(2, 5, 7, 3))
    m2 = Homogeneous_Poisson_NN_Legacy(**configs.hpnn_tiny()['model'], device='cpu')
    assert m2._cf(x) is x                                                                # channels_first: untouched
    with pytest.raises(ValueError):
        Homogeneous_Poisson_NN_Legacy(**dict(configs.hpnn_tiny()['model'], data_format='NHWC'), device='cpu')


def test_chain_models_structure_and_oracle_names():
    """Homogeneous_Poisson_NN_Metalearning / Homogeneous_Poisson_NN (hpnn_models.py): the parameter structure builds on the CPU from the
    train/hpnn_train.py-style configs, the oracle restatement consumes exactly those names (every trainable one reaches the output), and the
    constructor errors are the reference's."""
    import copy
    import pytest
    import torch
    from oracle import hpnn_chain as och
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.hpnn_models import Homogeneous_Poisson_NN, Homogeneous_Poisson_NN_Metalearning
    rng = np.random.default_rng(0)
    for cfg, cls, fn in ((configs.hpnn_metalearning_tiny(), Homogeneous_Poisson_NN_Metalearning, och.metalearning_forward),
                         (configs.hpnn_plain_tiny(), Homogeneous_Poisson_NN, och.plain_forward)):
        kw = dict(cfg['model'])
        kw.pop('model_type')
        model = cls(**copy.deepcopy(kw), device='cpu')
        blocks = [b.f for b in model.bottleneck_blocks]
        assert blocks == sorted(blocks, reverse=True)
        p = {n: torch.tensor(model.store.w[n].numpy().astype(np.float64) * 1.3 + (0.1 if n.endswith('variance') else 0.0),
                             requires_grad=not n.endswith(('moving_mean', 'moving_variance'))) for n in model.store.names}
        rhs, dx = torch.tensor(rng.uniform(-1, 1, (2, 1, 24, 24))), torch.tensor(rng.uniform(5e-3, 5e-2, (2, 2)))
        y = fn(p, kw, rhs, dx)
        assert tuple(y.shape) == (2, 1, 24, 24)
        y.sum().backward()
        assert [n for n in model.store.trainable_names() if p[n].grad is None] == []
        with pytest.raises(ValueError):
            cls(**{**copy.deepcopy(kw), 'bottleneck_config': None}, device='cpu')
        with pytest.raises(ValueError):
            cls(**{**copy.deepcopy(kw), 'bottleneck_upsampling': 'bicubic'}, device='cpu')
    # use_bias reaches every Dense layer of the hyper-networks (layers/metalearning_conv.py:113,128): the example configuration sets it False
    # everywhere, so no Dense bias exists and the first layer emits exactly the 19 x 19 x 3 x 4 kernel (hand count: 4*8 + 8*16 + 16*4332 weights)
    big = {k: v for k, v in configs.hpnn_metalearning()['model'].items() if k != 'model_type'}
    m0 = Homogeneous_Poisson_NN_Metalearning(**copy.deepcopy(big), device='cpu')
    assert [n for n in m0.store.names if n.endswith('/bias')] == []
    assert tuple(m0.store.w['pre/conv0/dense2/kernel'].shape) == (16, 19 * 19 * 3 * 4)
    assert m0.count_params() == 5317256
    big['pre_bottleneck_convolutions_config']['use_bias'] = True
    m1 = Homogeneous_Poisson_NN_Metalearning(**copy.deepcopy(big), device='cpu')
    assert tuple(m1.store.w['pre/conv0/dense2/kernel'].shape) == (16, 19 * 19 * 3 * 4 + 4)
    assert [tuple(m1.store.w['pre/conv0/dense%d/bias' % i].shape) for i in range(3)] == [(8,), (16,), (19 * 19 * 3 * 4 + 4,)]
    # three layers gain (8 + 16 + n_out) biases and 16 more kernel columns per output channel
    extra = sum(8 + 16 + (k * k * ci * co + co) + 16 * co for k, ci, co in ((19, 3, 4), (17, 4, 6), (15, 6, 8)))
    assert m1.count_params() == 5317256 + extra


def test_rccl_load_failure_is_an_error_return_not_a_crash():
    """a missing librccl must come back as a non-zero return code with a message (csrc/collective.hip), never as a crash -
    run in a child process so that a segfault would be seen as a failed return code, with the library name forced to a missing file."""
    import subprocess
    import sys
    code = ('import ctypes, os\n'
            'from poisson_cnn_amd import _lib\n'
            'lib = _lib.load()\n'
            'buf = ctypes.create_string_buffer(512)\n'
            'rc = lib.pcnn_collective_available(buf, ctypes.c_size_t(512))\n'
            'rc2 = lib.pcnn_collective_available(buf, ctypes.c_size_t(512))\n'
            'print(rc, rc2, buf.value.decode())\n')
    env = dict(os.environ, PCNN_RCCL_LIBRARY='/nonexistent/librccl_missing.so')
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rc, rc2, msg = r.stdout.strip().split(' ', 2)
    assert rc == '1' and rc2 == '1'
    assert 'librccl_missing.so' in msg and len(msg) > 30


def test_weights_version_bookkeeping_of_the_filter_cache():
    """Host side of pcnn_set_filter_version (no GPU): the layer classes' weights version moves on every torch-side write to a parameter bucket
    (Tensor._version), on the library's own raw-pointer writers (they call ops.weights_changed) and when a bucket dies; with the cache switched off the
    version handed to the library is 0, i.e. nothing is ever kept."""
    import gc
    import torch
    from poisson_cnn_amd import layers, ops
    store = layers.ParamStore()
    store.add('a/kernel', (3, 3, 2, 4), 'glorot')
    store.add('a/bias', (4,), 'zeros')
    store.finalize(torch.device('cpu'))
    store.initialize(seed=1)
    v0 = store.filter_version()
    assert v0 == ops.filter_version() and v0 > 0
    assert store.filter_version() == v0                                   # nothing written: the version stands
    store.w['a/kernel'].mul_(2.0)                                         # a view of the bucket, written through torch
    v1 = store.filter_version()
    assert v1 > v0 and store.filter_version() == v1
    store.flat_w.add_(1.0)
    assert store.filter_version() > v1
    v2 = ops.filter_version()
    ops.weights_changed()                                                 # what train.Adam / SGD and parallel.attach call after their raw-pointer writes
    assert ops.filter_version() == v2 + 1 and store.filter_version() == v2 + 1
    ops.set_filter_cache(False)
    try:
        assert ops.filter_version() == 0 and store.filter_version() == 0
    finally:
        ops.set_filter_cache(True)
    epoch, v3 = ops._filter_epoch, ops.filter_version()
    del store
    gc.collect()
    assert ops._filter_epoch == epoch + 1 and ops.filter_version() > v3    # a dying bucket: every handle empties its cache before its next cached call


def test_host_logs_and_inline_prefetcher_on_cpu():
    """fit()'s helpers without a GPU: _host_logs turns a dict of device scalars / floats into floats through one stacked read (order and keys kept);
    _Prefetcher on a CPU device generates in line - dataset[i] is called exactly once per request, in order."""
    import torch
    from poisson_cnn_amd import models
    logs = {'loss': torch.tensor(1.5), 'mse': torch.tensor([0.25]), 'lr': 1e-3}
    out = models._host_logs(logs)
    assert list(out) == ['loss', 'mse', 'lr'] and out == {'loss': 1.5, 'mse': 0.25, 'lr': 1e-3} and all(isinstance(v, float) for v in out.values())

    class Seq:
        def __init__(self):
            self.calls = []

        def __getitem__(self, i):
            self.calls.append(i)
            return [torch.zeros(1)], torch.zeros(1)
    s = Seq()
    f = models._Prefetcher(s, 'cpu')
    assert not f.on
    for i in range(3):
        f.request(i)
        f.take()
    f.close()
    assert s.calls == [0, 1, 2]
