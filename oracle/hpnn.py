"""CPU oracle: restatement of Homogeneous_Poisson_NN_Legacy (models/Homogeneous_Poisson_NN_Legacy.py:10-257)
and the blocks/layers it composes, on top of an op namespace (``oracle.np_ops`` for the fp64 numpy
oracle, ``oracle.torch_twin`` for the fp64 autograd twin used to check gradients).

TEST INFRASTRUCTURE ONLY - see oracle/np_ops.py header.  PARITY UNPINNED (no TF here).

Parameters live in an ordered dict name -> array.  The order restates the attribute-assignment order
of the reference constructors (Keras tracks sub-layers in that order):
  pre-bottleneck convs (+BN) -> deconv bottleneck blocks (descending factor) -> multilinear blocks
  (descending factor) -> non_bottleneck_conv -> post_merge_conv -> post_merge_resnet -> final
  convolutions -> dx dense layers -> Scaling (convs, dense_0..2).
"""
import copy
from collections import OrderedDict

import numpy as np


def get_init_arguments_from_config(cfg, k, fields_in_cfg, fields_in_args):
    """models/Homogeneous_Poisson_NN_Metalearning.py:10-25."""
    out = {key: cfg[key] for key in cfg if key not in fields_in_cfg}
    out.update({a: cfg[c][k] for a, c in zip(fields_in_args, fields_in_cfg)})
    return out


# --------------------------------------------------------------------------- structure
def _conv_spec(name, k, cin, cout, bias_init='zeros'):
    return [(name + '/kernel', (k, k, cin, cout), 'glorot'), (name + '/bias', (cout,), bias_init)]


def _bn_spec(name, c):
    return [(name + '/gamma', (c,), 'ones'), (name + '/beta', (c,), 'zeros'),
            (name + '/moving_mean', (c,), 'zeros'), (name + '/moving_variance', (c,), 'ones')]


def _resnet_spec(name, k, c, use_bn):
    s = []
    for i in range(3):
        s += _conv_spec('%s/conv%d' % (name, i), k, c, c)
    if use_bn:
        s += _bn_spec(name + '/bn0', c) + _bn_spec(name + '/bn1', c)
    return s


def _bottleneck_spec(name, cin, filters, k, n_convs, use_bn, deconv_k=None):
    # blocks/bottleneck_block.py:35-55 (pool + use_resnet): first conv, then n_convs-1 resnets
    s = _conv_spec(name + '/conv0', k, cin, filters)
    for i in range(n_convs - 1):
        s += _resnet_spec('%s/res%d' % (name, i), k, filters, use_bn)
    if deconv_k is not None:  # layers/deconvupscale.py:58-62: kernel (k,k,filters,in); default (Glorot) bias init
        s += [(name + '/deconv/kernel', (deconv_k, deconv_k, filters, filters), 'glorot'),
              (name + '/deconv/bias', (filters,), 'glorot')]
    return s


def build_structure(cfg):
    """Returns (meta, spec) for a model config (the "model" section of experiments/hpnn.json).
    meta: plain dict describing each stage; spec: ordered list of (name, shape, init)."""
    cfg = copy.deepcopy(cfg)
    use_bn = cfg.get('use_batchnorm', False)
    use_pos = cfg.get('use_positional_embeddings', True)
    meta = {'use_bn': use_bn, 'use_pos': use_pos, 'bc_type': cfg.get('bc_type', 'dirichlet').lower(),
            'postsmoother_iterations': cfg.get('postsmoother_iterations', 5),
            'use_scaling': cfg.get('use_scaling', False)}
    if meta['bc_type'] not in ('dirichlet', 'neumann'):
        raise ValueError('bc_type can only be neumann or dirichlet.')
    spec = []
    # pre-bottleneck (models/Homogeneous_Poisson_NN_Legacy.py:41-57)
    pre = cfg['pre_bottleneck_convolutions_config']
    meta['pre'] = {'padding_mode': pre.get('padding_mode', 'CONSTANT'), 'pad_value': pre.get('constant_padding_value', 0.0),
                   'activation': pre.get('activation', 'linear'), 'layers': []}
    cin = 3 if use_pos else 1
    for i, (f, k) in enumerate(zip(pre['filters'], pre['kernel_sizes'])):
        spec += _conv_spec('pre/conv%d' % i, k, cin, f)
        if use_bn:
            spec += _bn_spec('pre/bn%d' % i, f)
        meta['pre']['layers'].append((k, cin, f))
        cin = f
    c0 = cin
    # bottleneck blocks (:59-69)
    dcfg, mcfg = cfg['bottleneck_deconv_config'], cfg['bottleneck_multilinear_config']
    assert dcfg['filters'] == mcfg['filters']
    F = dcfg['filters']
    meta['filters'] = F
    blocks = []
    for i, f in enumerate(dcfg['downsampling_factors']):
        blocks.append({'kind': 'deconv', 'factor': f, 'up': dcfg['upsampling_factors'][i], 'k': dcfg['conv_kernel_sizes'][i],
                       'deconv_k': dcfg['deconv_kernel_sizes'][i], 'n_convs': dcfg['n_convs'][i],
                       'padding_mode': dcfg.get('padding_mode', 'constant'), 'pad_value': dcfg.get('constant_padding_value', 0.0),
                       'activation': dcfg.get('conv_activation', 'linear'), 'pool': dcfg.get('pool_downsampling_method', 'max'),
                       'name': 'deconv_f%d' % f})
    dblocks = sorted(blocks, key=lambda b: b['factor'], reverse=True)
    blocks = []
    for i, f in enumerate(mcfg['downsampling_factors']):
        blocks.append({'kind': 'multilinear', 'factor': f, 'up': mcfg['upsampling_factors'][i], 'k': mcfg['conv_kernel_sizes'][i],
                       'n_convs': mcfg['n_convs'][i], 'padding_mode': mcfg.get('padding_mode', 'constant'),
                       'pad_value': mcfg.get('constant_padding_value', 0.0), 'activation': mcfg.get('conv_activation', 'linear'),
                       'pool': mcfg.get('pool_downsampling_method', 'max'),
                       'resize_method': mcfg['resize_methods'][i] if 'resize_methods' in mcfg else 'bilinear',
                       'name': 'multilinear_f%d' % f})
    mblocks = sorted(blocks, key=lambda b: b['factor'], reverse=True)
    for b in dblocks + mblocks:
        src = dcfg if b['kind'] == 'deconv' else mcfg
        if src.get('downsampling_method', 'conv') != 'pool' or not src.get('use_resnet', False):
            raise NotImplementedError('oracle covers the shipped pool + resnet bottleneck configuration')
        spec += _bottleneck_spec(b['name'], c0, F, b['k'], b['n_convs'], use_bn, b.get('deconv_k'))
    meta['blocks'] = dblocks + mblocks
    # merge + post merge (:71-76)
    spec += _conv_spec('non_bottleneck_conv', 5, c0, F)
    spec += _conv_spec('post_merge_conv', 7, 2 * F, F)
    spec += _resnet_spec('post_merge_resnet', 7, F, False)
    # final convolutions (:78-96)
    fc = cfg['final_convolutions_config']
    nreg = fc.get('final_regular_conv_stages', 2)
    meta['final'] = {'padding_mode': fc.get('padding_mode', 'CONSTANT'), 'pad_value': fc.get('constant_padding_value', 0.0),
                     'activation': fc.get('activation', 'linear'), 'stages': [], 'tail': [], 'use_bias': fc.get('use_bias', True)}
    cin = F
    nst = len(fc['filters'])
    for i in range(nst - nreg):
        f, k = fc['filters'][i], fc['kernel_sizes'][i]
        spec += _conv_spec('final/stage%d/conv' % i, k, cin, f)
        spec += _resnet_spec('final/stage%d/res' % i, k, f, False)
        meta['final']['stages'].append((k, cin, f))
        cin = f
    for j, i in enumerate(range(nst - nreg, nst)):
        f, k = fc['filters'][i], fc['kernel_sizes'][i]
        spec += _conv_spec('final/out%d' % j, k, cin, f)
        meta['final']['tail'].append((k, cin, f))
        cin = f
    # dx dense layers (:98-102)
    units = [100, 100, F]
    din = 3
    for i, u in enumerate(units):
        spec += [('dx_dense%d/kernel' % i, (din, u), 'glorot'), ('dx_dense%d/bias' % i, (u,), 'zeros')]
        din = u
    # scaling (layers/Scaling.py:18-34)
    if meta['use_scaling']:
        sc = cfg['scaling_config']
        meta['scaling'] = {'stages': sc.get('stages', 2), 'ratio': sc.get('downsampling_ratio_per_stage', 2), 'filters': sc['filters'],
                           'k': sc['kernel_size'], 'activation': sc.get('activation', 'linear'),
                           'spp_levels': sc.get('spp_levels', [[2, 2], 3, 5])}
        cin = 2
        for i in range(meta['scaling']['stages']):
            spec += _conv_spec('scaling/conv%d' % i, sc['kernel_size'], cin, sc['filters'])
            cin = sc['filters']
        nfeat = 0
        for lv in meta['scaling']['spp_levels']:
            lv = [lv, lv] if isinstance(lv, int) else (list(lv) * 2 if len(lv) == 1 else list(lv))
            nfeat += lv[0] * lv[1]
        din = nfeat
        for i, u in enumerate([100, 25, 1]):
            spec += [('scaling/dense%d/kernel' % i, (din, u), 'glorot'), ('scaling/dense%d/bias' % i, (u,), 'zeros')]
            din = u
    return meta, spec


def glorot_limit(shape):
    """Keras _compute_fans + GlorotUniform: limit = sqrt(6 / (fan_in + fan_out))."""
    if len(shape) == 1:
        fi = fo = shape[0]
    elif len(shape) == 2:
        fi, fo = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fi, fo = shape[-2] * rf, shape[-1] * rf
    return float(np.sqrt(6.0 / (fi + fo)))


def init_params(cfg, seed=0, gain=1.0, randomize_all=False):
    """Keras-default initialisation (Glorot-uniform kernels, zero biases, BN 1/0/0/1) from numpy default_rng(seed).
    randomize_all=True additionally perturbs biases and BN parameters/statistics so tests exercise every term."""
    _, spec = build_structure(cfg)
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shape, init in spec:
        if init == 'glorot':
            lim = glorot_limit(shape) * gain
            v = rng.uniform(-lim, lim, size=shape)
        elif init == 'zeros':
            v = rng.uniform(-0.1, 0.1, size=shape) if randomize_all else np.zeros(shape)
        elif init == 'ones':
            v = rng.uniform(0.6, 1.4, size=shape) if randomize_all else np.ones(shape)
        else:
            raise ValueError(init)
        p[name] = v.astype(np.float32).astype(np.float64)  # weights are fp32 values
    return p


def trainable_names(spec_or_params):
    names = [s[0] if isinstance(s, tuple) else s for s in spec_or_params]
    return [n for n in names if not (n.endswith('moving_mean') or n.endswith('moving_variance'))]


# --------------------------------------------------------------------------- forward graph
def _bn(ops, p, name, x, training):
    if training:
        return ops.batchnorm_training(x, p[name + '/gamma'], p[name + '/beta'])[0]
    return ops.batchnorm_inference(x, p[name + '/gamma'], p[name + '/beta'], p[name + '/moving_mean'], p[name + '/moving_variance'])


def _pconv(ops, p, name, x, mode, value, act):
    return ops.padded_conv2d(x, p[name + '/kernel'], p[name + '/bias'], mode, value, act)


def resnet_forward(ops, p, name, x, mode, value, act, use_bn, bn_training=False):
    """blocks/resnet.py:29-39."""
    o = _pconv(ops, p, name + '/conv0', x, mode, value, act)
    if use_bn:
        o = _bn(ops, p, name + '/bn0', o, bn_training)
    o = _pconv(ops, p, name + '/conv1', o, mode, value, act)
    if use_bn:
        o = _bn(ops, p, name + '/bn1', o, bn_training)
    o = x + o
    return _pconv(ops, p, name + '/conv2', o, mode, value, act)


def bottleneck_forward(ops, p, b, x, use_bn, bn_training=False):
    """blocks/bottleneck_block.py:69-86 (multilinear) and :100-118 (deconv)."""
    H, W = x.shape[2], x.shape[3]
    o = ops.pool2d_same(x, b['factor'], b['pool'])
    o = _pconv(ops, p, b['name'] + '/conv0', o, b['padding_mode'], b['pad_value'], b['activation'])
    for i in range(b['n_convs'] - 1):
        o = resnet_forward(ops, p, '%s/res%d' % (b['name'], i), o, b['padding_mode'], b['pad_value'], b['activation'], use_bn, bn_training)
    out_hw = (int((H / b['factor']) * b['up']), int((W / b['factor']) * b['up']))  # bottleneck_block.py:82,109
    if b['kind'] == 'deconv':
        return ops.conv2d_transpose_same(o, p[b['name'] + '/deconv/kernel'], p[b['name'] + '/deconv/bias'], out_hw, b['up'], 'linear')
    return ops.resize2d(o, out_hw, b['resize_method'])


def scaling_forward(ops, p, sc, x_to_scale, other):
    """layers/Scaling.py:36-55."""
    o = ops.concat([x_to_scale, other], 1)
    for i in range(sc['stages']):
        o = ops.same_conv2d(o, p['scaling/conv%d/kernel' % i], p['scaling/conv%d/bias' % i], sc['activation'])
        o = ops.pool2d_same(o, sc['ratio'], 'average')
    o = ops.spatial_pyramid_pool(o, sc['spp_levels'], 'max')
    o = ops.dense(o, p['scaling/dense0/kernel'], p['scaling/dense0/bias'], 'leaky_relu')
    o = ops.dense(o, p['scaling/dense1/kernel'], p['scaling/dense1/bias'], 'leaky_relu')
    o = ops.dense(o, p['scaling/dense2/kernel'], p['scaling/dense2/bias'], 'linear')
    return x_to_scale * (1.0 + o)[:, :, None, None]


def position_embeddings(ops, N, H, W):
    """generate_position_embeddings (models/Homogeneous_Poisson_NN_Legacy.py:172-180): cos(pi*linspace(0,1,n)) per axis."""
    ey = np.cos(np.pi * np.linspace(0.0, 1.0, H))[:, None] * np.ones((1, W))
    ex = np.ones((H, 1)) * np.cos(np.pi * np.linspace(0.0, 1.0, W))[None, :]
    e = np.broadcast_to(np.stack([ey, ex], 0)[None], (N, 2, H, W)).copy()
    return ops.asarray(e)


def forward(ops, cfg, p, rhs, dx, bn_training=False, taps=None):
    """Homogeneous_Poisson_NN_Legacy.call (models/Homogeneous_Poisson_NN_Legacy.py:183-257).
    rhs (N,1,H,W), dx (N,1).  `taps`, if a dict, receives named intermediates."""
    meta, _ = build_structure(cfg)
    N, _, H, W = rhs.shape
    domain_sizes = ops.concat([dx * (H - 1), dx * (W - 1)], 1)             # :193, :118-120
    x = ops.concat([rhs, position_embeddings(ops, N, H, W)], 1) if meta['use_pos'] else rhs
    dense_inp = ops.concat([dx, domain_sizes], 1)                             # :202
    pre = meta['pre']
    for i in range(len(pre['layers'])):
        x = _pconv(ops, p, 'pre/conv%d' % i, x, pre['padding_mode'], pre['pad_value'], pre['activation'])
        if meta['use_bn']:
            x = _bn(ops, p, 'pre/bn%d' % i, x, bn_training)
    initial = x
    if taps is not None:
        taps['initial'] = initial
    results = [bottleneck_forward(ops, p, b, initial, meta['use_bn'], bn_training) for b in meta['blocks']]
    if taps is not None:
        for b, r in zip(meta['blocks'], results):
            taps[b['name']] = r
    merged = results[0]
    for r in results[1:]:
        merged = merged + r
    merged = merged / float(len(results) * meta['filters'])                  # :222
    nb = ops.same_conv2d(initial, p['non_bottleneck_conv/kernel'], p['non_bottleneck_conv/bias'], 'leaky_relu')
    x = ops.same_conv2d(ops.concat([nb, merged], 1), p['post_merge_conv/kernel'], p['post_merge_conv/bias'], 'leaky_relu')
    x = resnet_forward(ops, p, 'post_merge_resnet', x, 'constant', 0.0, 'leaky_relu', False)
    if taps is not None:
        taps['post_merge'] = x
    d = dense_inp
    for i, act in enumerate(['leaky_relu', 'leaky_relu', 'linear']):
        d = ops.dense(d, p['dx_dense%d/kernel' % i], p['dx_dense%d/bias' % i], act)
    x = x * d[:, :, None, None]                                               # :231
    fin = meta['final']
    for i, (k, cin, f) in enumerate(fin['stages']):
        x = _pconv(ops, p, 'final/stage%d/conv' % i, x, fin['padding_mode'], fin['pad_value'], fin['activation'])
        x = resnet_forward(ops, p, 'final/stage%d/res' % i, x, 'constant', 0.0, fin['activation'], False)
    for j in range(len(fin['tail'])):
        x = ops.same_conv2d(x, p['final/out%d/kernel' % j], p['final/out%d/bias' % j] if fin['use_bias'] else None, 'linear')
    if taps is not None:
        taps['pre_scaling'] = x
    if meta['use_scaling']:
        x = scaling_forward(ops, p, meta['scaling'], x, rhs)
    x = ops.bc_ring(x, 'CONSTANT' if meta['bc_type'] == 'dirichlet' else 'SYMMETRIC')   # :251
    if meta['postsmoother_iterations'] > 0:
        x = ops.jacobi_iterations(x, rhs, ops.concat([dx, dx], 1), meta['postsmoother_iterations'])
    return x
