"""CPU oracle: restatement of Dirichlet_BC_NN_Legacy_2 (models/Dirichlet_BC_NN_Legacy.py:14-187), flip_and_rotate_tensor
(dataset/utils/flip_and_rotate_tensor.py:4-47, for the 4-D channels_first tensors the models pass) and Poisson_CNN_Legacy
(models/Poisson_CNN_Legacy.py:5-71) on the op namespaces of oracle.np_ops / oracle.torch_twin.

TEST INFRASTRUCTURE ONLY - see oracle/np_ops.py header.  PARITY UNPINNED (no TF here).

1-D tensors (N, C, L) of the reference are carried as (N, C, 1, L): a Conv1D kernel (k, Cin, Cout) is the (1, k, Cin, Cout)
kernel of a 2-D convolution whose "advanced" padding on the unit axis is (0, 0)
(utils/apply_advanced_padding_and_call_conv_layer.py:9-11 with k = 1).

Parameter order = attribute-assignment order of the reference constructor: boundary convolutions (conv, [BN], resnet per stage),
domain-info dense layers, final convolutions (conv + resnet per stage, then the plain tail convs).
"""
import copy
from collections import OrderedDict

import numpy as np

from . import hpnn
from .hpnn import get_init_arguments_from_config, glorot_limit, resnet_forward, _bn, _pconv


def _conv_spec(name, kh, kw, cin, cout):
    return [(name + '/kernel', (kh, kw, cin, cout), 'glorot'), (name + '/bias', (cout,), 'zeros')]


def _resnet_spec(name, kh, kw, c, use_bn):
    s = []
    for i in range(3):
        s += _conv_spec('%s/conv%d' % (name, i), kh, kw, c, c)
    if use_bn:
        s += hpnn._bn_spec(name + '/bn0', c) + hpnn._bn_spec(name + '/bn1', c)
    return s


def build_structure(cfg):
    """cfg: the "model" section of experiments/dbcnn.json.  Returns (meta, spec)."""
    cfg = copy.deepcopy(cfg)
    for key, msg in (('boundary_conv_config', 'Provide a config for the boundary convolutions.'), ('spp_config', 'Provide a config for the Spatial Pyramid Pooling.'),
                     ('final_convolutions_config', 'Provide a config for the domain convolutions.'), ('domain_info_mlp_config', 'Provide a config for the domain info MLP.')):
        if cfg.get(key) is None:
            raise ValueError(msg)                                             # reference :26-33
    use_bn = cfg.get('use_batchnorm', False)
    bcc, mlp, fc = cfg['boundary_conv_config'], cfg['domain_info_mlp_config'], cfg['final_convolutions_config']
    assert bcc['filters'][-1] == mlp['units'][-1]                             # :39
    meta = {'use_bn': use_bn, 'nmodes': mlp['units'][-1], 'postsmoother_iterations': cfg.get('postsmoother_iterations', 0)}
    spec = []
    meta['bc'] = {'padding_mode': bcc.get('padding_mode', 'CONSTANT'), 'pad_value': bcc.get('constant_padding_value', 0.0),
                  'activation': bcc.get('activation', 'linear'), 'stages': []}
    cin = 3                                                                    # bc + 2 positional-embedding channels (:139)
    for i, (f, k) in enumerate(zip(bcc['filters'], bcc['kernel_sizes'])):
        spec += _conv_spec('bc/stage%d/conv' % i, 1, k, cin, f)
        if use_bn:
            spec += hpnn._bn_spec('bc/stage%d/bn' % i, f)
        spec += _resnet_spec('bc/stage%d/res' % i, 1, k, f, use_bn)
        meta['bc']['stages'].append((k, cin, f))
        cin = f
    sp = cfg['spp_config']
    meta['spp'] = {'levels': [[1, lv] if isinstance(lv, int) else [1, lv[0]] for lv in sp['levels']], 'kind': sp.get('pooling_type', 'average')}
    nfeat = sum(lv[1] for lv in meta['spp']['levels'])
    din = 1 + 2 + nfeat                                                        # [dx, domain_sizes / max, spp] (:147)
    meta['mlp'] = []
    for i, (u, a) in enumerate(zip(mlp['units'], mlp['activations'])):
        spec += [('mlp/dense%d/kernel' % i, (din, u), 'glorot'), ('mlp/dense%d/bias' % i, (u,), 'zeros')]
        meta['mlp'].append(a)
        din = u
    nreg = fc.get('final_regular_conv_stages', 2)
    meta['final'] = {'padding_mode': fc.get('padding_mode', 'CONSTANT'), 'pad_value': fc.get('constant_padding_value', 0.0),
                     'activation': fc.get('activation', 'linear'), 'stages': [], 'tail': [], 'use_bias': fc.get('use_bias', True)}
    cin = meta['nmodes'] + 2
    nst = len(fc['filters'])
    for i in range(nst - nreg):
        f, k = fc['filters'][i], fc['kernel_sizes'][i]
        spec += _conv_spec('final/stage%d/conv' % i, k, k, cin, f)
        spec += _resnet_spec('final/stage%d/res' % i, k, k, f, False)
        meta['final']['stages'].append((k, cin, f))
        cin = f
    for j, i in enumerate(range(nst - nreg, nst)):
        f, k = fc['filters'][i], fc['kernel_sizes'][i]
        spec += _conv_spec('final/out%d' % j, k, k, cin, f)
        meta['final']['tail'].append((k, cin, f))
        cin = f
    return meta, spec


def init_params(cfg, seed=0, gain=1.0, randomize_all=False):
    _, spec = build_structure(cfg)
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shape, init in spec:
        if init == 'glorot':
            lim = glorot_limit(shape) * gain
            v = rng.uniform(-lim, lim, size=shape)
        elif init == 'zeros':
            v = rng.uniform(-0.1, 0.1, size=shape) if randomize_all else np.zeros(shape)
        else:
            v = rng.uniform(0.6, 1.4, size=shape) if randomize_all else np.ones(shape)
        p[name] = v.astype(np.float32).astype(np.float64)
    return p


def sinh_basis(nmodes, X):
    """build_series_x_dir_components (:106-111): sinh(m pi (xbar - 1)), each mode scaled to max magnitude 1."""
    xbar = np.linspace(0.0, 1.0, X)
    m = np.arange(1, nmodes + 1, dtype=np.float64)
    v = np.sinh(np.einsum('m,x->mx', m, np.pi * (xbar - 1.0)))
    return v / np.abs(v).max(axis=1, keepdims=True)


def position_embeddings(N, X, L):
    """generate_position_embeddings (:113-124): channel 0 varies along x (axis 2), channel 1 along y (axis 3)."""
    ex = np.cos(np.pi * np.linspace(0.0, 1.0, X))[:, None] * np.ones((1, L))
    ey = np.ones((X, 1)) * np.cos(np.pi * np.linspace(0.0, 1.0, L))[None, :]
    return np.broadcast_to(np.stack([ex, ey], 0)[None], (N, 2, X, L)).copy()


def forward(ops, cfg, p, bc, dx, X, bn_training=False, taps=None):
    """Dirichlet_BC_NN_Legacy_2.call (:126-170).  bc (N,1,L), dx (N,1), X = x_output_resolution.  Returns (N,1,X,L)."""
    meta, _ = build_structure(cfg)
    bc = ops.asarray(bc)
    dx = ops.asarray(dx)
    N, _, L = bc.shape
    M = meta['nmodes']
    domain_sizes = ops.concat([dx * (X - 1), dx * (L - 1)], 1)                # compute_domain_sizes(concat([dx,dx]), shape) (:131)
    pos = position_embeddings(N, X, L)
    o = ops.concat([bc[:, :, None, :], ops.asarray(pos[:, :, 0:1, :])], 1)     # (N,3,1,L) (:136-139)
    b = meta['bc']
    for i in range(len(b['stages'])):
        o = _pconv(ops, p, 'bc/stage%d/conv' % i, o, b['padding_mode'], b['pad_value'], b['activation'])
        if meta['use_bn']:
            o = _bn(ops, p, 'bc/stage%d/bn' % i, o, bn_training)
        o = resnet_forward(ops, p, 'bc/stage%d/res' % i, o, b['padding_mode'], b['pad_value'], b['activation'], meta['use_bn'], bn_training)
    bc_conv = o                                                                # (N,M,1,L)
    feats = ops.spatial_pyramid_pool(bc_conv, meta['spp']['levels'], 'avg' if meta['spp']['kind'].lower() in ('average', 'avg') else 'max')
    dsz_max = domain_sizes.max(axis=1, keepdims=True) if isinstance(domain_sizes, np.ndarray) else domain_sizes.max(dim=1, keepdim=True).values
    d = ops.concat([dx, domain_sizes / dsz_max, feats], 1)                    # (:147)
    for i, a in enumerate(meta['mlp']):
        d = ops.dense(d, p['mlp/dense%d/kernel' % i], p['mlp/dense%d/bias' % i], a)
    if taps is not None:
        taps['bc_conv'], taps['mlp'] = bc_conv, d
    sh = ops.asarray(sinh_basis(M, X))
    out = ops.einsum('bmy,mx,bm->bmxy', bc_conv[:, :, 0, :], sh, d)           # (:156)
    out = ops.concat([out, ops.asarray(pos)], 1)                              # (:159)
    fin = meta['final']
    for i in range(len(fin['stages'])):
        out = _pconv(ops, p, 'final/stage%d/conv' % i, out, fin['padding_mode'], fin['pad_value'], fin['activation'])
        out = resnet_forward(ops, p, 'final/stage%d/res' % i, out, 'constant', 0.0, fin['activation'], False)
    for j in range(len(fin['tail'])):
        out = ops.same_conv2d(out, p['final/out%d/kernel' % j], p['final/out%d/bias' % j] if fin['use_bias'] else None, 'tanh')   # (:98)
    if taps is not None:
        taps['pre_norm'] = out
    out = ops.set_max_magnitude_in_batch(out, 1.0)                             # (:163)
    out = ops.concat([bc[:, :, None, :], out[:, :, 1:, :]], 2)                # (:165-166)
    if meta['postsmoother_iterations'] > 0:
        out = ops.jacobi_iterations(out, ops.zeros_like(out), ops.concat([dx, dx], 1), meta['postsmoother_iterations'])
    return out


# --------------------------------------------------------------------------- flip_and_rotate_tensor
def flip_and_rotate(ops, x, rotation_count=0, flip_axes=()):
    """dataset/utils/flip_and_rotate_tensor.py:4-47 for a 4-D channels_first tensor with the default rotation_axis = 4 (the unit
    axis the function appends): rotation_count != 0 transposes axes 2 <-> 3 iff rotation_count is odd (the two rotatable axes are
    cycled by rotation_count % 2, :31-33), then axes are reversed: the explicit flip_axes plus, per the lookup table (:36-38)
    [[0,0],[1,0],[1,1],[0,1]][(|r| % 4) * sign(r)] added to axes (2, 3); counts are taken modulo 2 (:41-42)."""
    flips = {2: 0, 3: 0}
    for a in flip_axes:
        flips[a] += 1
    out = x
    if rotation_count != 0:
        if rotation_count % 2 == 1:
            out = ops.transpose(out, (0, 1, 3, 2))
        table = [[0, 0], [1, 0], [1, 1], [0, 1]]
        idx = (abs(rotation_count) % 4) * (1 if rotation_count > 0 else -1)
        req = table[idx]                                                       # negative idx wraps like tf.gather? not used by the models
        flips[2] += req[0]
        flips[3] += req[1]
    axes = [a for a in (2, 3) if flips[a] % 2 == 1]
    return ops.flip(out, axes)


def pcnn_forward(ops, hp_cfg, hp_params, db_cfg, db_params, rhs, left, top, right, bottom, dx):
    """Poisson_CNN_Legacy.call (models/Poisson_CNN_Legacy.py:16-54) without the optional Jacobi layer (its constructor line :11
    references an unimported module name, so jacobi_iterations > 0 raises NameError in the reference)."""
    rhs, left, top, right, bottom, dx = [ops.asarray(v) for v in (rhs, left, top, right, bottom, dx)]
    N, _, H, W = rhs.shape
    rhs_n, rhs_f = ops.set_max_magnitude_in_batch(rhs, 1.0, True)
    sides = {}
    for name, v in (('left', left), ('top', top), ('right', right), ('bottom', bottom)):
        sides[name] = ops.set_max_magnitude_in_batch(v, 1.0, True)
    h = hpnn.forward(ops, hp_cfg, hp_params, rhs_n, dx)
    dom = ops.concat([dx * (H - 1), dx * (W - 1)], 1)
    dmax = dom.max(axis=1) if isinstance(dom, np.ndarray) else dom.max(dim=1).values
    h = ops.einsum('bcxy,b->bcxy', h, dmax ** 2 / rhs_f)                       # (:30)
    l_ = ops.einsum('bcxy,b->bcxy', forward(ops, db_cfg, db_params, sides['left'][0], dx, H), 1.0 / sides['left'][1])
    t_ = ops.einsum('bcxy,b->bcxy', forward(ops, db_cfg, db_params, sides['top'][0], dx, W), 1.0 / sides['top'][1])
    t_ = flip_and_rotate(ops, t_, rotation_count=3, flip_axes=())
    r_ = ops.einsum('bcxy,b->bcxy', forward(ops, db_cfg, db_params, sides['right'][0], dx, H), 1.0 / sides['right'][1])
    r_ = flip_and_rotate(ops, r_, rotation_count=0, flip_axes=(2,))
    b_ = ops.einsum('bcxy,b->bcxy', forward(ops, db_cfg, db_params, sides['bottom'][0], dx, W), 1.0 / sides['bottom'][1])
    b_ = flip_and_rotate(ops, b_, rotation_count=1, flip_axes=(2,))
    return l_ + r_ + t_ + b_ + h
