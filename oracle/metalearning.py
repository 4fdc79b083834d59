"""CPU oracle: restatement of the metalearning family on the fp64 autograd twin (oracle/torch_twin.py).

TEST INFRASTRUCTURE ONLY - see oracle/np_ops.py header.  PARITY UNPINNED (no TF here).

  hyper                 the dense chain (+ tf.keras.layers.LayerNormalization, epsilon 1e-3) that emits kernel and bias
                        (layers/metalearning_conv.py:124-129,152-157)
  mconv                 layers/metalearning_conv.py:148-169: per-sample tf.pad + conv2d(strides) + bias + activation
  mdeconv               layers/metalearning_deconvupscale.py:104-137: per-sample conv2d_transpose(SAME) + bias
  mresnet               blocks/metalearning_resnet.py:27-37
  mbottleneck           blocks/metalearning_bottleneck_block.py:96-118 (deconv) and :173-191 (multilinear)
Parameters: dict name -> tensor with the names poisson_cnn_amd/metalearning.py registers.
"""
import torch

from . import np_ops, torch_twin as T


def hyper(p, name, dense_input, acts, use_layernorm=False):
    """The Dense layers carry a bias only when the layer's `use_bias` is set (it is forwarded into every Dense layer through
    `_other_dense_layer_args`, layers/metalearning_conv.py:113,128): a missing '<name>/dense<i>/bias' entry means no bias."""
    kb = dense_input
    for i, a in enumerate(acts):
        kb = T.dense(kb, p['%s/dense%d/kernel' % (name, i)], p.get('%s/dense%d/bias' % (name, i)), a)
    if use_layernorm:
        mu = kb.mean(dim=-1, keepdim=True)
        var = ((kb - mu) ** 2).mean(dim=-1, keepdim=True)
        kb = (kb - mu) / torch.sqrt(var + 1e-3) * p[name + '/layernorm/gamma'] + p[name + '/layernorm/beta']
    return kb


def mconv(p, name, x, dense_input, k, cin, cout, acts, *, same=True, mode='CONSTANT', value=0.0, act='linear', stride=1, use_bias=True, use_layernorm=False,
          kh=None):
    """kh = 1: the dimensions = 1 layer (tf.nn.conv1d, layers/metalearning_conv.py:115-116) on (N, C, 1, L) tensors - a (1, k) kernel whose padding on
    the unit axis is [0, 0] (:103-107 with ks = 1)."""
    kb = hyper(p, name, dense_input, acts, use_layernorm)
    kh = k if kh is None else kh
    nk = kh * k * cin * cout
    outs = []
    for n in range(x.shape[0]):
        kern = kb[n, :nk].reshape(kh, k, cin, cout)
        bias = kb[n, nk:] if use_bias else None
        if same:
            outs.append(T.padded_conv2d(x[n:n + 1], kern, bias, mode, value, act, stride=stride))
        else:
            outs.append(T.activation(T.conv2d_valid(x[n:n + 1], kern, bias, stride=stride), act))
    return torch.cat(outs, 0)


def mdeconv(p, name, x, dense_input, k, cin, cout, acts, out_hw, use_bias=True):
    kb = hyper(p, name, dense_input, acts)
    nk = k * k * cout * cin
    outs = [T.conv2d_transpose_same(x[n:n + 1], kb[n, :nk].reshape(k, k, cout, cin), kb[n, nk:] if use_bias else None, out_hw, k, 'linear') for n in range(x.shape[0])]
    return torch.cat(outs, 0)


def _bn(p, name, x):
    return T.batchnorm_inference(x, p[name + '/gamma'], p[name + '/beta'], p[name + '/moving_mean'], p[name + '/moving_variance'])


def mresnet(p, name, x, d, k, c, acts, use_bn, **ckw):
    o = mconv(p, name + '/conv0', x, d, k, c, c, acts, **ckw)
    if use_bn:
        o = _bn(p, name + '/bn0', o)
    o = mconv(p, name + '/conv1', o, d, k, c, c, acts, **ckw)
    if use_bn:
        o = _bn(p, name + '/bn1', o)
    return mconv(p, name + '/conv2', x + o, d, k, c, c, acts, **ckw)


def mbottleneck(p, name, x, d, *, kind, f, up, filters, k, n_convs, acts, mode, value, act, method, pool, use_resnet, use_bn, kdown=None, kdeconv=None, dacts=None,
                use_bias=True, deconv_use_bias=True):
    H, W = x.shape[2], x.shape[3]
    cin = x.shape[1]
    ckw = dict(same=True, mode=mode, value=value, act=act, use_bias=use_bias)
    n_layers = 0
    if method == 'conv':
        o = mconv(p, name + '/downsample', x, d, kdown if kdown is not None else k, cin, filters, acts, stride=f, **ckw)
    else:
        o = T.pool2d_same(x, f, pool)
        o = mconv(p, name + '/conv0', o, d, k, cin, filters, acts, **ckw)
        n_layers = 1
    i = 0
    while n_layers < n_convs:
        if use_resnet:
            o = mresnet(p, '%s/res%d' % (name, i), o, d, k, filters, acts, use_bn, **ckw)
            n_layers += 1
        else:
            o = mconv(p, '%s/stage%d' % (name, i), o, d, k, filters, filters, acts, **ckw)
            n_layers += 1
            if use_bn and kind == 'multilinear':
                o = _bn(p, '%s/stage_bn%d' % (name, i), o)
                n_layers += 1
        i += 1
    out_hw = (int((H / f) * up), int((W / f) * up))
    if kind == 'deconv':
        o = mdeconv(p, name + '/deconv', o, d, kdeconv, filters, filters, dacts, out_hw, use_bias=deconv_use_bias)
        return _bn(p, name + '/bn', o) if use_bn else o
    return T.resize2d(o, out_hw, 'bilinear')
