"""The metalearning family: layers whose per-sample filters and biases are emitted by a small dense
"hyper-network" from `dense_input`, and the blocks composed of them.

    metalearning_conv                                  layers/metalearning_conv.py:52-169          call([x, dense_input])
    metalearning_deconvupscale                         layers/metalearning_deconvupscale.py:40-137 call([x, dense_input, output_shape])
    metalearning_resnet                                blocks/metalearning_resnet.py:6-37          call([x, dense_input])
    metalearning_bottleneck_block_deconvupsample       blocks/metalearning_bottleneck_block.py:8-118    call([x, dense_input])
    metalearning_bottleneck_block_multilinearupsample  blocks/metalearning_bottleneck_block.py:120-191  call([x, dense_input, domain_sizes])

The reference applies the per-sample kernels with tf.map_fn; here the hyper-network runs on the libpcnn dense / layer-norm kernels and every
sample's convolution, transposed convolution, filter gradient and data gradient is one launch of the fused kernels on that sample's
(1,H,W,C) slice with its own filter - a grouped implicit GEMM by launch.  Constructor kwargs are the reference's (activations may be given as
"tf.nn.*" strings); the input-channel count and the dense-input width are taken from the first call (lazy build) unless passed explicitly.
`layer(inputs, training=True)` keeps what `layer.backward(dy)` needs; backward returns (dx, d_dense_input) and leaves the hyper-network's
parameter gradients in `layer.store.g`.  The tensor API is channels_first like the reference; `forward` / `backward` take NHWC tensors.

Where the reference's own constructor paths cannot run (blocks/metalearning_bottleneck_block.py:66-69 hands `use_batchnorm` to
metalearning_conv, which passes it on to tf.keras.layers.Layer.__init__ and raises), the evident intent is implemented: a plain
metalearning_conv stage.
"""
import numpy as np
import torch

from . import layers as L
from . import ops
from .utils import canonical_activation


def _acts(dense_activations, n):
    if isinstance(dense_activations, (list, tuple)):
        if len(dense_activations) != n:
            raise ValueError('dense_activations needs %d entries (one per dense layer)' % n)
        return [a if a is not None else 'linear' for a in dense_activations]
    return [dense_activations if dense_activations is not None else 'linear'] * n


def _scalar(v, what):
    if np.isscalar(v):
        return int(v)
    if len(set(int(a) for a in v)) != 1:
        raise NotImplementedError('anisotropic %s' % what)
    return int(v[0])


class _Hyper:
    """Shared machinery: own-or-shared ParamStore, lazy build, the dense chain (+ optional LayerNormalization) that emits kernel and bias."""

    def _init_common(self, data_format, store, ctx, name, device, seed):
        if data_format != 'channels_first':
            raise NotImplementedError('channels_first only (every config of the reference)')
        self.name = name
        self.own_store = store is None
        self.store = store if store is not None else L.ParamStore()
        self.ctx = ctx if ctx is not None else L.Context()
        self.device = device
        self.seed = seed
        self.built = False

    def _build_dense(self, din, n_out, units, dense_activations, use_layernorm):
        units = list(units) + [n_out]
        acts = _acts(dense_activations, len(units))
        self.dense = []
        for i, (u, a) in enumerate(zip(units, acts)):
            self.dense.append(L.Dense(self.store, '%s/dense%d' % (self.name, i), int(din), int(u), a, use_bias=self.use_bias))
            din = u
        self.use_layernorm = bool(use_layernorm)
        if self.use_layernorm:
            self.store.add(self.name + '/layernorm/gamma', (n_out,), 'ones')
            self.store.add(self.name + '/layernorm/beta', (n_out,), 'zeros')
        self.built = True
        if self.own_store:
            self.finalize()

    def finalize(self):
        """Allocates and initialises an own parameter bucket (a composing block finalises the shared one itself)."""
        if not self.store.finalized:
            dev = torch.device(self.device) if self.device is not None else torch.device('cuda', torch.cuda.current_device())
            self.store.finalize(dev)
            self.store.initialize(self.seed)

    def _emit(self, dense_input, training):
        kb = dense_input
        for d in self.dense:
            kb = d.forward(kb, training)
        self._ln = None
        if self.use_layernorm:
            pre = kb
            kb, st = ops.layernorm_fwd(pre, self.store.w[self.name + '/layernorm/gamma'], self.store.w[self.name + '/layernorm/beta'])
            self._ln = (pre, st) if training else None
        return kb

    def _emit_backward(self, dkb):
        d = dkb
        if self.use_layernorm:
            pre, st = self._ln
            g = self.store.g
            d = ops.layernorm_bwd(pre, self.store.w[self.name + '/layernorm/gamma'], st, d, g[self.name + '/layernorm/gamma'], g[self.name + '/layernorm/beta'])
        for lyr in reversed(self.dense):
            d = lyr.backward(d, need_dx=True)
        return d

    # Keras-style access
    @property
    def trainable_variables(self):
        return [self.store.w[n] for n in self.store.trainable_names()]

    @property
    def gradients(self):
        return {n: self.store.g[n] for n in self.store.trainable_names()}


class metalearning_conv(_Hyper):
    def __init__(self, filters, kernel_size, previous_layer_filters=None, dense_input_features=None, strides=None, padding='valid', padding_mode='constant',
                 constant_padding_value=0.0, data_format='channels_first', dilation_rate=None, conv_activation='linear', use_bias=True,
                 dense_activations='linear', pre_output_dense_units=(8, 16), use_layernorm=False, dimensions=None, store=None, ctx=None,
                 name='metalearning_conv', device=None, seed=0, **unused_initializers):
        self._init_common(data_format, store, ctx, name, device, seed)
        if dilation_rate not in (None, 1):
            raise NotImplementedError('dilation other than 1 is not used by any reference config')
        if dimensions not in (None, 1, 2):
            raise NotImplementedError('dimensions = 1 or 2')
        # dimensions = 1 (tf.nn.conv1d, layers/metalearning_conv.py:115-116; the boundary convolutions of models/Dirichlet_BC_NN_Metalearning.py:43-55):
        # the same kernels with one filter row - tensors (N, L, C) are handled as (N, 1, L, C)
        self.ndims = 1 if (dimensions == 1 or (dimensions is None and not isinstance(kernel_size, int) and len(kernel_size) == 1)) else 2
        self.k = _scalar(kernel_size, 'kernel_size')
        self.stride = 1 if strides is None else _scalar(strides, 'strides')
        self.cout = int(filters)
        self.same = padding.lower() == 'same'
        if self.stride > 1 and not self.same:
            raise NotImplementedError('strided metalearning_conv is implemented for padding="same" (its only strided use, blocks/metalearning_bottleneck_block.py:60-62)')
        self.mode = padding_mode.upper()
        if self.mode not in ops.PAD_MODES:
            raise ValueError('unknown padding mode ' + padding_mode)
        self.pad_value = float(constant_padding_value)
        self.act = canonical_activation(conv_activation)
        self.use_bias = use_bias
        self._dense_args = (list(pre_output_dense_units), dense_activations, use_layernorm)
        self.cin = None
        if previous_layer_filters is not None and dense_input_features is not None:
            self.build(previous_layer_filters, dense_input_features)

    def build(self, cin, dense_features):
        self.cin = int(cin)
        units, acts, ln = self._dense_args
        self.kh = 1 if self.ndims == 1 else self.k
        self._build_dense(dense_features, self.kh * self.k * self.cin * self.cout + (self.cout if self.use_bias else 0), units, acts, ln)

    def _pads(self):
        # 'same': [ks//2, ks//2 if odd else ks//2-1] (metalearning_conv.py:103-107); 'valid': none
        return (self.k // 2, self.k // 2 - (1 - self.k % 2)) if self.same else (0, 0)

    def forward(self, x, dense_input, training=True):
        """x (N,H,W,Cin) NHWC - (N,L,Cin) or (N,1,L,Cin) for a 1-D layer, the result has the same rank -, dense_input (N,F)."""
        self._squeeze = self.ndims == 1 and x.dim() == 3
        if self._squeeze:
            x = x.unsqueeze(1)
        if self.ndims == 1 and x.shape[1] != 1:
            raise ValueError('a 1-D metalearning_conv takes (N,L,C) or (N,1,L,C) tensors')
        if not self.built:
            self.build(x.shape[3], dense_input.shape[1])
        N, H, W, _ = x.shape
        kb = self._emit(dense_input, training)
        kh, kw = self.kh, self.k
        nk = kh * kw * self.cin * self.cout
        pt, pb = self._pads()
        pty, pby = (0, 0) if self.ndims == 1 else (pt, pb)
        Ho, Wo = H + pty + pby - kh + 1, W + pt + pb - kw + 1
        # ONE launch for the batch: sample n takes its filter and bias from row n of the hyper-network's output (the reference serialises the
        # samples with tf.map_fn, layers/metalearning_conv.py:18,30)
        y = ops.grouped_conv2d_fwd(x, kb, (kh, kw, self.cin, self.cout), kb[:, nk:] if self.use_bias else None, pad_top=pty, pad_left=pt, out_hw=(Ho, Wo),
                                   pad_mode=self.mode if self.same else 'CONSTANT', pad_value=self.pad_value, act=self.act)
        self.saved = (x, kb, y) if training else None
        if self.stride > 1:
            y = ops.subsample(y, self.stride)
        return y.squeeze(1) if self._squeeze else y

    def backward(self, dy, need_dx=True):
        """Returns (dx, d_dense_input); parameter gradients of the hyper-network go to store.g."""
        x, kb, y = self.saved
        self.saved = None
        if self._squeeze:
            dy = dy.unsqueeze(1)
        if self.stride > 1:
            dy = ops.subsample_bwd(dy, (y.shape[1], y.shape[2]), self.stride)
        N, H, W, _ = x.shape
        kh, kw = self.kh, self.k
        nk = kh * kw * self.cin * self.cout
        pt, pb = self._pads()
        pty, pby = (0, 0) if self.ndims == 1 else (pt, pb)
        dkb = ops.zeros(tuple(kb.shape), x.device)
        dx = ops.empty(tuple(x.shape), x.device) if need_dx else None
        mode = self.mode if self.same else 'CONSTANT'
        wshape = (kh, kw, self.cin, self.cout)
        if self.act == 'linear':
            dz = dy if dy.is_contiguous() else dy.contiguous()
        else:
            dz = ops.empty(tuple(dy.shape), x.device)
            ops.epilogue_bwd(dy, y, act=self.act, dz=dz, ws=self.ctx.ws)
        if self.use_bias:
            ops.grouped_bias_grad(dz, dkb[:, nk:])
        ops.grouped_conv2d_wgrad(x, dz, wshape, dkb, pad_top=pty, pad_left=pt, pad_mode=mode, pad_value=self.pad_value, ws=self.ctx.ws)
        if need_dx:          # the forward filters read flipped and transposed inside the kernel: no per-sample flip launches
            if mode == 'CONSTANT':
                ops.grouped_conv2d_fwd(dz, kb, wshape, None, pad_top=kh - 1 - pty, pad_left=kw - 1 - pt, out_hw=(H, W), flip_transpose=True, out=dx)
            else:
                gp = ops.grouped_conv2d_fwd(dz, kb, wshape, None, pad_top=kh - 1, pad_left=kw - 1, out_hw=(H + pty + pby, W + pt + pb), flip_transpose=True)
                ops.pad_fold_bwd(gp, (H, W), ((pty, pby), (pt, pb)), mode, out=dx)
        if need_dx and self._squeeze:
            dx = dx.squeeze(1)
        return dx, self._emit_backward(dkb)

    def __call__(self, inputs, training=False):
        """Reference call convention: [conv_input (N,C,H,W), dense_input (N,F)] -> (N,filters,H',W')."""
        x, dense_input = inputs
        if self.ndims == 1:                     # (N, C, L) -> (N, L, C)
            return self.forward(x.permute(0, 2, 1).contiguous(), dense_input.contiguous(), training=training).permute(0, 2, 1)
        x = x.permute(0, 2, 3, 1).contiguous()
        return self.forward(x, dense_input.contiguous(), training=training).permute(0, 3, 1, 2)


class metalearning_deconvupscale(_Hyper):
    """Per-sample tf.nn.conv2d_transpose(x_n, K_n, output_shape, strides=upsample_ratio, 'SAME') + bias_n, K_n (k, k, filters, Cin) and bias_n
    emitted by the hyper-network (layers/metalearning_deconvupscale.py:40-137).  kernel_size == upsample_ratio, linear conv_activation."""

    def __init__(self, upsample_ratio, filters, kernel_size, data_format='channels_first', conv_activation='linear', use_bias=True, dimensions=None,
                 dense_activations='linear', pre_output_dense_units=(8, 16), previous_layer_filters=None, dense_input_features=None, store=None, ctx=None,
                 name='metalearning_deconvupscale', device=None, seed=0, **unused_initializers):
        self._init_common(data_format, store, ctx, name, device, seed)
        self.up, self.k = _scalar(upsample_ratio, 'upsample_ratio'), _scalar(kernel_size, 'kernel_size')
        if self.up != self.k:
            raise NotImplementedError('metalearning_deconvupscale is implemented for kernel_size == upsample_ratio')
        if canonical_activation(conv_activation) != 'linear':
            raise NotImplementedError('conv_activation other than linear is not used by any reference config')
        if dimensions not in (None, 2):
            raise NotImplementedError('dimensions = 2 only')
        self.cout, self.use_bias = int(filters), use_bias
        self._dense_args = (list(pre_output_dense_units), dense_activations)
        if previous_layer_filters is not None and dense_input_features is not None:
            self.build(previous_layer_filters, dense_input_features)

    def build(self, cin, dense_features):
        self.cin = int(cin)
        units, acts = self._dense_args
        self._build_dense(dense_features, self.k * self.k * self.cout * self.cin + (self.cout if self.use_bias else 0), units, acts, False)

    def forward(self, x, dense_input, out_hw, training=True):
        if not self.built:
            self.build(x.shape[3], dense_input.shape[1])
        N = x.shape[0]
        kb = self._emit(dense_input, training)
        nk = self.k * self.k * self.cout * self.cin
        y = ops.grouped_deconv_fwd(x, kb, (self.k, self.k, self.cout, self.cin), kb[:, nk:] if self.use_bias else None, out_hw, self.up)
        self.saved = (x, kb) if training else None
        return y

    def backward(self, dy, need_dx=True):
        x, kb = self.saved
        self.saved = None
        N = x.shape[0]
        nk = self.k * self.k * self.cout * self.cin
        dkb = ops.zeros(tuple(kb.shape), x.device)
        dx = ops.empty(tuple(x.shape), x.device) if need_dx else None
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        ops.grouped_deconv_bwd_filter(x, dyc, self.up, dkb, dkb[:, nk:] if self.use_bias else None)
        if need_dx:
            ops.grouped_deconv_bwd_data(dyc, kb, (self.k, self.k, self.cout, self.cin), (x.shape[1], x.shape[2]), self.up, out=dx)
        return dx, self._emit_backward(dkb)

    def __call__(self, inputs, training=False):
        x, dense_input, output_shape = inputs
        shp = [int(v) for v in (output_shape.tolist() if hasattr(output_shape, 'tolist') else output_shape)]
        x = x.permute(0, 2, 3, 1).contiguous()
        return self.forward(x, dense_input.contiguous(), (shp[-2], shp[-1]), training=training).permute(0, 3, 1, 2)


class _BatchNorm:
    """Inference-mode tf.keras.layers.BatchNormalization(axis=1) as its own layer (between metalearning layers nothing can be fused into a
    convolution epilogue): y = a * scale + shift with the folded statistics of the shared ParamStore; backward feeds the store's S1 / S2 sums."""

    def __init__(self, store, ctx, name, c):
        self.store, self.ctx, self.c = store, ctx, c
        self.off = store.add_bn(name, c)

    def forward(self, a, training=True, out=None):
        s = self.store
        self.saved = a if training else None
        return ops.channel_affine(a, s.bn_scale[self.off:self.off + self.c], s.bn_shift[self.off:self.off + self.c], out=out)

    def backward(self, dy):
        s = self.store
        dz = ops.empty(tuple(dy.shape), dy.device)
        ops.epilogue_bwd(dy, self.saved, act='linear', bn_scale=s.bn_scale[self.off:self.off + self.c], dz=dz, s_dy_a=s.bn_s1[self.off:self.off + self.c],
                         s_dy=s.bn_s2[self.off:self.off + self.c], ws=self.ctx.ws)
        self.saved = None
        return dz


class _Block(_Hyper):
    """A composition sharing one parameter bucket; d(dense_input) is the sum over the member layers."""

    def _pre(self):
        if self.store.nbn:
            self.store.refresh_bn()

    def _post(self):
        self.ctx.join()
        if self.store.nbn:
            self.store.finish_bn_grads()


class metalearning_resnet(_Block):
    """blocks/metalearning_resnet.py:6-37: o = conv0([x, d]); [BN0]; o = conv1([o, d]); [BN1]; o = x + o; o = conv2([o, d]), all 'same'."""

    def __init__(self, filters, kernel_size, use_batchnorm=False, batchnorm_trainable=True, store=None, ctx=None, name='metalearning_resnet', device=None, seed=0,
                 previous_layer_filters=None, dense_input_features=None, **other_metalearning_conv_args):
        dfmt = other_metalearning_conv_args.pop('data_format', 'channels_first')
        self._init_common(dfmt, store, ctx, name, device, seed)
        other_metalearning_conv_args.pop('padding', None)
        self.filters, self.use_bn = int(filters), use_batchnorm
        self.convs = [metalearning_conv(filters, kernel_size, padding='same', store=self.store, ctx=self.ctx, name='%s/conv%d' % (name, i), **other_metalearning_conv_args)
                      for i in range(3)]
        self.bns = None
        self._shape_args = (previous_layer_filters, dense_input_features)
        if previous_layer_filters is not None and dense_input_features is not None:
            self.build(previous_layer_filters, dense_input_features)

    def build(self, cin, dense_features):
        if int(cin) != self.filters:
            raise ValueError('metalearning_resnet needs as many input channels as filters (%d vs %d)' % (cin, self.filters))
        for c in self.convs:
            c.build(cin, dense_features)
        if self.use_bn:
            self.bns = [_BatchNorm(self.store, self.ctx, '%s/bn%d' % (self.name, i), self.filters) for i in range(2)]
        self.built = True
        if self.own_store:
            self.finalize()

    def forward(self, x, dense_input, training=True):
        if not self.built:
            self.build(x.shape[3], dense_input.shape[1])
        if self.own_store:
            self._pre()
        o = self.convs[0].forward(x, dense_input, training)
        if self.bns:
            o = self.bns[0].forward(o, training)
        o = self.convs[1].forward(o, dense_input, training)
        if self.bns:
            o = self.bns[1].forward(o, training)
        s = ops.empty(tuple(x.shape), x.device)
        ops.axpby(1.0, x, 0.0, s)
        ops.axpby(1.0, o, 1.0, s)
        return self.convs[2].forward(s, dense_input, training)

    def backward(self, dy, need_dx=True):
        ds, dd = self.convs[2].backward(dy)
        d = ds
        if self.bns:
            d = self.bns[1].backward(d)
        d, dd1 = self.convs[1].backward(d)
        if self.bns:
            d = self.bns[0].backward(d)
        dx, dd0 = self.convs[0].backward(d)
        ops.axpby(1.0, ds, 1.0, dx)                                  # the skip connection
        ops.axpby_flat(1.0, dd1, 1.0, dd)
        ops.axpby_flat(1.0, dd0, 1.0, dd)
        if self.own_store:
            self._post()
        return dx, dd

    def __call__(self, inputs, training=False):
        x, dense_input = inputs
        if self.convs[0].ndims == 1:            # dimensions = 1: (N, C, L) -> (N, 1, L, C) and back
            return self.forward(x.permute(0, 2, 1).unsqueeze(1).contiguous(), dense_input.contiguous(), training=training).squeeze(1).permute(0, 2, 1)
        return self.forward(x.permute(0, 2, 3, 1).contiguous(), dense_input.contiguous(), training=training).permute(0, 3, 1, 2)


class _metalearning_bottleneck(_Block):
    def _setup(self, ndims, downsampling_factor, filters, conv_kernel_size, n_convs, conv_padding_mode, conv_constant_padding_value, conv_conv_activation,
               conv_dense_activation, conv_pre_output_dense_units, conv_use_bias, use_resnet, upsampling_factor, data_format, downsampling_method,
               conv_downsampling_kernel_size, pool_downsampling_method, use_batchnorm, bn_counts_as_stage, store, ctx, name, device, seed):
        self._init_common(data_format, store, ctx, name, device, seed)
        if ndims != 2:
            raise NotImplementedError('ndims = 2 only (the hot path of BASELINE.json)')
        self.f = int(downsampling_factor)
        self.up = int(upsampling_factor) if upsampling_factor is not None else self.f
        self.filters = int(filters)
        self.method, self.pool = downsampling_method.lower(), pool_downsampling_method.lower()
        if self.method not in ('conv', 'pool'):
            raise ValueError('Downsampling method can only be conv or pool')
        cargs = dict(padding_mode=conv_padding_mode, constant_padding_value=conv_constant_padding_value, conv_activation=conv_conv_activation,
                     dense_activations=conv_dense_activation, pre_output_dense_units=conv_pre_output_dense_units, use_bias=conv_use_bias,
                     store=self.store, ctx=self.ctx)
        self.down = None
        self.stages = []          # ('conv' | 'resnet' | 'bn', object)
        n_layers = 0
        if self.method == 'conv':
            kd = conv_downsampling_kernel_size if conv_downsampling_kernel_size is not None else conv_kernel_size
            self.down = metalearning_conv(filters, kd, strides=self.f, padding='same', name=name + '/downsample', **cargs)
        else:
            self.stages.append(('conv', metalearning_conv(filters, conv_kernel_size, padding='same', name=name + '/conv0', **cargs)))
            n_layers = 1
        i = 0
        self._bn_names = []
        while n_layers < n_convs:
            if use_resnet:
                self.stages.append(('resnet', metalearning_resnet(filters, conv_kernel_size, use_batchnorm=use_batchnorm, name='%s/res%d' % (name, i), **cargs)))
                n_layers += 1
            else:
                self.stages.append(('conv', metalearning_conv(filters, conv_kernel_size, padding='same', name='%s/stage%d' % (name, i), **cargs)))
                n_layers += 1
                if use_batchnorm and bn_counts_as_stage:
                    self.stages.append(('bn', '%s/stage_bn%d' % (name, i)))
                    n_layers += 1
            i += 1

    def _build_stages(self, cin, dense_features):
        c = cin
        if self.down is not None:
            self.down.build(c, dense_features)
            c = self.filters
        built = []
        for kind, obj in self.stages:
            if kind == 'bn':
                obj = _BatchNorm(self.store, self.ctx, obj, self.filters)
            else:
                obj.build(c, dense_features)
                c = self.filters
            built.append((kind, obj))
        self.stages = built

    def _down_and_stages(self, x, dense_input, training):
        self.x = x if training else None
        o = self.down.forward(x, dense_input, training) if self.down is not None else ops.pool2d_fwd(x, self.f, self.pool)
        for kind, obj in self.stages:
            o = obj.forward(o, training) if kind == 'bn' else obj.forward(o, dense_input, training)
        return o

    def _backward_stages_and_down(self, d, dd):
        for kind, obj in reversed(self.stages):
            if kind == 'bn':
                d = obj.backward(d)
            else:
                d, ddi = obj.backward(d)
                ops.axpby_flat(1.0, ddi, 1.0, dd)
        if self.down is not None:
            dx, ddi = self.down.backward(d)
            ops.axpby_flat(1.0, ddi, 1.0, dd)
        else:
            dx = ops.pool2d_bwd(self.x, d, self.f, self.pool)
        self.x = None
        return dx

    def out_hw(self, H, W):
        return int((H / self.f) * self.up), int((W / self.f) * self.up)


class metalearning_bottleneck_block_deconvupsample(_metalearning_bottleneck):
    """blocks/metalearning_bottleneck_block.py:8-118; call([x, dense_input]); use_batchnorm adds ONE BatchNormalization after the up-sampling
    (:91-94; resnet stages carry their own)."""

    def __init__(self, ndims, downsampling_factor, filters, conv_kernel_size, deconv_kernel_size, n_convs=1, conv_padding_mode='constant',
                 conv_constant_padding_value=0.0, conv_conv_activation='linear', conv_dense_activation='linear', conv_pre_output_dense_units=(8, 16),
                 conv_use_bias=True, deconv_conv_activation='linear', deconv_dense_activation='linear', deconv_pre_output_dense_units=(8, 16),
                 deconv_use_bias=True, use_resnet=False, upsampling_factor=None, data_format='channels_first',
                 conv_initializer_constraint_regularizer_options=None, deconv_initializer_constraint_regularizer_options=None, downsampling_method='conv',
                 conv_downsampling_kernel_size=None, pool_downsampling_method='max', use_batchnorm=False, batchnorm_trainable=True, store=None, ctx=None,
                 name='metalearning_bottleneck_deconv', device=None, seed=0):
        self._setup(ndims, downsampling_factor, filters, conv_kernel_size, n_convs, conv_padding_mode, conv_constant_padding_value, conv_conv_activation,
                    conv_dense_activation, conv_pre_output_dense_units, conv_use_bias, use_resnet, upsampling_factor, data_format, downsampling_method,
                    conv_downsampling_kernel_size, pool_downsampling_method, use_batchnorm, False, store, ctx, name, device, seed)
        self.upsample = metalearning_deconvupscale(self.up, filters, deconv_kernel_size, conv_activation=deconv_conv_activation, use_bias=deconv_use_bias,
                                                   dense_activations=deconv_dense_activation, pre_output_dense_units=deconv_pre_output_dense_units,
                                                   store=self.store, ctx=self.ctx, name=name + '/deconv')
        self.use_bn = use_batchnorm
        self.bn = None

    def build(self, cin, dense_features):
        self._build_stages(cin, dense_features)
        self.upsample.build(self.filters, dense_features)
        if self.use_bn:
            self.bn = _BatchNorm(self.store, self.ctx, self.name + '/bn', self.filters)
        self.built = True
        if self.own_store:
            self.finalize()

    def forward(self, x, dense_input, training=True, out=None):
        """out: optional NHWC destination, possibly a channel slice of a wider buffer (the chained models write a block's result next to the
        tensor it is concatenated with instead of copying both, models/Homogeneous_Poisson_NN_Metalearning.py:249-256)."""
        if not self.built:
            self.build(x.shape[3], dense_input.shape[1])
        if self.own_store:
            self._pre()
        o = self._down_and_stages(x, dense_input, training)
        o = self.upsample.forward(o, dense_input, self.out_hw(x.shape[1], x.shape[2]), training)
        if self.bn is not None:
            return self.bn.forward(o, training, out=out)
        return o if out is None else ops.axpby(1.0, o, 0.0, out)

    def backward(self, dy):
        d = self.bn.backward(dy) if self.bn is not None else dy
        d, dd = self.upsample.backward(d)
        dx = self._backward_stages_and_down(d, dd)
        if self.own_store:
            self._post()
        return dx, dd

    def __call__(self, inputs, training=False):
        x, dense_input = inputs
        return self.forward(x.permute(0, 2, 3, 1).contiguous(), dense_input.contiguous(), training=training).permute(0, 3, 1, 2)


class metalearning_bottleneck_block_multilinearupsample(_metalearning_bottleneck):
    """blocks/metalearning_bottleneck_block.py:120-191; call([x, dense_input, domain_sizes]); without resnets every plain stage is followed by
    a BatchNormalization layer that the reference counts as a stage (:164-167); up-sampling = Upsample(ndims) i.e. bilinear tf.image.resize."""

    def __init__(self, ndims, downsampling_factor, filters, conv_kernel_size, n_convs=1, conv_padding_mode='constant', conv_constant_padding_value=0.0,
                 conv_conv_activation='linear', conv_dense_activation='linear', conv_pre_output_dense_units=(8, 16), conv_use_bias=True, use_resnet=False,
                 upsampling_factor=None, data_format='channels_first', conv_initializer_constraint_regularizer_options=None, downsampling_method='conv',
                 conv_downsampling_kernel_size=None, pool_downsampling_method='max', use_batchnorm=False, batchnorm_trainable=True, store=None, ctx=None,
                 name='metalearning_bottleneck_multilinear', device=None, seed=0):
        self._setup(ndims, downsampling_factor, filters, conv_kernel_size, n_convs, conv_padding_mode, conv_constant_padding_value, conv_conv_activation,
                    conv_dense_activation, conv_pre_output_dense_units, conv_use_bias, use_resnet, upsampling_factor, data_format, downsampling_method,
                    conv_downsampling_kernel_size, pool_downsampling_method, use_batchnorm, True, store, ctx, name, device, seed)

    def build(self, cin, dense_features):
        self._build_stages(cin, dense_features)
        self.built = True
        if self.own_store:
            self.finalize()

    def forward(self, x, dense_input, training=True, out=None):
        """out: optional NHWC destination (a channel slice of a wider buffer is fine: the resize kernel takes the channel stride)."""
        if not self.built:
            self.build(x.shape[3], dense_input.shape[1])
        if self.own_store:
            self._pre()
        o = self._down_and_stages(x, dense_input, training)
        self._coarse = (o.shape[1], o.shape[2])
        return ops.resize_fwd(o, self.out_hw(x.shape[1], x.shape[2]), 'bilinear', out=out)

    def backward(self, dy):
        d = ops.resize_bwd(dy, self._coarse, 'bilinear')
        dd = ops.zeros((dy.shape[0], self._dense_features()), dy.device)
        dx = self._backward_stages_and_down(d, dd)
        if self.own_store:
            self._post()
        return dx, dd

    def _dense_features(self):
        first = self.down if self.down is not None else self.stages[0][1]
        conv = first.convs[0] if isinstance(first, metalearning_resnet) else first
        return conv.store.w[conv.name + '/dense0/kernel'].shape[0]

    def __call__(self, inputs, training=False):
        x, dense_input, _domain_sizes = inputs
        return self.forward(x.permute(0, 2, 3, 1).contiguous(), dense_input.contiguous(), training=training).permute(0, 3, 1, 2)
