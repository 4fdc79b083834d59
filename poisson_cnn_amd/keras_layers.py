"""Stand-alone layers and blocks with the reference's Keras call conventions (poisson_CNN/layers, poisson_CNN/blocks, poisson_CNN/utils):

    from poisson_cnn_amd.keras_layers import (Conv2D, apply_advanced_padding_and_call_conv_layer, resnet, bottleneck_block_multilinearupsample,
                                              bottleneck_block_deconvupsample, deconvupscale, Upsample, Scaling, SpatialPyramidPool,
                                              JacobiIterationLayer, MergeWithAttention)

Every class takes the reference's constructor kwargs (strings such as "tf.nn.leaky_relu" are accepted for activations), builds lazily on
its first call (own parameter bucket, Keras-default initialisers), and is called as `layer(inputs)` with the reference's input lists:

    deconvupscale([x, output_shape_int32[4]])         layers/deconvupscale.py:100-101
    Upsample([x, domain_sizes, out_hw])               layers/Upsample.py:33
    Scaling([x_to_scale, other])                      layers/Scaling.py:48-49
    SpatialPyramidPool(x)                             layers/SpatialPyramidPool.py:51
    JacobiIterationLayer([guess, rhs, dx])            layers/JacobiIterationLayer.py:57
    MergeWithAttention([x_0, ..., x_{n-1}])           layers/MergeWithAttention.py:30
    resnet(x)                                         blocks/resnet.py:29
    bottleneck_block_multilinearupsample([x, domain_sizes]),  bottleneck_block_deconvupsample(x)      blocks/bottleneck_block.py:70,100

Tensors are channels_first (N,C,H,W) at this API like the reference's configs (numpy arrays or torch tensors in, torch CUDA tensors out);
inside they are NHWC and every arithmetic op is a libpcnn HIP kernel.  Instead of a tape, `layer(inputs, training=True)` keeps what
`layer.backward(dy)` needs; backward returns the gradient(s) of the tensor input(s) and leaves the parameter gradients in `layer.gradients`
(name -> tensor, the same order as `layer.trainable_variables`).  A model file of the reference that composes these classes can therefore be
re-hosted by replacing its imports and writing the backward chain that Keras' tape would have recorded (models.py is the worked example).
"""
import numpy as np
import torch

from . import layers as L
from . import ops
from .utils import canonical_activation, split_indices


def _dev(device):
    if device is None and not torch.cuda.is_available():
        raise RuntimeError('the libpcnn layers need an AMD GPU (there is no CPU fallback)')
    return torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())


def _t(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return torch.as_tensor(x).to(device=device, dtype=torch.float32)


def _nhwc(x, device):
    x = _t(x, device)
    if x.dim() != 4:
        raise ValueError('expected a (N,C,H,W) tensor, got shape %s' % (tuple(x.shape),))
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(y):
    return y.permute(0, 3, 1, 2).contiguous()


class _Layer:
    """Lazy build, Keras-style weight access; I/O in the layer's data_format (channels_first as in every reference config, or channels_last =
    the kernels' own NHWC layout, in which case nothing is permuted)."""

    def __init__(self, data_format='channels_first', device=None, seed=0, name=None):
        if data_format not in ('channels_first', 'channels_last'):
            raise ValueError('data_format must be channels_first or channels_last')
        self.data_format = data_format
        self.device = _dev(device)
        self.seed = seed
        self.name = name or type(self).__name__
        self.built = False
        self.store, self.ctx = None, None

    # -- layout: user tensor <-> NHWC
    def _in(self, x):
        if self.data_format == 'channels_first':
            return _nhwc(x, self.device)
        x = _t(x, self.device)
        if x.dim() != 4:
            raise ValueError('expected a (N,H,W,C) tensor, got shape %s' % (tuple(x.shape),))
        return x.contiguous()

    def _out(self, y):
        return _nchw(y) if self.data_format == 'channels_first' else y.contiguous()

    # -- building
    def _begin_build(self):
        self.store, self.ctx = L.ParamStore(), L.Context()

    def _end_build(self):
        self.store.finalize(self.device)
        self.store.initialize(self.seed)
        self.built = True

    # -- weights (Keras order = declaration order)
    @property
    def weight_names(self):
        return self.store.names if self.store is not None else []

    @property
    def trainable_variables(self):
        return [self.store.w[n] for n in self.store.trainable_names()] if self.store is not None else []

    @property
    def gradients(self):
        return {n: self.store.g[n] for n in self.store.trainable_names()} if self.store is not None else {}

    def get_weights(self):
        return [self.store.w[n].detach().cpu().numpy().copy() for n in self.weight_names]

    def set_weights(self, weights):
        if isinstance(weights, dict):
            weights = [weights[n] for n in self.weight_names]
        if len(weights) != len(self.weight_names):
            raise ValueError('expected %d weight arrays, got %d' % (len(self.weight_names), len(weights)))
        for n, v in zip(self.weight_names, weights):
            v = np.asarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(self.store.w[n].shape):
                raise ValueError('shape mismatch for %s: %s vs %s' % (n, v.shape, tuple(self.store.w[n].shape)))
            self.store.w[n].copy_(torch.from_numpy(v))

    def count_params(self):
        return self.store.n_trainable if self.store is not None else 0

    def __call__(self, inputs, training=False):
        return self.call(inputs, training=training)

    def _pre(self):
        if self.store is not None and self.store.nbn:
            self.store.refresh_bn()

    def _post_backward(self):
        if self.ctx is not None:
            self.ctx.join()
        if self.store is not None and self.store.nbn:
            self.store.finish_bn_grads()


# ----------------------------------------------------------------------------------------------------------------- convolution
class Conv2D(_Layer):
    """tf.keras.layers.Conv2D(filters, kernel_size, strides=1, padding='valid'|'same', activation, use_bias) - what
    utils/choose_conv_layer.py:3-4 returns for ndims = 2.  On its own it pads like Keras; wrapped by
    apply_advanced_padding_and_call_conv_layer it becomes the reference's boundary-condition padded convolution."""

    def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format='channels_first', activation=None, use_bias=True,
                 kernel_initializer='glorot_uniform', bias_initializer='zeros', device=None, seed=0, name=None, **unused):
        super().__init__(data_format, device, seed, name)
        self.filters, self.kernel_size = int(filters), kernel_size
        self.strides = int(strides if np.isscalar(strides) else strides[0])
        self.padding = padding.lower()
        if self.padding not in ('valid', 'same'):
            raise ValueError('padding must be "valid" or "same"')
        self.activation = canonical_activation(activation)
        self.use_bias = use_bias
        self._advanced = None          # (padding_mode, constant_padding_value) once wrapped by apply_advanced_padding_and_call_conv_layer

    def _build(self, cin):
        self._begin_build()
        if self._advanced is not None:
            mode, val = self._advanced
            self.unit = L.ConvUnit(self.store, self.ctx, 'conv', self.kernel_size, cin, self.filters, padding_mode=mode, pad_value=val,
                                   activation=self.activation, use_bias=self.use_bias, stride=self.strides)
        else:
            if self.strides != 1:
                raise NotImplementedError('strided Conv2D is implemented in its advanced-padding form (apply_advanced_padding_and_call_conv_layer)')
            self.unit = L.ConvUnit(self.store, self.ctx, 'conv', self.kernel_size, cin, self.filters, pad='same', activation=self.activation,
                                   use_bias=self.use_bias)
        self._end_build()

    def call(self, x, training=False):
        x = self._in(x)
        if not self.built:
            self._build(x.shape[3])
        if self._advanced is None and self.padding == 'valid':
            k = self.unit.kh
            y = ops.conv2d_fwd(x, self.store.w['conv/kernel'], self.store.w['conv/bias'] if self.use_bias else None, pad_top=0, pad_left=0,
                               out_hw=(x.shape[1] - k + 1, x.shape[2] - self.unit.kw + 1), act=self.activation)
            if training:
                raise NotImplementedError('training through an un-padded VALID Conv2D is not part of the hot path; wrap it with '
                                          'apply_advanced_padding_and_call_conv_layer or use padding="same"')
            return self._out(y)
        return self._out(self.unit.forward(x, training=training))

    def backward(self, dy):
        dx = self.unit.backward(self._in(dy), need_dx=True)
        self._post_backward()
        return self._out(dx)


def apply_advanced_padding_and_call_conv_layer(padding_mode, conv_layer, constant_padding_value=0.0):
    """utils/apply_advanced_padding_and_call_conv_layer.py:3-21: forces the layer to VALID and returns the closure x -> conv(tf.pad(x)) with
    k//2 before and k//2 - (1 - k%2) after on each spatial axis, mode CONSTANT(value) / SYMMETRIC / REFLECT.  Here the padding happens
    inside the convolution kernel's tile loader; the returned callable also has `.backward(dy)`."""
    if not isinstance(conv_layer, Conv2D):
        raise TypeError('conv_layer must be a poisson_cnn_amd.keras_layers.Conv2D')
    if conv_layer.built:
        raise ValueError('wrap the layer before its first call')
    if padding_mode.upper() not in ops.PAD_MODES:
        raise ValueError('unknown padding mode ' + padding_mode)
    conv_layer.padding = 'valid'
    conv_layer._advanced = (padding_mode.upper(), float(constant_padding_value))

    def pad_and_apply_convolution(x, training=False):
        return conv_layer(x, training=training)
    pad_and_apply_convolution.backward = conv_layer.backward
    pad_and_apply_convolution.layer = conv_layer
    return pad_and_apply_convolution


# ----------------------------------------------------------------------------------------------------------------- blocks
def _conv_args(conv_args):
    a = dict(conv_args)
    filters, k = a.pop('filters'), a.pop('kernel_size')
    act, ub = a.pop('activation', None), a.pop('use_bias', True)
    df = a.pop('data_format', 'channels_first')
    for key in ('kernel_initializer', 'bias_initializer', 'kernel_regularizer', 'bias_regularizer', 'activity_regularizer', 'kernel_constraint',
                'bias_constraint', 'padding'):
        a.pop(key, None)
    if a:
        raise TypeError('unsupported convolution arguments: %s' % sorted(a))
    return int(filters), k, canonical_activation(act), ub, df


class resnet(_Layer):
    """blocks/resnet.py:6-39: o = conv0(x); [BN0]; o = conv1(o); [BN1]; o = x + o; o = conv2(o), three padded convolutions."""

    def __init__(self, ndims, use_batchnorm=False, batchnorm_trainable=True, padding_mode='constant', constant_padding_value=0.0, device=None, seed=0,
                 **conv_args):
        filters, k, act, ub, df = _conv_args(conv_args)
        super().__init__(df, device, seed)
        if ndims != 2:
            raise NotImplementedError('ndims = 2 only (the hot path of BASELINE.json)')
        self.args = dict(filters=filters, kernel_size=k, use_batchnorm=use_batchnorm, padding_mode=padding_mode,
                         constant_padding_value=constant_padding_value, activation=act, use_bias=ub)

    def call(self, x, training=False):
        x = self._in(x)
        if not self.built:
            if x.shape[3] != self.args['filters']:
                raise ValueError('resnet needs as many input channels as filters (%d vs %d): the skip connection adds them' % (x.shape[3], self.args['filters']))
            self._begin_build()
            self.block = L.resnet(self.store, self.ctx, 'resnet', **self.args)
            self._end_build()
        self._pre()
        return self._out(self.block.forward(x, training=training))

    def backward(self, dy):
        dx = self.block.backward(self._in(dy))
        self._post_backward()
        return self._out(dx)


class _bottleneck(_Layer):
    KIND = None

    def __init__(self, ndims, downsampling_factor, filters, conv_kernel_size, data_format='channels_first', conv_activation=None, conv_use_bias=True,
                 use_resnet=False, padding_mode='constant', constant_padding_value=0.0, n_convs=1, upsampling_factor=None,
                 conv_initializer_constraint_regularizer_options=None, downsampling_method='conv', conv_downsampling_kernel_size=None,
                 pool_downsampling_method='max', use_batchnorm=False, batchnorm_trainable=True, device=None, seed=0, **extra):
        super().__init__(data_format, device, seed)
        if ndims != 2:
            raise NotImplementedError('ndims = 2 only (the hot path of BASELINE.json)')
        self.kw = dict(downsampling_factor=downsampling_factor, filters=filters, conv_kernel_size=conv_kernel_size, n_convs=n_convs,
                       upsampling_factor=upsampling_factor, padding_mode=padding_mode, constant_padding_value=constant_padding_value,
                       conv_activation=canonical_activation(conv_activation), conv_use_bias=conv_use_bias, use_resnet=use_resnet,
                       downsampling_method=downsampling_method, conv_downsampling_kernel_size=conv_downsampling_kernel_size,
                       pool_downsampling_method=pool_downsampling_method, use_batchnorm=use_batchnorm)
        self.kw.update(extra)
        self.filters = filters

    def _run(self, x, training):
        x = self._in(x)
        if not self.built:
            self._begin_build()
            self.block = self.KIND(self.store, self.ctx, 'block', x.shape[3], **self.kw)
            self._end_build()
        self._pre()
        N, H, W, _ = x.shape
        Ho, Wo = self.block.out_hw(H, W)
        out = ops.empty((N, Ho, Wo, self.filters), x.device)
        self.block.forward_into(x, out, 1.0, 0.0, training=training)
        self._in_shape = tuple(x.shape)
        return self._out(out)

    def backward(self, dy):
        d_in = ops.zeros(self._in_shape, self.device)
        self.block.backward_from(self._in(dy), 1.0, d_in)
        self._post_backward()
        return self._out(d_in)


class bottleneck_block_multilinearupsample(_bottleneck):
    """blocks/bottleneck_block.py:8-86; call([x, domain_sizes]) - domain_sizes is unused for ndims = 2 (tf.image.resize path, layers/Upsample.py:55-59)."""
    KIND = L.bottleneck_block_multilinearupsample

    def __init__(self, *args, resize_method='bilinear', **kw):
        super().__init__(*args, resize_method=resize_method, **kw)

    def call(self, inputs, training=False):
        x, _domain_sizes = inputs
        return self._run(x, training)


class bottleneck_block_deconvupsample(_bottleneck):
    """blocks/bottleneck_block.py:88-118; call(x)."""
    KIND = L.bottleneck_block_deconvupsample

    def __init__(self, ndims, downsampling_factor, filters, conv_kernel_size, deconv_kernel_size, *args, deconv_activation=None, deconv_use_bias=True,
                 deconv_initializer_constraint_regularizer_options=None, **kw):
        if canonical_activation(deconv_activation) != 'linear':
            raise NotImplementedError('deconv_activation other than linear is not used by any reference config')
        super().__init__(ndims, downsampling_factor, filters, conv_kernel_size, *args, deconv_kernel_size=deconv_kernel_size, deconv_use_bias=deconv_use_bias, **kw)

    def call(self, x, training=False):
        return self._run(x, training)


# ----------------------------------------------------------------------------------------------------------------- layers
class deconvupscale(_Layer):
    """layers/deconvupscale.py:8-109: tf.nn.conv2d_transpose(x, K, output_shape, strides=upsample_ratio, padding='SAME') + bias + activation;
    kernel (k, k, filters, Cin); Keras-default (Glorot) initialiser for kernel AND bias (:37-38).  kernel_size == upsample_ratio (every reference
    config) runs on the dedicated per-tap MFMA kernels, any other kernel size as zero insertion + the fused pad+conv kernels (ops.deconv_*)."""

    def __init__(self, upsample_ratio, filters, kernel_size, data_format='channels_first', activation=None, use_bias=True, dimensions=None, device=None,
                 seed=0, **unused):
        super().__init__(data_format, device, seed)
        up = upsample_ratio if np.isscalar(upsample_ratio) else upsample_ratio[0]
        k = kernel_size if np.isscalar(kernel_size) else kernel_size[0]
        if (not np.isscalar(upsample_ratio) and len(set(upsample_ratio)) != 1) or (not np.isscalar(kernel_size) and len(set(kernel_size)) != 1):
            raise NotImplementedError('anisotropic upsample_ratio / kernel_size')
        if canonical_activation(activation) != 'linear':
            raise NotImplementedError('activation other than linear is not used by any reference config')
        self.up, self.k, self.filters, self.use_bias = int(up), int(k), int(filters), use_bias

    def call(self, inputs, training=False):
        x, output_shape = inputs
        x = self._in(x)
        if not self.built:
            self._begin_build()
            self.store.add('kernel', (self.k, self.k, self.filters, x.shape[3]), 'glorot')
            if self.use_bias:
                self.store.add('bias', (self.filters,), 'glorot')
            self._end_build()
        shp = [int(v) for v in (output_shape.tolist() if hasattr(output_shape, 'tolist') else output_shape)]
        if len(shp) != 4:
            raise ValueError('output_shape must have 4 entries ((N, C, H, W), or (N, H, W, C) for channels_last)')
        H, W = (shp[2], shp[3]) if self.data_format == 'channels_first' else (shp[1], shp[2])
        y = ops.deconv_fwd(x, self.store.w['kernel'], self.store.w['bias'] if self.use_bias else None, (H, W), self.up)
        self._saved = (x, (H, W)) if training else None
        return self._out(y)

    def backward(self, dy):
        x, _ = self._saved
        dy = self._in(dy)
        g = self.store.g
        ops.deconv_bwd_filter(x, dy, self.up, dk=g['kernel'], dbias=g['bias'] if self.use_bias else None, ws=self.ctx.ws, kernel_size=(self.k, self.k))
        return self._out(ops.deconv_bwd_data(dy, self.store.w['kernel'], (x.shape[1], x.shape[2]), self.up))


class Upsample(_Layer):
    """layers/Upsample.py:14-61 for ndims = 2: tf.image.resize(x, out_hw, method, antialias=False) between two layout transposes."""

    def __init__(self, ndims, data_format='channels_first', resize_method='bilinear', device=None):
        super().__init__(data_format, device)
        if ndims != 2:
            raise NotImplementedError('ndims = 2 only (the tfp.math.batch_interp_regular_nd_grid path is not part of the hot path)')
        self.method = str(resize_method).lower()
        if self.method not in ops.RESIZE:
            raise ValueError('unsupported resize method ' + str(resize_method))
        self.built = True

    def call(self, inputs, training=False):
        x, _domain_sizes, out_hw = inputs
        x = self._in(x)
        hw = tuple(int(v) for v in (out_hw.tolist() if hasattr(out_hw, 'tolist') else out_hw))
        self._coarse = (x.shape[1], x.shape[2])
        return self._out(ops.resize_fwd(x, hw, self.method))

    def backward(self, dy):
        return self._out(ops.resize_bwd(self._in(dy), self._coarse, self.method))


class SpatialPyramidPool(_Layer):
    """layers/SpatialPyramidPool.py:5-66: for every level (ly, lx) the map is cut into ly x lx bins (dataset/utils/split_indices.py: the first
    n % bins bins get one extra element) and each bin is reduced over (C, h, w) TOGETHER (tf.reduce_max / reduce_mean of the whole slice,
    :43-44) -> (N, sum ly*lx) features."""

    def __init__(self, levels, ndims, data_format='channels_first', pooling_type='average', receive_padded_values=False, device=None):
        super().__init__(data_format, device)
        if ndims != 2 or receive_padded_values:
            raise NotImplementedError('ndims = 2 without padding masks (the hot path of BASELINE.json)')
        self.levels = [[lv, lv] if isinstance(lv, int) else (list(lv) * 2 if len(lv) == 1 else list(lv)) for lv in levels]
        for lv in self.levels:
            if len(lv) != 2:
                raise ValueError('Each SPP level must have a pool size with ndims or 1 element(s). Got ' + str(len(lv)))
        pt = pooling_type.lower()
        if pt not in ('average', 'avg', 'max'):
            raise ValueError('pooling_type must be average or max')
        self.max = pt == 'max'
        self.built = True
        self._bins = {}

    def _bin_table(self, H, W):
        if (H, W) not in self._bins:
            bins = []
            for ly, lx in self.levels:
                iy, ix = split_indices(H, ly), split_indices(W, lx)
                if (np.diff(iy) <= 0).any() or (np.diff(ix) <= 0).any():
                    raise ValueError('a %dx%d pyramid level over a %dx%d map has empty bins' % (ly, lx, H, W))
                bins += [[iy[a], iy[a + 1], ix[b], ix[b + 1]] for a in range(ly) for b in range(lx)]
            self._bins[(H, W)] = torch.tensor(np.array(bins, dtype=np.int32), device=self.device)
        return self._bins[(H, W)]

    def call(self, x, training=False):
        x = self._in(x)
        bins = self._bin_table(x.shape[1], x.shape[2])
        if self.max:
            out, arg = ops.spp_max_fwd(x, bins)
            self._saved = (arg, tuple(x.shape))
        else:
            out = ops.spp_avg_fwd(x, bins)
            self._saved = (bins, tuple(x.shape))
        return out

    def backward(self, dout):
        a, shape = self._saved
        dout = _t(dout, self.device).contiguous()
        return self._out(ops.spp_max_bwd(a, dout, shape) if self.max else ops.spp_avg_bwd(a, dout, shape))


class Scaling(_Layer):
    """layers/Scaling.py:18-55: [x_to_scale, other] -> concat -> `stages` x (SAME conv, SAME average pool) -> SpatialPyramidPool(MAX) ->
    Dense 100 / 25 / 1 -> x_to_scale * (1 + gain).  convargs are those of the stage convolutions (filters, kernel_size, activation, ...)."""

    def __init__(self, ndims, stages=2, downsampling_ratio_per_stage=2, data_format='channels_first', padding='same', spp_levels=((2, 2), 3, 5), device=None,
                 seed=0, **convargs):
        super().__init__(data_format, device, seed)
        if ndims != 2 or padding.lower() != 'same':
            raise NotImplementedError('ndims = 2, padding = "same" (layers/Scaling.py defaults; the hot path of BASELINE.json)')
        filters, k, act, ub, _ = _conv_args(convargs)
        if not ub:
            raise NotImplementedError('Scaling convolutions without bias')
        self.kw = dict(stages=stages, downsampling_ratio_per_stage=downsampling_ratio_per_stage, spp_levels=spp_levels, filters=filters, kernel_size=k, activation=act)

    def call(self, inputs, training=False):
        x, other = inputs
        x, other = self._in(x), self._in(other)
        if x.shape[3] != 1 or other.shape[3] != 1:
            raise NotImplementedError('Scaling is implemented for one-channel inputs (the model scales its (N,1,H,W) output by the (N,1,H,W) rhs)')
        if not self.built:
            self._begin_build()
            self.layer = L.Scaling(self.store, self.ctx, 'scaling', **self.kw)
            self._end_build()
        return self._out(self.layer.forward(x, other, training=training))

    def backward(self, dy):
        """Gradient w.r.t. x_to_scale (`other` - the right-hand side in the model - gets none, as in the reference's graph)."""
        d = self.layer.backward(self._in(dy))
        self._post_backward()
        return self._out(d)


class MergeWithAttention(_Layer):
    """layers/MergeWithAttention.py:4-33: a list of n same-shape tensors -> sum_n x_n * sm[n, c] with the trainable attention_weights (n, C),
    sm = exp(w) / sum(exp(w)) normalised over ALL n*C entries (:31), 'uniform' initialiser (:27).  Built on the first call from the input list,
    or at construction when n_channels and n_inputs are both given (:12-19)."""

    def __init__(self, data_format='channels_first', n_channels=None, n_inputs=None, device=None, seed=0):
        super().__init__(data_format, device, seed)
        if (n_channels is None) != (n_inputs is None):
            raise ValueError('Both n_channels and n_inputs must be None, or both must be an integer value')
        if n_channels is not None:
            self._build(int(n_inputs), int(n_channels))

    def _build(self, n, C):
        self._begin_build()
        self.store.add('attention_weights', (n, C), 'uniform')
        self._end_build()
        self.n, self.C = n, C
        self._zero = torch.zeros(C, dtype=torch.float32, device=self.device)

    def _softmax(self):
        e = torch.exp(self.store.w['attention_weights'])
        return (e / e.sum()).contiguous()

    def call(self, inputs, training=False):
        xs = [self._in(x) for x in inputs]
        if not self.built:
            self._build(len(xs), xs[0].shape[-1])
        if len(xs) != self.n or any(tuple(x.shape) != tuple(xs[0].shape) for x in xs) or xs[0].shape[-1] != self.C:
            raise ValueError('MergeWithAttention was built for %d inputs of %d channels and needs same-shape inputs' % (self.n, self.C))
        sm = self._softmax()
        out = ops.channel_affine(xs[0], sm[0], self._zero)
        for i in range(1, self.n):
            ops.channel_affine(xs[i], sm[i], self._zero, residual=out, out=out)
        self._saved = (xs, sm) if training else None
        return self._out(out)

    def backward(self, dy):
        xs, sm = self._saved
        dy = self._in(dy)
        dsm = torch.empty_like(sm)
        dxs = []
        for i in range(self.n):
            dxs.append(self._out(ops.channel_affine(dy, sm[i], self._zero)))
            ops.epilogue_bwd(dy, xs[i], s_dy_a=dsm[i], ws=self.ctx.ws)      # dsm[i, c] = sum dy * x_i
        self.store.g['attention_weights'].copy_(sm * (dsm - (dsm * sm).sum()))   # softmax over the whole (n, C) table
        return dxs


class JacobiIterationLayer(_Layer):
    """layers/JacobiIterationLayer.py:7-66 for the model's second-order 3-point stencils: n_iterations weighted-Jacobi sweeps
    new = D^-1 (rhs - (L+U) guess) on the interior, the boundary ring kept; call([guess, rhs, dx]), dx (N, 2) or (N, 1)."""

    def __init__(self, stencil_sizes, orders, ndims=None, data_format='channels_first', n_iterations=5, device=None):
        super().__init__(data_format, device)
        ss = [stencil_sizes] * 2 if isinstance(stencil_sizes, int) else list(stencil_sizes)
        od = [orders] * 2 if isinstance(orders, int) else list(orders)
        if (ndims not in (None, 2)) or ss != [3, 3] or od != [2, 2]:
            raise NotImplementedError('JacobiIterationLayer is implemented for stencil_sizes [3,3], orders [2,2] (models/Homogeneous_Poisson_NN_Legacy.py:104)')
        self.layer = L.JacobiIterationLayer(n_iterations)
        self.built = True

    def call(self, inputs, training=False):
        guess, rhs, dx = inputs
        g, r = self._in(guess), self._in(rhs)
        if g.shape[3] != 1 or r.shape[3] != 1:
            raise ValueError('guess and rhs must have one channel')
        dx = _t(dx, self.device).reshape(g.shape[0], -1)
        dx2 = (torch.cat([dx, dx], 1) if dx.shape[1] == 1 else dx[:, :2]).contiguous()
        return self._out(self.layer.forward(g, r, dx2, training=training))

    def backward(self, dout):
        return self._out(self.layer.backward(self._in(dout)))
