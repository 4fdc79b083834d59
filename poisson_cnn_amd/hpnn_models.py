"""The two non-legacy homogeneous-Poisson models of the reference, hosted on the libpcnn HIP kernels:

    Homogeneous_Poisson_NN_Metalearning   models/Homogeneous_Poisson_NN_Metalearning.py:84-325   (train/hpnn_train.py model_type 'cnn_metalearning')
    Homogeneous_Poisson_NN                models/Homogeneous_Poisson_NN.py:11-252                 (train/hpnn_train.py model_type 'cnn')

Both share one graph: [rhs | position embeddings] -> pre-bottleneck convolutions -> a CHAIN of bottleneck blocks sorted by descending
down-sampling factor, every block after the first reading concat([initial, previous block's result]) -> final convolutions on
concat([initial, last result]) -> per-sample output scaling.  In the metalearning model every convolution's filter is emitted per sample by a
hyper-network from dense_inp = [dx / domain_sizes, domain_sizes / max_domain_size] (reference :225, :240).

Neither reference class can be constructed as shipped: the metalearning constructor's signature lists bottleneck_deconv_config /
bottleneck_multilinear_config while its body (and its own __main__ example, :334-377) uses `bottleneck_upsampling` and `bottleneck_config`
(:100, :104, :128-141); Homogeneous_Poisson_NN takes those two but reads an undefined `bottleneck_deconv_config` (:59-66).  What is built here is
the evident intent - the signature of the sibling class / of the __main__ example - and otherwise the code AS WRITTEN, including
Homogeneous_Poisson_NN appending the BatchNormalization layers of its final stages to the PRE-bottleneck chain (:83-86).  There is no TF here
and no reference test for either model: PARITY UNPINNED, checked against the fp64 autograd restatement in oracle/hpnn_chain.py.

Call conventions as in models.py: model([rhs (N,1,H,W), dx (N,2)]) -> (N,1,H,W); compile(loss, optimizer); train_step(((rhs, dx), y)).
"""
import copy

import torch

from . import layers as L
from . import metalearning as M
from . import ops
from .models import _ModelBase, _as_device, process_normalizations, process_output_scaling_modes
from .utils import get_init_arguments_from_config

_CONV_FIELDS = (['filters', 'kernel_sizes'], ['filters', 'kernel_size'])


def _bottleneck_fields(cfg, deconv):
    """Per-block list fields of the bottleneck config and the constructor argument each one feeds (reference :130-131, :135-136)."""
    pairs = [('downsampling_factors', 'downsampling_factor'), ('upsampling_factors', 'upsampling_factor'), ('conv_kernel_sizes', 'conv_kernel_size'),
             ('n_convs', 'n_convs'), ('conv_downsampling_kernel_sizes', 'conv_downsampling_kernel_size')]
    if deconv:
        pairs.insert(3, ('deconv_kernel_sizes', 'deconv_kernel_size'))
    pairs = [(c, a) for c, a in pairs if c in cfg]
    return [c for c, _ in pairs], [a for _, a in pairs]


class _ChainModel(_ModelBase):
    """Input assembly, output scaling and the training step the two models share; subclasses provide _forward_body / _backward_body."""

    def _init_common(self, ndims, data_format, input_normalization, output_scaling, use_batchnorm, bottleneck_upsampling, device, what):
        if ndims != 2:
            raise NotImplementedError('ndims = 2 only (the hot path of BASELINE.json)')
        if data_format not in ('channels_first', 'channels_last'):
            raise ValueError('data_format must be channels_first or channels_last')
        if bottleneck_upsampling not in ('deconv', 'multilinear'):
            raise ValueError('Invalid bottleneck block upsampling method')
        if device is None and not torch.cuda.is_available():
            raise RuntimeError('%s needs an AMD GPU: the HIP kernels are the only compute path' % what)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.ndims, self.data_format = 2, data_format
        self.input_normalization = process_normalizations(input_normalization)       # kept, unused: the reference's normalisation is commented out
        self.output_scaling = process_output_scaling_modes(output_scaling)
        if self.output_scaling['match_peak_laplacian_magnitude_to_peak_rhs']:
            raise NotImplementedError('match_peak_laplacian_magnitude_to_peak_rhs calls an undefined method in the reference (scale_outputs)')
        self.use_batchnorm = use_batchnorm
        self.bottleneck_upsampling = bottleneck_upsampling
        self.store = L.ParamStore()
        self.ctx = L.Context()
        self.optimizer = self.loss_fn = self.grad_sync = None

    # ------------------------------------------------------------------ scale_outputs (reference :197-214)
    def _scale_factors(self, rhs, max_domain_sizes):
        f = None
        if self.output_scaling['rhs_max_magnitude']:
            _, fac = ops.set_max_magnitude_fwd(rhs.contiguous(), 1.0)               # fac = 1 / max|rhs| per sample
            f = 1.0 / fac
        if self.output_scaling['max_domain_size_squared']:
            f = max_domain_sizes ** 2 if f is None else f * max_domain_sizes ** 2
        return f

    def call(self, inp, training=False):
        rhs, dx = inp
        rhs, dx = _as_device(rhs, self.device), _as_device(dx, self.device)
        if rhs.dim() != 4 or rhs.shape[1] != 1:
            raise ValueError('rhs must have shape (N,1,H,W)')
        N, _, H, W = rhs.shape
        dx = dx.reshape(N, -1)
        if dx.shape[1] == 1:
            dx = dx.repeat(1, 2)
        if dx.shape[1] != 2:
            raise ValueError('dx must have shape (N,2)')
        self.store.refresh_bn()
        # tiny (N, 2..4) host-side assemblies: domain sizes (:166-169), dense_inp (:240)
        domain_sizes = dx * torch.tensor([float(H - 1), float(W - 1)], device=self.device)
        max_domain_sizes = domain_sizes.amax(dim=1)
        dense_inp = torch.cat([dx / domain_sizes, domain_sizes / max_domain_sizes[:, None]], 1).contiguous()
        x = ops.assemble_input(rhs.view(N, H, W), True)                              # [rhs | cos(pi y) | cos(pi x)] (:216-223, :241)
        out = self._forward_body(x, dense_inp, training)
        if out.shape[3] != 1:
            raise ValueError('the last final convolution must have 1 filter')
        fac = self._scale_factors(rhs, max_domain_sizes)
        self._saved_scale = None
        if self.output_scaling['soln_max_magnitude']:                                # takes precedence over the product of factors (:204-207)
            pre = out
            out, _ = ops.set_max_magnitude_fwd(pre, 1.0)
            self._saved_scale = ('max', pre) if training else None
        elif fac is not None:
            pre, fac = out, (fac - 1.0).contiguous()                                 # pcnn_sample_scale computes x * (1 + g) (the Scaling layer's form)
            out = ops.sample_scale_fwd(pre, fac)
            self._saved_scale = ('fac', pre, fac) if training else None
        return out.view(N, 1, H, W)

    def backward(self, dpred):
        N, _, H, W = dpred.shape
        d = dpred.contiguous().view(N, H, W, 1)
        if self._saved_scale is not None:
            if self._saved_scale[0] == 'max':
                d = ops.set_max_magnitude_bwd(self._saved_scale[1], d, 1.0)
            else:
                d, _ = ops.sample_scale_bwd(self._saved_scale[1], self._saved_scale[2], d)
            self._saved_scale = None
        self._backward_body(d)
        self.ctx.join()
        self.store.finish_bn_grads()

    def _train_step_cf(self, data):
        """reference :279-309: loss_fn(y_true, y_pred, rhs, dx), gradients, optimizer step; returns loss and mse."""
        (rhs, dx), y_true = data
        rhs, dx, y_true = _as_device(rhs, self.device), _as_device(dx, self.device), _as_device(y_true, self.device)
        dx = dx.reshape(dx.shape[0], -1)
        if dx.shape[1] == 1:
            dx = dx.repeat(1, 2)
        pred = self.call([rhs, dx], training=True)
        loss, dpred = self.loss_fn.value_and_grad(y_true, pred, rhs, dx.contiguous())
        self.backward(dpred)
        if self.grad_sync is not None:
            self.grad_sync(self.store.flat_g)
        self.optimizer.apply_gradients()
        return self._logs(loss, self.loss_fn.mse_metric(y_true, pred))

    def _finish_init(self, seed):
        self.store.finalize(self.device)
        self.store.initialize(seed)


# =====================================================================================================================
class Homogeneous_Poisson_NN_Metalearning(_ChainModel):
    model_name = 'Homogeneous_Poisson_NN_Metalearning'

    def __init__(self, ndims=2, data_format='channels_first', final_convolutions_config=None, pre_bottleneck_convolutions_config=None,
                 bottleneck_upsampling='deconv', bottleneck_config=None, input_normalization=None, output_scaling=None, use_batchnorm=False,
                 postsmoother_iterations=5, device=None, seed=0):
        if pre_bottleneck_convolutions_config is None:
            raise ValueError('Provide a config for pre bottleneck convolutions')
        if bottleneck_config is None:
            raise ValueError('Provide a config for bottleneck blocks')
        if final_convolutions_config is None:
            raise ValueError('Provide a config for final convolutions')
        self._init_common(ndims, data_format, input_normalization, output_scaling, use_batchnorm, bottleneck_upsampling, device, self.model_name)
        S, C, DF = self.store, self.ctx, 4                                           # dense_inp has 2 * ndims features
        shared = dict(store=S, ctx=C, dense_input_features=DF)
        # pre-bottleneck convolutions (:113-126): metalearning_conv('same') [+ BatchNormalization]
        self.pre = []
        cin = 3
        pre = copy.deepcopy(pre_bottleneck_convolutions_config)
        for k in range(len(pre['filters'])):
            a = get_init_arguments_from_config(pre, k, *_CONV_FIELDS)
            self.pre.append(M.metalearning_conv(padding='same', previous_layer_filters=cin, name='pre/conv%d' % k, **shared, **a))
            cin = int(a['filters'])
            if use_batchnorm:
                self.pre.append(M._BatchNorm(S, C, 'pre/bn%d' % k, cin))
        self.c0 = cin
        # bottleneck blocks (:128-142), sorted by descending down-sampling factor
        bc = copy.deepcopy(bottleneck_config)
        deconv = bottleneck_upsampling == 'deconv'
        cls = M.metalearning_bottleneck_block_deconvupsample if deconv else M.metalearning_bottleneck_block_multilinearupsample
        fields, args = _bottleneck_fields(bc, deconv)
        blocks = [cls(ndims=2, use_batchnorm=use_batchnorm, store=S, ctx=C, name='bottleneck%d' % k, **get_init_arguments_from_config(bc, k, fields, args))
                  for k in range(len(bc['downsampling_factors']))]
        self.bottleneck_blocks = sorted(blocks, key=lambda b: b.f, reverse=True)
        self.F = int(bc['filters'])
        for i, b in enumerate(self.bottleneck_blocks):
            b.build(self.c0 if i == 0 else self.c0 + self.F, DF)
        # final convolutions (:144-158)
        fc = copy.deepcopy(final_convolutions_config)
        nst = len(fc['filters'])
        self.final_regular_conv_stages = nreg = fc.pop('final_regular_conv_stages', 2)
        self.final_meta, self.final_regular = [], []
        cin = self.c0 + self.F
        for k in range(nst - nreg):
            a = get_init_arguments_from_config(fc, k, *_CONV_FIELDS)
            self.final_meta.append(M.metalearning_conv(padding='same', previous_layer_filters=cin, name='final/stage%d/conv' % k, **shared, **a))
            cin = int(a['filters'])
            self.final_meta.append(M.metalearning_resnet(use_batchnorm=use_batchnorm, previous_layer_filters=cin, name='final/stage%d/res' % k, **shared, **a))
        for j, k in enumerate(range(nst - nreg, nst)):
            self.final_regular.append(L.ConvUnit(S, C, 'final/out%d' % j, fc['kernel_sizes'][k], cin, fc['filters'][k], pad='same', activation='linear',
                                                 use_bias=fc.get('use_bias', True)))
            cin = int(fc['filters'][k])
        self._finish_init(seed)

    def _forward_body(self, x, d, training):
        for lyr in self.pre:
            x = lyr.forward(x, training) if isinstance(lyr, M._BatchNorm) else lyr.forward(x, d, training)
        initial, c0 = x, self.c0
        N, H, W, _ = initial.shape
        inp = initial
        for b in self.bottleneck_blocks:            # tf.concat([initial, result], axis=1) (:249, :253, :256): every block writes its result IN PLACE
            cat = ops.empty((N, H, W, c0 + self.F), initial.device)      # next to a copy of `initial` (one buffer per block: each is a saved input)
            ops.axpby(1.0, initial, 0.0, cat[..., :c0])
            b.forward(inp, d, training, out=cat[..., c0:])
            inp = cat
        o = inp
        for lyr in self.final_meta:
            o = lyr.forward(o, d, training)
        for lyr in self.final_regular:
            o = lyr.forward(o, training=training)
        return o

    def _backward_body(self, d):
        c0 = self.c0
        for lyr in reversed(self.final_regular):
            d = lyr.backward(d, inplace=True)
        for lyr in reversed(self.final_meta):
            d, _ = lyr.backward(d)
        d_initial, dres = d[..., :c0].contiguous(), d[..., c0:].contiguous()
        for b in reversed(self.bottleneck_blocks[1:]):
            dcat, _ = b.backward(dres)
            ops.axpby(1.0, dcat[..., :c0], 1.0, d_initial)
            dres = dcat[..., c0:].contiguous()
        dx0, _ = self.bottleneck_blocks[0].backward(dres)
        d = ops.axpby(1.0, dx0, 1.0, d_initial)
        for i, lyr in enumerate(reversed(self.pre)):
            if isinstance(lyr, M._BatchNorm):
                d = lyr.backward(d)
            else:
                d, _ = lyr.backward(d, need_dx=(i < len(self.pre) - 1))


# =====================================================================================================================
class Homogeneous_Poisson_NN(_ChainModel):
    model_name = 'Homogeneous_Poisson_NN'

    def __init__(self, ndims=2, data_format='channels_first', final_convolutions_config=None, pre_bottleneck_convolutions_config=None,
                 bottleneck_upsampling='deconv', bottleneck_config=None, use_batchnorm=False, input_normalization=None, output_scaling=None,
                 device=None, seed=0):
        if pre_bottleneck_convolutions_config is None:
            raise ValueError('Provide a config for pre bottleneck convolutions')
        if bottleneck_config is None:
            raise ValueError('Provide a config for bottleneck blocks')
        if final_convolutions_config is None:
            raise ValueError('Provide a config for final convolutions')
        self._init_common(ndims, data_format, input_normalization, output_scaling, use_batchnorm, bottleneck_upsampling, device, self.model_name)
        S, C = self.store, self.ctx
        C.enable_side_stream()

        def conv_chain(cfg, prefix, cin, bn_for):
            cfg = copy.deepcopy(cfg)
            mode, val = cfg.pop('padding_mode', 'CONSTANT'), cfg.pop('constant_padding_value', 0.0)
            units = []
            for k in range(len(cfg['filters'])):
                a = get_init_arguments_from_config(cfg, k, *_CONV_FIELDS)
                units.append(L.ConvUnit(S, C, '%s/conv%d' % (prefix, k), a['kernel_size'], cin, a['filters'], padding_mode=mode, pad_value=val,
                                        activation=a.get('activation', 'linear'), use_bias=a.get('use_bias', True),
                                        bn_name=('%s/bn%d' % (prefix, k)) if bn_for(k) else None))
                cin = int(a['filters'])
            return units, cin

        # pre-bottleneck convolutions, each followed by its BatchNormalization (:43-55) - fused into the convolution's epilogue
        self.pre, self.c0 = conv_chain(pre_bottleneck_convolutions_config, 'pre', 3, lambda k: use_batchnorm)
        # as written (:83-86): the final stages' BatchNormalization layers (all but the last stage) are appended to the PRE-bottleneck chain
        nfinal = len(final_convolutions_config['filters'])
        self.pre_extra_bn = [M._BatchNorm(S, C, 'pre/extra_bn%d' % k, self.c0) for k in range(nfinal - 1)] if use_batchnorm else []
        # bottleneck blocks (:58-70)
        bc = copy.deepcopy(bottleneck_config)
        deconv = bottleneck_upsampling == 'deconv'
        cls = L.bottleneck_block_deconvupsample if deconv else L.bottleneck_block_multilinearupsample
        fields, args = _bottleneck_fields(bc, deconv)
        self.F = int(bc['filters'])
        cfgs = sorted([(k, get_init_arguments_from_config(bc, k, fields, args)) for k in range(len(bc['downsampling_factors']))],
                      key=lambda ka: ka[1]['downsampling_factor'], reverse=True)
        self.bottleneck_blocks = [cls(S, C, 'bottleneck%d' % k, self.c0 if i == 0 else self.c0 + self.F, use_batchnorm=use_batchnorm, **a)
                                  for i, (k, a) in enumerate(cfgs)]
        # final convolutions (:72-82): padded convolutions, no BatchNormalization of their own (see above)
        self.final, cout = conv_chain(final_convolutions_config, 'final', self.c0 + self.F, lambda k: False)
        self._finish_init(seed)

    def _cat(self, initial):
        N, H, W, c0 = initial.shape
        cat = ops.empty((N, H, W, c0 + self.F), initial.device)
        ops.axpby(1.0, initial, 0.0, cat[..., :c0])
        return cat

    def _forward_body(self, x, d, training):
        for c in self.pre:
            x = c.forward(x, training=training)
        for bn in self.pre_extra_bn:
            x = bn.forward(x, training)
        initial, c0 = x, self.c0
        inp = initial
        for b in self.bottleneck_blocks:                                              # each block writes its result next to `initial` (:196-203)
            cat = self._cat(initial)
            b.forward_into(inp, cat[..., c0:], 1.0, 0.0, training=training)
            inp = cat
        o = inp
        for c in self.final:
            o = c.forward(o, training=training)
        return o

    def _backward_body(self, d):
        c0 = self.c0
        for c in reversed(self.final):
            d = c.backward(d, inplace=True)
        d_initial = d[..., :c0].contiguous()
        dres = d[..., c0:]
        for i in range(len(self.bottleneck_blocks) - 1, -1, -1):
            b = self.bottleneck_blocks[i]
            if i == 0:
                b.backward_from(dres, 1.0, d_initial)
            else:
                dcat = torch.zeros(d.shape[:3] + (c0 + self.F,), dtype=torch.float32, device=d.device)
                b.backward_from(dres, 1.0, dcat)
                ops.axpby(1.0, dcat[..., :c0], 1.0, d_initial)
                dres = dcat[..., c0:]
        d = d_initial
        for bn in reversed(self.pre_extra_bn):
            d = bn.backward(d)
        for i, c in enumerate(reversed(self.pre)):
            d = c.backward(d, need_dx=(i < len(self.pre) - 1), inplace=True)
