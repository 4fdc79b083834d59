"""metalearning_conv (poisson_CNN/layers/metalearning_conv.py:52-169): a convolution whose per-sample filter and bias are
emitted by a small dense network from `dense_input`, applied sample by sample (the reference uses tf.map_fn over the batch).

Here the hyper-network runs on the libpcnn dense kernels and every sample's convolution / filter gradient / data gradient is
one launch of the fused MFMA conv kernels on that sample's (1,H,W,C) slice with its own (kh,kw,Cin,Cout) filter - a grouped
implicit GEMM by launch, which is what SURVEY.md row H16 asks for at "lower priority" (the layer is only reachable from
reference models that raise NameError at construction).  `metalearning_resnet` / `metalearning_bottleneck_block_*` /
`metalearning_deconvupscale` are compositions of this layer and are not built yet.
"""
import numpy as np
import torch

from . import layers as L
from . import ops
from .utils import canonical_activation


class metalearning_conv:
    def __init__(self, filters, kernel_size, previous_layer_filters, dense_input_features, strides=None, padding='valid', padding_mode='constant',
                 constant_padding_value=0.0, data_format='channels_first', dilation_rate=None, conv_activation='linear', use_bias=True,
                 dense_activations='linear', pre_output_dense_units=(8, 16), use_layernorm=False, store=None, ctx=None, name='metalearning_conv',
                 device=None, seed=0):
        if data_format != 'channels_first':
            raise NotImplementedError('channels_first only')
        if strides not in (None, 1) or dilation_rate not in (None, 1):
            raise NotImplementedError('strides / dilation other than 1 are not used by any reference config')
        if use_layernorm:
            raise NotImplementedError('use_layernorm=True is not implemented')
        self.k = int(kernel_size) if isinstance(kernel_size, int) else int(kernel_size[0])
        self.cin, self.cout = int(previous_layer_filters), int(filters)
        self.same = padding.lower() == 'same'
        self.mode = padding_mode.upper()
        self.pad_value = float(constant_padding_value)
        self.act = canonical_activation(conv_activation)
        self.use_bias = use_bias
        units = list(pre_output_dense_units) + [self.k * self.k * self.cin * self.cout + (self.cout if use_bias else 0)]
        acts = [dense_activations] * len(units) if isinstance(dense_activations, str) or dense_activations is None else list(dense_activations)
        self.own_store = store is None
        self.store = store if store is not None else L.ParamStore()
        self.ctx = ctx if ctx is not None else L.Context()
        self.dense = []
        din = int(dense_input_features)
        for i, (u, a) in enumerate(zip(units, acts)):
            self.dense.append(L.Dense(self.store, '%s/dense%d' % (name, i), din, u, a if a is not None else 'linear'))
            din = u
        if self.own_store:
            dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
            self.store.finalize(dev)
            self.store.initialize(seed)

    def _pads(self):
        # 'same': [ks//2, ks//2 if odd else ks//2-1] (metalearning_conv.py:103-107); 'valid': none
        return (self.k // 2, self.k // 2 - (1 - self.k % 2)) if self.same else (0, 0)

    def forward(self, x, dense_input, training=True):
        """x (N,H,W,Cin) NHWC, dense_input (N,F)."""
        N, H, W, _ = x.shape
        kb = dense_input
        for d in self.dense:
            kb = d.forward(kb, training)
        nk = self.k * self.k * self.cin * self.cout
        pt, pb = self._pads()
        Ho, Wo = H + pt + pb - self.k + 1, W + pt + pb - self.k + 1
        y = ops.empty((N, Ho, Wo, self.cout), x.device)
        for n in range(N):
            w = kb[n, :nk].view(self.k, self.k, self.cin, self.cout)
            b = kb[n, nk:] if self.use_bias else None
            ops.conv2d_fwd(x[n:n + 1], w, b, pad_top=pt, pad_left=pt, out_hw=(Ho, Wo), pad_mode=self.mode if self.same else 'CONSTANT',
                           pad_value=self.pad_value, act=self.act, out=y[n:n + 1])
        self.saved = (x, kb, y) if training else None
        return y

    def backward(self, dy, need_dx=True):
        """Returns (dx, d_dense_input); parameter gradients of the hyper-network go to store.g."""
        x, kb, y = self.saved
        self.saved = None
        N, H, W, _ = x.shape
        nk = self.k * self.k * self.cin * self.cout
        pt, pb = self._pads()
        dkb = ops.zeros(tuple(kb.shape), x.device)
        dx = ops.empty(tuple(x.shape), x.device) if need_dx else None
        mode = self.mode if self.same else 'CONSTANT'
        for n in range(N):
            dyn, yn = dy[n:n + 1], y[n:n + 1]
            dz = ops.empty(tuple(dyn.shape), x.device)
            ops.epilogue_bwd(dyn, yn if self.act != 'linear' else None, act=self.act, dz=dz, dbias=dkb[n, nk:] if self.use_bias else None, ws=self.ctx.ws)
            w = kb[n, :nk].view(self.k, self.k, self.cin, self.cout)
            ops.conv2d_wgrad(x[n:n + 1], dz, w.shape, pad_top=pt, pad_left=pt, pad_mode=mode, pad_value=self.pad_value,
                             out=dkb[n, :nk].view(self.k, self.k, self.cin, self.cout), ws=self.ctx.ws)
            if need_dx:
                wf = ops.flip_transpose_weights(w, out=self.ctx.wflip((self.k, self.k, self.cout, self.cin), x.device))
                if mode == 'CONSTANT':
                    ops.conv2d_fwd(dz, wf, None, pad_top=self.k - 1 - pt, pad_left=self.k - 1 - pt, out_hw=(H, W), out=dx[n:n + 1])
                else:
                    gp = ops.conv2d_fwd(dz, wf, None, pad_top=self.k - 1, pad_left=self.k - 1, out_hw=(H + pt + pb, W + pt + pb))
                    ops.pad_fold_bwd(gp, (H, W), ((pt, pb), (pt, pb)), mode, out=dx[n:n + 1])
        d = dkb
        for i, lyr in enumerate(reversed(self.dense)):
            d = lyr.backward(d, need_dx=True)
        return dx, d

    def __call__(self, inputs):
        """Reference call convention: [conv_input (N,C,H,W), dense_input (N,F)] -> (N,filters,H',W')."""
        x, dense_input = inputs
        x = x.permute(0, 2, 3, 1).contiguous()
        return self.forward(x, dense_input.contiguous(), training=False).permute(0, 3, 1, 2)
