"""Layers and blocks of the Poisson CNN hot path, hosted on the libpcnn HIP kernels.

Mirrors poisson_CNN/layers and poisson_CNN/blocks (reference paths relative to poisson_CNN/):
  ConvUnit                      <- utils/apply_advanced_padding_and_call_conv_layer.py:3-21 + Keras Conv2D (+ the fused
                                   BatchNormalization and residual add that follow it in blocks/resnet.py:29-39)
  resnet                        <- blocks/resnet.py:6-39
  bottleneck_block_deconvupsample / bottleneck_block_multilinearupsample <- blocks/bottleneck_block.py:8-118
  deconvupscale                 <- layers/deconvupscale.py:8-109
  Upsample                      <- layers/Upsample.py:14-61
  Scaling / SpatialPyramidPool  <- layers/Scaling.py:18-55, layers/SpatialPyramidPool.py:5-66
  JacobiIterationLayer          <- layers/JacobiIterationLayer.py:7-66
Instead of a tracing autograd every layer has an explicit forward() that saves what backward() needs; data is NHWC
float32 on the GPU, weights live in one flat parameter bucket (ParamStore) so that the optimizer step and the
data-parallel all-reduce are single launches.
"""
import numpy as np
import torch

from . import ops
from .utils import advanced_pad_amounts, canonical_activation, glorot_limit, same_pad_amounts, split_indices


# ----------------------------------------------------------------------------- parameters
class ParamStore:
    """Ordered registry of parameters.  Physical layout: one flat fp32 buffer [conv/dense/deconv params | all BN gammas |
    all BN betas] for the trainable ones (+ same-shaped grad / Adam moment buffers) and [all BN means | all BN variances]
    for the moving statistics; `names` keeps the Keras-like declaration order used by get_weights()/set_weights()."""

    def __init__(self):
        self.specs = []      # (name, shape, init, kind)
        self.bn_layers = []  # (name, channels)
        self.finalized = False

    def add(self, name, shape, init, kind='w'):
        assert not self.finalized
        self.specs.append((name, tuple(int(s) for s in shape), init, kind))
        return name

    def add_bn(self, name, c):
        off = sum(ch for _, ch in self.bn_layers)
        self.bn_layers.append((name, c))
        for suffix, init, kind in (('gamma', 'ones', 'bn_gamma'), ('beta', 'zeros', 'bn_beta'), ('moving_mean', 'zeros', 'bn_mean'),
                                   ('moving_variance', 'ones', 'bn_var')):
            self.add('%s/%s' % (name, suffix), (c,), init, kind)
        return off

    @property
    def names(self):
        return [s[0] for s in self.specs]

    def finalize(self, device):
        nbn = sum(c for _, c in self.bn_layers)
        self.nbn = nbn
        nw = sum(int(np.prod(s[1])) for s in self.specs if s[3] == 'w')
        self.n_trainable = nw + 2 * nbn
        self.flat_w = torch.zeros(self.n_trainable, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(self.n_trainable, dtype=torch.float32, device=device)
        self.flat_stats = torch.zeros(max(2 * nbn, 1), dtype=torch.float32, device=device)
        self.bn_scale = torch.zeros(max(nbn, 1), dtype=torch.float32, device=device)
        self.bn_shift = torch.zeros(max(nbn, 1), dtype=torch.float32, device=device)
        self.bn_s1 = torch.zeros(max(nbn, 1), dtype=torch.float32, device=device)
        self.bn_s2 = torch.zeros(max(nbn, 1), dtype=torch.float32, device=device)
        self.w, self.g = {}, {}
        off = 0
        bn_off = {'bn_gamma': 0, 'bn_beta': 0, 'bn_mean': 0, 'bn_var': 0}
        for name, shape, init, kind in self.specs:
            n = int(np.prod(shape))
            if kind == 'w':
                self.w[name] = self.flat_w[off:off + n].view(shape)
                self.g[name] = self.flat_g[off:off + n].view(shape)
                off += n
            elif kind in ('bn_gamma', 'bn_beta'):
                base = nw + (0 if kind == 'bn_gamma' else nbn) + bn_off[kind]
                self.w[name] = self.flat_w[base:base + n]
                self.g[name] = self.flat_g[base:base + n]
                bn_off[kind] += n
            else:
                base = (0 if kind == 'bn_mean' else nbn) + bn_off[kind]
                self.w[name] = self.flat_stats[base:base + n]
                bn_off[kind] += n
        self.gamma_all = self.flat_w[nw:nw + nbn]
        self.beta_all = self.flat_w[nw + nbn:nw + 2 * nbn]
        self.dgamma_all = self.flat_g[nw:nw + nbn]
        self.dbeta_all = self.flat_g[nw + nbn:nw + 2 * nbn]
        self.mean_all = self.flat_stats[0:nbn]
        self.var_all = self.flat_stats[nbn:2 * nbn]
        self.finalized = True

    def trainable_names(self):
        return [s[0] for s in self.specs if s[3] in ('w', 'bn_gamma', 'bn_beta')]

    def filter_version(self):
        """The weights version a convolution of this bucket passes to libpcnn (ops.filter_version).  torch-side writes to the bucket - copy_, an
        optimizer of torch.optim through autograd.Differentiable, a user's in-place op - move Tensor._version and are noticed here; the library's own
        writers (train.Adam / SGD, parallel's broadcast) call ops.weights_changed() themselves."""
        v = self.flat_w._version
        if v != getattr(self, '_seen_version', None):
            self._seen_version = v
            ops.weights_changed()
        return ops.filter_version()

    def __del__(self):
        try:
            ops.weights_released()
        except Exception:
            pass

    def initialize(self, seed=0):
        """Keras defaults: Glorot-uniform kernels (and deconv bias, layers/deconvupscale.py:37-38,60-62), zero biases,
        BN gamma=1 beta=0 mean=0 var=1."""
        rng = np.random.default_rng(seed)
        for name, shape, init, kind in self.specs:
            if init == 'glorot':
                lim = glorot_limit(shape)
                v = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif init == 'zeros':
                v = np.zeros(shape, dtype=np.float32)
            elif init == 'ones':
                v = np.ones(shape, dtype=np.float32)
            elif init == 'uniform':                                  # Keras' 'uniform' = RandomUniform(-0.05, 0.05)
                v = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
            else:
                raise ValueError(init)
            self.w[name].copy_(torch.from_numpy(v))

    def refresh_bn(self):
        if self.nbn:
            ops.bn_fold(self.gamma_all, self.beta_all, self.mean_all, self.var_all, self.bn_scale, self.bn_shift)

    def finish_bn_grads(self):
        if self.nbn and not getattr(self, 'bn_training', False):
            ops.bn_fold_bwd(self.bn_s1, self.bn_s2, self.mean_all, self.var_all, self.dgamma_all, self.dbeta_all)


class Context:
    """Per-model scratch: workspace + the scratch filter used by data-gradient convolutions - one set PER STREAM (the coarse bottleneck branches
    of the homogeneous model run on streams of their own, models.Homogeneous_Poisson_NN_Legacy.call), so two streams never share a scratch buffer."""

    def __init__(self):
        self._ws = {}
        self._wflips = {}
        # Weight gradients run on a second HIP stream: their HBM-bound pre-pass (abs-max, fp16 plane conversion) and the kernel itself
        # overlap with the clock-bound data-gradient convolution of the same layer on the main stream (PCNN_WGRAD_STREAM=0 disables).
        import os
        self.use_side = False                                  # the model classes switch it on (they join() at the end of backward())
        self.side_allowed = os.environ.get('PCNN_WGRAD_STREAM', '1') != '0'
        self.side = None
        self.ws_side = ops.Workspace()
        self.side_reads = {}                                   # data_ptr -> event: tensors a side-stream weight gradient is still reading
        self.branch_streams = []                               # streams of the coarse bottleneck branches (created on demand)

    @staticmethod
    def _stream_key():
        return torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0

    @property
    def ws(self):
        """The scratch workspace of the CURRENT stream."""
        return self._ws.setdefault(self._stream_key(), ops.Workspace())

    def drop_stream(self, stream_ptr):
        """Forgets the scratch buffers kept for a stream that no longer exists (graphs._Captured.close: a recycled stream pointer must not
        inherit them, and a dict of captured graphs must not grow the process's memory without bound)."""
        self._ws.pop(int(stream_ptr), None)
        self._wflips.pop(int(stream_ptr), None)

    def all_scratch(self):
        """Every scratch buffer this context holds right now (a captured hipGraph keeps them alive, graphs._Captured)."""
        return [w.buf for w in list(self._ws.values()) + [self.ws_side] if w.buf is not None] + list(self._wflips.values())

    def enable_side_stream(self):
        self.use_side = self.side_allowed

    def side_stream(self):
        if not self.use_side:
            return None
        if self.side is None:
            self.side = torch.cuda.Stream()
        return self.side

    def branch_stream(self, k):
        while len(self.branch_streams) <= k:
            self.branch_streams.append(torch.cuda.Stream())
        return self.branch_streams[k]

    def join(self):
        """Main stream waits for the weight gradients enqueued on the side stream (call before the gradients are consumed)."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.side_reads.clear()

    def before_inplace_write(self, t):
        """The main stream is about to overwrite `t` in place: if a weight gradient on the side stream still reads it (its dz can alias the
        incoming gradient when the layer's epilogue is trivial - linear activation, no BN), wait for that launch first."""
        ev = self.side_reads.pop(t.data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def wflip(self, shape, device):
        n = int(np.prod(shape))
        key = self._stream_key()
        buf = self._wflips.get(key)
        if buf is None or buf.numel() < n:
            buf = self._wflips[key] = torch.empty(n, dtype=torch.float32, device=device)
        return buf[:n].view(shape)


# ----------------------------------------------------------------------------- conv unit
class ConvUnit:
    """tf.pad + Conv2D(VALID) + bias + activation [+ BatchNormalization] [+ residual] = ONE fused kernel launch."""

    def __init__(self, store, ctx, name, k, cin, cout, *, pad='advanced', padding_mode='CONSTANT', pad_value=0.0, activation='linear',
                 use_bias=True, bn_name=None, stride=1):
        """k: int (square kernel) or (kh, kw); a Conv1D of the reference is a (1, k) kernel on an (N, 1, L, C) tensor.
        stride > 1 (blocks/bottleneck_block.py:28-34): the reference pads as for stride 1 and lets the VALID convolution stride, i.e. it keeps
        every stride-th output - computed here as the fused stride-1 launch followed by pcnn_subsample (unused by the shipped configs)."""
        self.store, self.ctx, self.name = store, ctx, name
        self.stride = int(stride)
        if self.stride > 1 and pad != 'advanced':
            raise NotImplementedError('strided convolutions are implemented for the advanced-padding form (the only strided use in the reference)')
        self.kh, self.kw = (int(k), int(k)) if np.isscalar(k) else (int(k[0]), int(k[1]))
        self.k = self.kw
        self.cin, self.cout = int(cin), int(cout)
        amounts = advanced_pad_amounts if pad == 'advanced' else same_pad_amounts
        self.pads_y, self.pads_x = amounts(self.kh), amounts(self.kw)
        self.pads = self.pads_x
        self.mode = padding_mode.upper() if pad == 'advanced' else 'CONSTANT'
        if self.mode not in ops.PAD_MODES:
            raise ValueError('unknown padding mode ' + padding_mode)
        self.pad_value = float(pad_value) if pad == 'advanced' else 0.0
        self.act = canonical_activation(activation)
        self.use_bias = use_bias
        store.add(name + '/kernel', (self.kh, self.kw, self.cin, self.cout), 'glorot')
        if use_bias:
            store.add(name + '/bias', (self.cout,), 'zeros')
        self.bn_name = bn_name
        if bn_name is not None:
            self.bn_off = store.add_bn(bn_name, self.cout)
        self.saved = None

    def _bn(self):
        if self.bn_name is None:
            return None, None
        s = self.store
        return s.bn_scale[self.bn_off:self.bn_off + self.cout], s.bn_shift[self.bn_off:self.bn_off + self.cout]

    def forward(self, x, residual=None, out=None, training=True, x_absmax=None):
        """x_absmax: optional 1-element device tensor holding max|x| (the producing layer's `out_absmax`): saved for the weight gradient,
        which would otherwise spend a pass over x on it (split math mode).  After the call `self.out_absmax` holds this layer's."""
        if self.stride > 1:
            assert residual is None and out is None
            full = self._forward(x, training=training, x_absmax=x_absmax)
            self.out_absmax = None
            self._full_hw = (full.shape[1], full.shape[2])
            return ops.subsample(full, self.stride)
        return self._forward(x, residual=residual, out=out, training=training, x_absmax=x_absmax)

    def _forward(self, x, residual=None, out=None, training=True, x_absmax=None):
        self.out_absmax = None
        ops.sync_flipped_filters(self.store.filter_version())
        w = self.store.w[self.name + '/kernel']
        b = self.store.w[self.name + '/bias'] if self.use_bias else None
        sc, sh = self._bn()
        N, H, W, _ = x.shape
        if out is None:
            out = ops.empty((N, H, W, self.cout), x.device)
        if self.bn_name is not None and training and getattr(self.store, 'bn_training', False):
            # training-mode BatchNormalization: batch statistics need the whole activation first (two passes)
            a = ops.empty((N, H, W, self.cout), x.device)
            ops.conv2d_fwd(x, w, b, pad_top=self.pads_y[0], pad_left=self.pads_x[0], pad_mode=self.mode, pad_value=self.pad_value, act=self.act, out=a,
                           w_version=self.store.filter_version())
            sw = self.store.w
            _, stats = ops.bn_train_forward(a, sw[self.bn_name + '/gamma'], sw[self.bn_name + '/beta'], sw[self.bn_name + '/moving_mean'],
                                            sw[self.bn_name + '/moving_variance'], residual=residual, out=out, ws=self.ctx.ws)
            self.saved = (x, a, stats, x_absmax)
            return out
        need_a = training and (sc is not None or residual is not None)
        a = ops.empty((N, H, W, self.cout), x.device) if need_a else None
        if training and ops.get_math_mode() == 'split_f16':       # only the split-mode weight gradient scales by max|x| / max|dz|
            self.out_absmax = ops.empty((1,), x.device)          # a fresh buffer per call: a layer may run several times per step
        ops.conv2d_fwd(x, w, b, pad_top=self.pads_y[0], pad_left=self.pads_x[0], pad_mode=self.mode, pad_value=self.pad_value, act=self.act,
                       bn_scale=sc, bn_shift=sh, residual=residual, out=out, act_out=a, y_absmax=self.out_absmax, w_version=self.store.filter_version())
        if training:
            self.saved = (x, a if a is not None else out, None, x_absmax)   # without BN/residual the output itself is the activation
        return out

    def _flipped(self, w):
        """(flipped / transposed filter (kh, kw, Cout, Cin), weights version): a buffer of the layer's own - a stable address, which is what lets libpcnn
        keep ITS spectrum across calls - rewritten when the weights version moves: by ops.sync_flipped_filters for all layers at once (before the first
        convolution under the new version), here only for a layer's first backward pass or with the cache switched off."""
        ver = self.store.filter_version()
        wf = getattr(self, '_wf', None)
        if wf is None or wf.device != w.device:
            self._wf = torch.empty((self.kh, self.kw, self.cout, self.cin), dtype=torch.float32, device=w.device)
            self._wf_ver = None
            ops.register_flipped(self)
        if ver and self._wf_ver is not None and self._wf_ver != ver:
            # a new weights version met in a BACKWARD pass (another model's optimizer step between this layer's forward and backward): the cached
            # convolution below makes the handle refresh every kept filter, so every registered layer's flipped filter must be current first
            ops.sync_flipped_filters(ver)
        if not ver or self._wf_ver != ver:
            self._reflip(ver)
        return self._wf, ver

    def _reflip(self, ver):
        ops.flip_transpose_weights(self.store.w[self.name + '/kernel'], out=self._wf)
        self._wf_ver = ver

    def post_spec(self, want_raw=False):
        """This layer's activation backward (dz = dy act'(a), dbias = sum dz) as an offer to the data-gradient kernel of the layer that CONSUMES
        its output (ops.Post; pass it as `post=` to that layer's backward, then call this layer's backward with dz_ready=post.applied).  None
        where there is nothing to fuse or the epilogue carries more than that (training-mode BatchNormalization, strides, a linear layer without
        BatchNormalization).  An inference-mode BatchNormalization behind the activation rides along as bn_scale + the two sums its gamma / beta
        gradients need: only the fold route of SYMMETRIC / REFLECT layers takes such an offer (ops.pad_fold_bwd_post), the spectral kernels leave it."""
        if self.saved is None or self.stride > 1 or self.saved[2] is not None:
            return None
        sc, _ = self._bn()
        if sc is None and self.act == 'linear':
            return None
        s = self.store
        s1 = s.bn_s1[self.bn_off:self.bn_off + self.cout] if sc is not None else None
        s2 = s.bn_s2[self.bn_off:self.bn_off + self.cout] if sc is not None else None
        return ops.Post(self.saved[1], self.act, s.g[self.name + '/bias'] if self.use_bias else None, want_raw, bn_scale=sc, s_dy_a=s1, s_dy=s2)

    def backward(self, dy, need_dx=True, inplace=False, add_to=None, dz_ready=False, post=None):
        """add_to: optional tensor added to the returned input gradient inside the data-gradient kernel's epilogue (a skip connection's
        gradient); it may be overwritten.  dz_ready: dy already IS dz - the consumer's data-gradient kernel applied this layer's post_spec()
        (and wrote its bias gradient).  post: the producing layer's post_spec(), taken by the spectral data-gradient kernel where it can
        (post.applied tells)."""
        if self.stride > 1:
            dy, inplace = ops.subsample_bwd(dy, self._full_hw, self.stride), True
        x, a, bn_stats, x_absmax = self.saved
        self.saved = None
        s, g = self.store, self.store.g
        N, H, W, _ = x.shape
        sc, _ = self._bn()
        if bn_stats is not None:   # training-mode BN: da first, then the plain conv epilogue
            dy = ops.bn_train_backward(dy, a, bn_stats, g[self.bn_name + '/gamma'], g[self.bn_name + '/beta'], ws=self.ctx.ws)
            sc, inplace = None, True
        dense = ops._ld(dy) == self.cout
        dz = dy if (inplace and dense) else ops.empty((N, H, W, self.cout), dy.device)
        s1 = s.bn_s1[self.bn_off:self.bn_off + self.cout] if sc is not None else None
        s2 = s.bn_s2[self.bn_off:self.bn_off + self.cout] if sc is not None else None
        trivial = self.act == 'linear' and sc is None
        amax = None                    # max|dz|: a by-product of the epilogue pass that the split-mode weight gradient would otherwise recompute
        if dz_ready:
            assert bn_stats is None                                   # (an inference-mode BN's scale and sums were part of the offer: ops.Post.bn_scale)
            dz = dy
        elif trivial and not self.use_bias:
            dz = dy
        else:
            if not trivial and ops.get_math_mode() == 'split_f16':
                amax = ops.empty((1,), dy.device)             # fresh per call: read later by the side-stream weight gradient
            ops.epilogue_bwd(dy, a if (self.act != 'linear' or sc is not None) else None, act=self.act, bn_scale=sc, dz=None if trivial else dz,
                             dbias=g[self.name + '/bias'] if self.use_bias else None, s_dy_a=s1, s_dy=s2, ws=self.ctx.ws, dz_absmax=amax)
            if trivial:
                dz = dy
        w = s.w[self.name + '/kernel']
        kh, kw = self.kh, self.kw
        wf, wver = None, 0
        if need_dx:
            # wide filters: both gradients in one call on the spectral route (the spectrum of dz is shared); None = not eligible
            wf, wver = self._flipped(w)
            res = add_to if (self.mode == 'CONSTANT') else None
            out = ops.conv2d_bwd_fused(x, dz, w.shape, wf, pad_top=self.pads_y[0], pad_left=self.pads_x[0], pad_mode=self.mode, pad_value=self.pad_value,
                                       dw=g[self.name + '/kernel'], residual=res, post=post, w_version=wver)
            if out is not None:
                if self.mode == 'CONSTANT':
                    return out
                if post is not None and not post.applied:              # SYMMETRIC / REFLECT: the fold back onto the grid takes the producer's activation backward
                    dzp = ops.pad_fold_bwd_post(out, (H, W), (self.pads_y, self.pads_x), self.mode, post, add_to=add_to, ws=self.ctx.ws)
                    if dzp is not None:
                        return dzp
                if add_to is not None and add_to.is_contiguous():
                    self.ctx.before_inplace_write(add_to)
                    return ops.pad_fold_bwd(out, (H, W), (self.pads_y, self.pads_x), self.mode, out=add_to, accumulate=True)
                dx = ops.pad_fold_bwd(out, (H, W), (self.pads_y, self.pads_x), self.mode)
                return dx if add_to is None else ops.axpby(1.0, add_to, 1.0, dx)
        side = self.ctx.side_stream()
        if side is None:
            ops.conv2d_wgrad(x, dz, w.shape, pad_top=self.pads_y[0], pad_left=self.pads_x[0], pad_mode=self.mode, pad_value=self.pad_value,
                             out=g[self.name + '/kernel'], ws=self.ctx.ws, x_absmax=x_absmax, dz_absmax=amax)
        else:
            ev = torch.cuda.Event()
            ev.record()                                        # dz (and max|dz|) are complete on the main stream here
            with torch.cuda.stream(side):
                side.wait_event(ev)
                for t in (x, dz, x_absmax, amax):              # keep the allocator from recycling them while the side stream reads
                    if t is not None:
                        t.record_stream(side)
                ops.conv2d_wgrad(x, dz, w.shape, pad_top=self.pads_y[0], pad_left=self.pads_x[0], pad_mode=self.mode, pad_value=self.pad_value,
                                 out=g[self.name + '/kernel'], ws=self.ctx.ws_side, x_absmax=x_absmax, dz_absmax=amax)
                done = torch.cuda.Event()
                done.record()
            self.ctx.side_reads[dz.data_ptr()] = done          # see Context.before_inplace_write
        if not need_dx:
            return None
        # wf: the flipped filter formed above (the layer's own buffer) - no second launch
        if self.mode == 'CONSTANT':
            if post is not None and not post.applied:                  # narrow layers: the producer's activation backward in the data-gradient kernel's epilogue
                dxp = ops.conv2d_dgrad_post(dz, wf, pad_top=kh - 1 - self.pads_y[0], pad_left=kw - 1 - self.pads_x[0], out_hw=(H, W), residual=add_to, post=post)
                if dxp is not None:
                    return dxp
            return ops.conv2d_fwd(dz, wf, None, pad_top=kh - 1 - self.pads_y[0], pad_left=kw - 1 - self.pads_x[0], residual=add_to, w_version=wver)
        gp = ops.conv2d_fwd(dz, wf, None, pad_top=kh - 1, pad_left=kw - 1, out_hw=(H + kh - 1, W + kw - 1), w_version=wver)
        if post is not None and not post.applied:
            dzp = ops.pad_fold_bwd_post(gp, (H, W), (self.pads_y, self.pads_x), self.mode, post, add_to=add_to, ws=self.ctx.ws)
            if dzp is not None:
                return dzp
        if add_to is not None and add_to.is_contiguous():
            self.ctx.before_inplace_write(add_to)
            return ops.pad_fold_bwd(gp, (H, W), (self.pads_y, self.pads_x), self.mode, out=add_to, accumulate=True)
        dx = ops.pad_fold_bwd(gp, (H, W), (self.pads_y, self.pads_x), self.mode)
        return dx if add_to is None else ops.axpby(1.0, add_to, 1.0, dx)


class resnet:
    """blocks/resnet.py:6-39: o = conv0(x); [BN0]; o = conv1(o); [BN1]; o = x + o; o = conv2(o)."""

    def __init__(self, store, ctx, name, filters, kernel_size, *, use_batchnorm=False, padding_mode='constant', constant_padding_value=0.0,
                 activation='linear', use_bias=True):
        kw = dict(padding_mode=padding_mode, pad_value=constant_padding_value, activation=activation, use_bias=use_bias)
        # declaration order follows the reference: the three convs, then batchnorm0, batchnorm1
        self.c0 = ConvUnit(store, ctx, name + '/conv0', kernel_size, filters, filters, **kw)
        self.c1 = ConvUnit(store, ctx, name + '/conv1', kernel_size, filters, filters, **kw)
        self.c2 = ConvUnit(store, ctx, name + '/conv2', kernel_size, filters, filters, **kw)
        if use_batchnorm:
            for i, c in enumerate((self.c0, self.c1)):
                c.bn_name = '%s/bn%d' % (name, i)
                c.bn_off = store.add_bn(c.bn_name, filters)

    def _fusable(self, x, out):
        """The narrow stages (3 x 3, 4 / 8 channels, zero padding, no BatchNormalization) run as ONE launch (ops.resnet3_fwd, fp32 math mode; at 12
        channels the recomputed halo costs what the saved tensor passes bring: tools/probe_stage.py)."""
        cs = (self.c0, self.c1, self.c2)
        c = self.c0
        return (all(u.kh == 3 and u.kw == 3 and u.cin == c.cin and u.cout == c.cin and u.mode == 'CONSTANT' and u.pad_value == 0.0 and u.stride == 1
                    and u.bn_name is None and u.act == c.act and u.pads_y == (1, 1) and u.pads_x == (1, 1) for u in cs)
                and x.is_contiguous() and x.shape[3] == c.cin and (out is None or out.is_contiguous())
                and ops.get_math_mode() == 'fp32' and ops.resnet3_eligible(c.cin, c.act))

    def forward(self, x, out=None, training=True, x_absmax=None):
        if self._fusable(x, out):
            w = self.c0.store.w
            wb = [(w[u.name + '/kernel'], w[u.name + '/bias'] if u.use_bias else None) for u in (self.c0, self.c1, self.c2)]
            y, o0, a1, o1 = ops.resnet3_fwd(x, wb[0][0], wb[0][1], wb[1][0], wb[1][1], wb[2][0], wb[2][1], act=self.c0.act, training=training, out=out)
            if training:                                     # what the three launches would have saved: the backward pass runs unchanged
                self.c0.saved, self.c1.saved, self.c2.saved = (x, o0, None, None), (o0, a1, None, None), (o1, y, None, None)
            self.c0.out_absmax = self.c1.out_absmax = self.c2.out_absmax = self.out_absmax = None
            return y
        o = self.c0.forward(x, training=training, x_absmax=x_absmax)
        o = self.c1.forward(o, residual=x, training=training, x_absmax=self.c0.out_absmax)
        o = self.c2.forward(o, out=out, training=training, x_absmax=self.c1.out_absmax)
        self.out_absmax = self.c2.out_absmax
        return o

    def post_spec(self, want_raw=False):
        return self.c2.post_spec(want_raw)

    def backward(self, dy, inplace=False, dz_ready=False, post=None):
        """Where the data gradients run on the spectral route, each convolution's data-gradient kernel also applies the activation backward of
        the convolution before it (ops.Post): conv2's produces dz1 AND the raw gradient for the skip connection, conv1's produces dz0, conv0's
        (+ skip) takes the caller's `post` - 7 of the block's 12 activation-backward tensor passes disappear."""
        p1 = self.c1.post_spec(want_raw=True)
        d1 = self.c2.backward(dy, inplace=inplace, dz_ready=dz_ready, post=p1)           # gradient at (x + BN1(a1))
        ready1 = p1 is not None and p1.applied
        d1raw = p1.raw if ready1 else d1
        p0 = self.c0.post_spec()
        d0 = self.c1.backward(d1, inplace=ready1, dz_ready=ready1, post=p0)              # not fused: d1 is still needed for the skip connection
        ready0 = p0 is not None and p0.applied
        return self.c0.backward(d0, inplace=True, add_to=d1raw, dz_ready=ready0, post=post)   # + the skip connection's gradient, fused into the epilogue


# ----------------------------------------------------------------------------- bottleneck blocks
class _bottleneck_base:
    """blocks/bottleneck_block.py:9-66: down-sampling (pool, or a strided convolution), n_convs convolution stages (plain convolutions,
    each followed by a BatchNormalization layer when use_batchnorm - which the reference counts as a stage of its own, :52-55 - or resnet
    blocks), then the up-sampling of the subclass."""

    def __init__(self, store, ctx, name, cin, *, downsampling_factor, filters, conv_kernel_size, n_convs=1, upsampling_factor=None,
                 padding_mode='constant', constant_padding_value=0.0, conv_activation='linear', conv_use_bias=True, use_resnet=False,
                 downsampling_method='conv', conv_downsampling_kernel_size=None, pool_downsampling_method='max', use_batchnorm=False, **unused):
        self.name, self.f = name, int(downsampling_factor)
        self.up = int(upsampling_factor) if upsampling_factor is not None else self.f
        self.downsampling_factor = self.f
        self.filters = filters
        self.method = downsampling_method.lower()
        self.pool = pool_downsampling_method.lower()
        kw = dict(padding_mode=padding_mode, pad_value=constant_padding_value, activation=conv_activation, use_bias=conv_use_bias)
        self.down_conv = None
        self.stages = []                                                # ConvUnit / resnet objects in call order (a conv's BN is fused into it)
        n_layers = 0                                                    # the reference's len(self.conv_layers): BN layers count
        if self.method == 'conv':
            if conv_downsampling_kernel_size is None:
                raise ValueError('conv_downsampling_kernel_size is required for downsampling_method="conv"')
            self.down_conv = ConvUnit(store, ctx, name + '/downsample', conv_downsampling_kernel_size, cin, filters, stride=self.f, **kw)
            c = filters
        elif self.method == 'pool':
            c = cin
            if use_resnet:
                self.stages.append(ConvUnit(store, ctx, name + '/conv0', conv_kernel_size, cin, filters, **kw))
                n_layers, c = 1, filters
        else:
            raise ValueError('Downsampling method can only be conv or pool')
        i = 0
        while n_layers < n_convs:
            if use_resnet:
                self.stages.append(resnet(store, ctx, '%s/res%d' % (name, i), filters, conv_kernel_size, use_batchnorm=use_batchnorm, padding_mode=padding_mode,
                                          constant_padding_value=constant_padding_value, activation=conv_activation, use_bias=conv_use_bias))
                n_layers += 1
            else:
                self.stages.append(ConvUnit(store, ctx, '%s/conv%d' % (name, i), conv_kernel_size, c, filters,
                                            bn_name=('%s/bn%d' % (name, i)) if use_batchnorm else None, **kw))
                n_layers += 2 if use_batchnorm else 1
                c = filters
            i += 1
        # names kept from round 1 for the shipped (pool + resnet) configuration: conv0, res0, res1, ...
        self.conv0 = self.stages[0] if (self.method == 'pool' and use_resnet) else None
        self.res = [st for st in self.stages if isinstance(st, resnet)]

    def _down_and_convs(self, x, training, pooled=None):
        """pooled: the already down-sampled input (the model's pooling pyramid, models.py) - its gradient is then RETURNED by
        _backward_convs_and_down instead of being pooled back into d_in."""
        self.x = x if training else None
        self._pooled_given = pooled is not None
        hint = None
        if self.down_conv is not None:
            o = self.down_conv.forward(x, training=training)
        elif pooled is not None:
            o = pooled
        else:
            o = ops.pool2d_fwd(x, self.f, self.pool)
        for st in self.stages:
            o = st.forward(o, training=training, x_absmax=hint)
            hint = st.out_absmax
        return o

    def _backward_convs_and_down(self, dcoarse, d_in):
        d = dcoarse
        for st in reversed(self.stages):
            d = st.backward(d, inplace=True)
        ret = None
        if self.down_conv is not None:
            dx = self.down_conv.backward(d, inplace=True)
            ops.axpby(1.0, dx, 1.0, d_in)
        elif self._pooled_given:
            ret = d
        else:
            ops.pool2d_bwd(self.x, d, self.f, self.pool, dx=d_in, accumulate=True)
        self.x = None
        return ret

    def out_hw(self, H, W):
        return int((H / self.f) * self.up), int((W / self.f) * self.up)   # blocks/bottleneck_block.py:82,109


class bottleneck_block_deconvupsample(_bottleneck_base):
    """blocks/bottleneck_block.py:88-118 + layers/deconvupscale.py."""

    def __init__(self, store, ctx, name, cin, *, deconv_kernel_size, deconv_use_bias=True, **kw):
        super().__init__(store, ctx, name, cin, **kw)
        self.dk = int(deconv_kernel_size)           # != upsample ratio: ops.deconv_* compose it from zero insertion + the fused pad+conv kernels
        self.store = store
        self.ctx = ctx
        self.deconv_use_bias = deconv_use_bias
        store.add(name + '/deconv/kernel', (self.dk, self.dk, self.filters, self.filters), 'glorot')
        if deconv_use_bias:
            store.add(name + '/deconv/bias', (self.filters,), 'glorot')

    # The branch in two halves - its down-sampled convolution stages and the up-sampling that accumulates into the merge buffer - so that the model
    # can run the first half of a coarse branch on a stream of its own (models.Homogeneous_Poisson_NN_Legacy.call / backward).
    def forward_convs(self, x, training=True, pooled=None):
        return self._down_and_convs(x, training, pooled)

    def forward_up(self, o, hw, merged, alpha, beta, training=True):
        self.note_upsampled(o, training)
        self.up_into(o, hw, merged, alpha, beta)

    def up_into(self, o, hw, merged, alpha, beta):
        """the up-sampling alone (no state): merged = beta merged + alpha deconv(o); `o` / `merged` may be the same batch slice of the branch output / merge buffer"""
        assert self.out_hw(*hw) == tuple(hw), 'deconv branch must restore the input resolution'
        ops.deconv_fwd(o, self.store.w[self.name + '/deconv/kernel'], self.store.w[self.name + '/deconv/bias'] if self.deconv_use_bias else None,
                       tuple(hw), self.up, alpha=alpha, beta=beta, out=merged)

    def note_upsampled(self, o, training=True):
        self.coarse = o if training else None

    def forward_into(self, x, merged, alpha, beta, training=True, pooled=None):
        self.forward_up(self.forward_convs(x, training, pooled), (x.shape[1], x.shape[2]), merged, alpha, beta, training)

    def backward_up(self, dmerged, alpha):
        g = self.store.g
        k = self.store.w[self.name + '/deconv/kernel']
        o = self.coarse
        self.coarse = None
        ops.deconv_bwd_filter(o, dmerged, self.up, alpha=alpha, dk=g[self.name + '/deconv/kernel'],
                              dbias=g[self.name + '/deconv/bias'] if self.deconv_use_bias else None, ws=self.ctx.ws, kernel_size=(self.dk, self.dk))
        return ops.deconv_bwd_data(dmerged, k, (o.shape[1], o.shape[2]), self.up, alpha=alpha)

    def backward_convs(self, dcoarse, d_in):
        return self._backward_convs_and_down(dcoarse, d_in)

    def backward_from(self, dmerged, alpha, d_in):
        return self.backward_convs(self.backward_up(dmerged, alpha), d_in)


class bottleneck_block_multilinearupsample(_bottleneck_base):
    """blocks/bottleneck_block.py:8-86 + layers/Upsample.py (tf.image.resize, half-pixel centres)."""

    def __init__(self, store, ctx, name, cin, *, resize_method='bilinear', **kw):
        super().__init__(store, ctx, name, cin, **kw)
        self.method = resize_method.lower()
        if self.method not in ops.RESIZE:
            raise ValueError('unsupported resize method ' + resize_method)

    def forward_convs(self, x, training=True, pooled=None):
        return self._down_and_convs(x, training, pooled)

    def forward_up(self, o, hw, merged, alpha, beta, training=True):
        self.note_upsampled(o, training)
        self.up_into(o, hw, merged, alpha, beta)

    def up_into(self, o, hw, merged, alpha, beta):
        assert self.out_hw(*hw) == tuple(hw)
        ops.resize_fwd(o, tuple(hw), self.method, alpha=alpha, beta=beta, out=merged)

    def note_upsampled(self, o, training=True):
        """what forward_up remembers for the backward pass (the caller may up-sample `o` itself: ops.resize_fwd_multi, batch chunks)"""
        self.coarse_hw = (o.shape[1], o.shape[2])

    def forward_into(self, x, merged, alpha, beta, training=True, pooled=None):
        self.forward_up(self.forward_convs(x, training, pooled), (x.shape[1], x.shape[2]), merged, alpha, beta, training)

    def backward_up(self, dmerged, alpha):
        return ops.resize_bwd(dmerged, self.coarse_hw, self.method, alpha=alpha)

    def backward_convs(self, dcoarse, d_in):
        return self._backward_convs_and_down(dcoarse, d_in)

    def backward_from(self, dmerged, alpha, d_in):
        return self.backward_convs(self.backward_up(dmerged, alpha), d_in)


# ----------------------------------------------------------------------------- dense / scaling / jacobi
class Dense:
    """tf.keras.layers.Dense.  use_bias=False allocates no bias (the metalearning hyper-networks pass their `use_bias` into every Dense layer,
    layers/metalearning_conv.py:113,128).  activation 'softmax' = the linear layer followed by the row-softmax kernel."""

    def __init__(self, store, name, din, units, activation='linear', use_bias=True):
        self.store, self.name, self.act = store, name, canonical_activation(activation, dense=True)
        self.softmax = self.act == 'softmax'
        if self.softmax:
            self.act = 'linear'
        self.use_bias = bool(use_bias)
        store.add(name + '/kernel', (din, units), 'glorot')
        if self.use_bias:
            store.add(name + '/bias', (units,), 'zeros')

    def forward(self, x, training=True):
        y = ops.dense_fwd(x, self.store.w[self.name + '/kernel'], self.store.w[self.name + '/bias'] if self.use_bias else None, self.act)
        if self.softmax:
            y = ops.softmax_fwd(y)
        self.saved = (x, y) if training else None
        return y

    def backward(self, dy, need_dx=True):
        x, y = self.saved
        self.saved = None
        g = self.store.g
        g[self.name + '/kernel'].zero_()
        db = None
        if self.use_bias:
            db = g[self.name + '/bias']
            db.zero_()
        if self.softmax:
            dy = ops.softmax_bwd(y, dy.contiguous())
        return ops.dense_bwd(x, self.store.w[self.name + '/kernel'], y, dy, self.act, g[self.name + '/kernel'], db, need_dx)


class LayerNormalization:
    """tf.keras.layers.LayerNormalization() between Dense layers (axis -1, epsilon 1e-3; models/Dirichlet_BC_NN_Metalearning.py:73-75)."""

    def __init__(self, store, name, features):
        self.store, self.name = store, name
        store.add(name + '/gamma', (features,), 'ones')
        store.add(name + '/beta', (features,), 'zeros')

    def forward(self, x, training=True):
        y, st = ops.layernorm_fwd(x, self.store.w[self.name + '/gamma'], self.store.w[self.name + '/beta'])
        self.saved = (x, st) if training else None
        return y

    def backward(self, dy, need_dx=True):
        x, st = self.saved
        self.saved = None
        g = self.store.g
        return ops.layernorm_bwd(x, self.store.w[self.name + '/gamma'], st, dy.contiguous(), g[self.name + '/gamma'], g[self.name + '/beta'])


class Scaling:
    """layers/Scaling.py:18-55 with SpatialPyramidPool MAX (layers/SpatialPyramidPool.py:35-66)."""

    def __init__(self, store, ctx, name='scaling', *, stages=2, downsampling_ratio_per_stage=2, spp_levels=((2, 2), 3, 5), filters=None,
                 kernel_size=None, activation='linear', **unused):
        self.store, self.ctx, self.name = store, ctx, name
        self.ratio = int(downsampling_ratio_per_stage)
        self.levels = [[lv, lv] if isinstance(lv, int) else (list(lv) * 2 if len(lv) == 1 else list(lv)) for lv in spp_levels]
        cin = 2
        self.convs = []
        for i in range(stages):
            self.convs.append(ConvUnit(store, ctx, '%s/conv%d' % (name, i), kernel_size, cin, filters, pad='same', activation=activation))
            cin = filters
        nfeat = sum(a * b for a, b in self.levels)
        self.d0 = Dense(store, name + '/dense0', nfeat, 100, 'leaky_relu')
        self.d1 = Dense(store, name + '/dense1', 100, 25, 'leaky_relu')
        self.d2 = Dense(store, name + '/dense2', 25, 1, 'linear')
        self._bins = {}

    def _bin_table(self, H, W, device):
        key = (H, W, str(device))
        if key not in self._bins:
            bins = []
            for ly, lx in self.levels:
                iy, ix = split_indices(H, ly), split_indices(W, lx)
                if (np.diff(iy) <= 0).any() or (np.diff(ix) <= 0).any():
                    raise ValueError('grid too small for the Scaling layer: a %dx%d pyramid level over a %dx%d map has empty bins' % (ly, lx, H, W))
                bins += [[iy[a], iy[a + 1], ix[b], ix[b + 1]] for a in range(ly) for b in range(lx)]
            self._bins[key] = ops.upload(np.array(bins, dtype=np.int32), device)
        return self._bins[key]

    def forward(self, x_to_scale, other, training=True):
        N, H, W, _ = x_to_scale.shape
        cat = ops.empty((N, H, W, 2), x_to_scale.device)
        ops.axpby(1.0, x_to_scale, 0.0, cat[..., 0:1])
        ops.axpby(1.0, other, 0.0, cat[..., 1:2])
        o = cat
        self.pool_in = []
        for c in self.convs:
            o = c.forward(o, training=training)
            self.pool_in.append(o)
            o = ops.pool2d_fwd(o, self.ratio, 'average')
        feats, arg = ops.spp_max_fwd(o, self._bin_table(o.shape[1], o.shape[2], o.device))
        g = self.d2.forward(self.d1.forward(self.d0.forward(feats, training), training), training)
        y = ops.sample_scale_fwd(x_to_scale, g)
        self.saved = (x_to_scale, g, arg, tuple(o.shape)) if training else None
        return y

    def backward(self, dy):
        x, g, arg, oshape = self.saved
        self.saved = None
        dx, dg = ops.sample_scale_bwd(x, g, dy)
        d = self.d0.backward(self.d1.backward(self.d2.backward(dg.view(-1, 1))))
        d = ops.spp_max_bwd(arg, d, oshape)
        for c, pin in zip(reversed(self.convs), reversed(self.pool_in)):
            d = ops.pool2d_bwd(pin, d, self.ratio, 'average')
            d = c.backward(d, inplace=True)
        self.pool_in = None
        return ops.axpby(1.0, d[..., 0:1], 1.0, dx)   # channel 0 of the concat is x_to_scale; `other` (the rhs) needs no gradient


class JacobiIterationLayer:
    """layers/JacobiIterationLayer.py:7-66 for the model's ([3,3],[2,2]) stencil."""

    def __init__(self, n_iterations=5):
        self.n = int(n_iterations)

    def forward(self, guess, rhs, dx2, training=True):
        u = guess
        for _ in range(self.n):
            u = ops.jacobi_sweep(u, rhs, dx2)
        self.dx2 = dx2
        return u

    def backward(self, dout):
        d = dout
        for _ in range(self.n):
            d = ops.jacobi_sweep_bwd(d, self.dx2)
        return d
