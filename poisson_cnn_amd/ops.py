"""Thin Python wrappers over the libpcnn C-ABI (include/pcnn.h).  Tensors are torch CUDA float32 in NHWC with an
arbitrary channel stride (channel slices of wider buffers are accepted).  torch is used for allocation, streams and
nothing else: every arithmetic op below is a HIP kernel reached through ctypes."""
import ctypes
from ctypes import byref, c_float, c_int, c_int64, c_size_t, c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc

PAD_MODES = {'CONSTANT': 0, 'SYMMETRIC': 1, 'REFLECT': 2}
ACTS = {'linear': 0, 'leaky_relu': 1, 'tanh': 2, 'relu': 3}
POOLS = {'average': 0, 'avg': 0, 'max': 1}
RESIZE = {'nearest': 0, 'bilinear': 1, 'bicubic': 2, 'bicubic_legacy': 3}
LEAKY_ALPHA = 0.2   # tf.nn.leaky_relu default
BN_EPS = 1e-3       # tf.keras.layers.BatchNormalization default

_handles = {}
MATH_MODES = {'fp32': 0, 'split_f16': 1}
_math_mode = MATH_MODES[__import__('os').environ.get('PCNN_MATH', 'fp32')]


def set_math_mode(mode):
    """'fp32' (default: exact fp32 MFMA) or 'split_f16' (3 x fp16 split MFMA, fp32 accumulate; include/pcnn.h PCNN_MATH_SPLIT_F16).
    Also selectable with the environment variable PCNN_MATH."""
    global _math_mode
    _math_mode = MATH_MODES[mode]
    for h in _handles.values():
        h.call('pcnn_set_math_mode', c_int(_math_mode))


_spectral_mode = int(__import__('os').environ.get('PCNN_SPECTRAL', '-1'))


def set_spectral_mode(mode):
    """-1 / 'auto' (cost model per layer, default), 0 / 'off' (direct implicit GEMM only), 1 / 'force' (tiled spectral convolution
    whenever the shape allows): include/pcnn.h pcnn_set_spectral_mode.  Environment: PCNN_SPECTRAL."""
    global _spectral_mode
    _spectral_mode = {'auto': -1, 'off': 0, 'force': 1}.get(mode, mode)
    for h in _handles.values():
        h.call('pcnn_set_spectral_mode', c_int(_spectral_mode))


def get_spectral_mode():
    return _spectral_mode


_spectral_tile = int(__import__('os').environ.get('PCNN_SPEC_T', '0'))


def set_spectral_tile(tile):
    """0 (default: per layer - 64 x 64 tiles for 11..15 taps on large images), 32 or 64: include/pcnn.h pcnn_set_spectral_tile.
    Environment: PCNN_SPEC_T."""
    global _spectral_tile
    _spectral_tile = int(tile)
    for h in _handles.values():
        h.call('pcnn_set_spectral_tile', c_int(_spectral_tile))


def get_spectral_tile():
    return _spectral_tile


XFORMS = {'mfma': 0, 'fft': 1}
_spectral_xform = XFORMS[{'m': 'mfma', '0': 'mfma'}.get(__import__('os').environ.get('PCNN_SPEC_XFORM', 'fft')[:1], 'fft')]      # default: the FFT kernels (round 5)


def set_spectral_transform(xform):
    """'mfma' (the DFT as a GEMM on the matrix cores) or 'fft' (in-register FFTs on the vector ALUs): the transform kernels of the spectral route,
    include/pcnn.h pcnn_set_spectral_transform.  Environment: PCNN_SPEC_XFORM."""
    global _spectral_xform
    _spectral_xform = XFORMS[xform]
    for h in _handles.values():
        h.call('pcnn_set_spectral_transform', c_int(_spectral_xform))


def get_spectral_transform():
    return [k for k, v in XFORMS.items() if v == _spectral_xform][0]


def get_math_mode():
    return [k for k, v in MATH_MODES.items() if v == _math_mode][0]


# ---- filter spectra across calls (include/pcnn.h pcnn_set_filter_version).  The layer classes know when their weights change: every change - an optimizer
# step, a load, a torch-side in-place write to the parameter bucket (torch counts those: Tensor._version) - bumps ONE process-wide version; a convolution
# called with `w_version` = that number lets its handle keep the filter's spectrum until the number moves.  Calls without `w_version` (raw ops users,
# tools, tests) run with version 0: nothing cached, any tensor contents are fine.  PCNN_FILTER_CACHE=0 turns the feature off.
_filter_cache_on = __import__('os').environ.get('PCNN_FILTER_CACHE', '1') != '0'
_filter_version = 1
_filter_epoch = 0


def weights_changed():
    """Some layer weights may hold new values: spectra cached under the old version are refreshed on their next use."""
    global _filter_version
    _filter_version += 1


def weights_released():
    """A parameter bucket went away: every handle drops its cached spectra before its next cached call (the refresh reads every known filter)."""
    global _filter_epoch, _filter_version
    _filter_epoch += 1
    _filter_version += 1


def filter_version():
    return _filter_version if _filter_cache_on else 0


def set_filter_cache(on):
    global _filter_cache_on
    _filter_cache_on = bool(on)


def filter_cache_stats():
    """Sum over this process's handles: dict(entries, bytes, hits, fills, refreshes) - pcnn_filter_cache_stats."""
    tot = dict(entries=0, bytes=0, hits=0, fills=0, refreshes=0)
    for h in _handles.values():
        v = [ctypes.c_longlong() for _ in range(5)]
        h.call('pcnn_filter_cache_stats', *[byref(x) for x in v])
        for k, x in zip(tot, v):
            tot[k] += x.value
    return tot


_flip_layers = __import__('weakref').WeakSet()   # layers.Conv objects that keep a flipped filter of their own (they have run a backward pass)
_flips_version = 0


def register_flipped(layer):
    _flip_layers.add(layer)


_flip_table = (None, None, 0, 0)                 # (key, device table, n, total) of pcnn_conv2d_flip_transpose_table


def _flip_layer_set():
    layers = sorted((l for l in _flip_layers if getattr(l, '_wf', None) is not None), key=id)
    return layers, tuple((id(l), l._wf.data_ptr()) for l in layers)


def _build_flip_table(layers, key):
    """Uploads the device table of (filter, flipped filter, shape) entries.  The previous table is NOT freed here by hand: a captured graph that recorded
    a table launch holds a reference to the tensor it recorded (graphs._Captured._keep, flip_table_tensor), so a retired table lives exactly as long
    as a graph can replay it (a replay must never read a recycled allocation as pointer entries)."""
    global _flip_table
    arr = np.zeros(len(layers), dtype=[('w', '<u8'), ('wt', '<u8'), ('kh', '<i4'), ('kw', '<i4'), ('ci', '<i4'), ('co', '<i4'), ('start', '<i8')])
    start = 0
    for i, l in enumerate(layers):
        w = l.store.w[l.name + '/kernel']
        arr[i] = (w.data_ptr(), l._wf.data_ptr(), l.kh, l.kw, l.cin, l.cout, start)
        start += w.numel()
    _flip_table = (key, torch.from_numpy(arr.view(np.uint8)).to(layers[0]._wf.device), len(layers), start)


def prepare_flip_table_for_capture():
    """Called by graphs._Captured right before a capture (outside it): the table is rebuilt for the layer set the warm-up left behind, so that the
    capture records ONE table launch instead of one flip per layer, and returned so that the graph keeps it (and the flipped filters it points
    to) alive for its replays.  Returns a list of tensors to hold."""
    layers, key = _flip_layer_set()
    if not layers:
        return []
    if key != _flip_table[0]:
        _build_flip_table(layers, key)
    return [_flip_table[1]] + [l._wf for l in layers]


def sync_flipped_filters(version):
    """The promise behind a weights version covers EVERY filter pointer a handle has seen, the flipped filters of the backward pass included: the
    first cached call under a new version makes its handle refresh all of them at once.  So before that call every layer's flipped filter is
    re-formed from the current weights - here, in ONE launch (pcnn_conv2d_flip_transpose_table; the table of pointers lives on the device and is
    rebuilt only when a layer joins), on the calling stream (the model's main stream: the first convolution of a step runs before any branch
    stream forks off)."""
    global _flips_version, _flip_table
    if not version or version == _flips_version:
        return
    layers, key = _flip_layer_set()
    if layers:
        if key != _flip_table[0] and torch.cuda.is_current_stream_capturing():
            for l in layers:                                           # the set of layers changed and nothing may be uploaded inside a capture:
                l._reflip(version)                                     # one launch per layer, as before the table existed
            _flips_version = version
            return
        if key != _flip_table[0]:
            _build_flip_table(layers, key)
        _, tab, n, total = _flip_table
        handle().call('pcnn_conv2d_flip_transpose_table', _p(tab), c_int(n), c_int64(total))
        for l in layers:
            l._wf_ver = version
    _flips_version = version


class _FilterVersion:
    """`with _FilterVersion(h, v):` - the handle runs the enclosed call under weights version v and returns to 0 (uncached) afterwards, so a later call
    that says nothing about its filter's contents can never meet a cached spectrum."""
    __slots__ = ('h', 'v')

    def __init__(self, h, v):
        self.h, self.v = h, int(v or 0)

    def __enter__(self):
        if self.v:
            h = self.h
            if getattr(h, '_fepoch', 0) != _filter_epoch:
                if torch.cuda.is_current_stream_capturing():         # clearing synchronises and frees: not inside a capture - this call runs uncached,
                    self.v = 0                                       # the cache is emptied by the first call after the capture
                    return
                h.call('pcnn_filter_cache_clear')
                h._fepoch = _filter_epoch
            h.call('pcnn_set_filter_version', self.v)

    def __exit__(self, *exc):
        if self.v:
            self.h.call('pcnn_set_filter_version', 0)
        return False


def handle():
    """The libpcnn handle of (current device, torch's current stream): one handle per stream (include/pcnn.h), so the weight gradients on
    the side stream have their own filter scratch and spectral workspace and never share library state with the main stream."""
    dev = torch.cuda.current_device()
    st = torch.cuda.current_stream().cuda_stream
    h = _handles.get((dev, st))
    if h is None:
        h = _lib.Handle(dev, st)
        h.call('pcnn_set_math_mode', c_int(_math_mode))
        h.call('pcnn_set_spectral_mode', c_int(_spectral_mode))
        h.call('pcnn_set_spectral_tile', c_int(_spectral_tile))
        h.call('pcnn_set_spectral_transform', c_int(_spectral_xform))
        _handles[(dev, st)] = h
    return h


def release_stream_handle(stream_ptr, device=None):
    """Drops the cached handle of (device, stream): pcnn_destroy frees the handle's scratch, its spectral workspace and every buffer parked by
    pcnn_set_workspace_retain.  For streams that go away (graphs._Captured.close); the next use of the stream simply creates a fresh handle."""
    dev = torch.cuda.current_device() if device is None else device
    h = _handles.pop((dev, int(stream_ptr)), None)
    if h is not None:
        try:
            h.call('pcnn_set_workspace_retain', c_int(0))
        finally:
            del h                                                      # Handle.__del__ -> pcnn_destroy


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _chk(t, name='tensor'):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError('%s must be a CUDA float32 tensor' % name)
    if t.dim() > 0 and t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError('%s must have unit channel stride (NHWC)' % name)
    return t


def _ld(t):
    """channel stride (floats between consecutive pixels) of an NHWC tensor; checks pixel-linear strides."""
    _chk(t)
    N, H, W, C = t.shape
    if W > 1:
        ld = t.stride(2)
    elif H > 1:
        ld = t.stride(1)
    elif N > 1:
        ld = t.stride(0)
    else:
        ld = max(C, 1)
    if (W > 1 and t.stride(2) != ld) or (H > 1 and t.stride(1) != W * ld) or (N > 1 and t.stride(0) != H * W * ld) or ld < C:
        raise ValueError('tensor is not pixel-linear NHWC (strides %s for shape %s)' % (t.stride(), tuple(t.shape)))
    return ld


def empty(shape, device='cuda'):
    return torch.empty(shape, dtype=torch.float32, device=device)


def zeros(shape, device='cuda'):
    return torch.zeros(shape, dtype=torch.float32, device=device)


class Workspace:
    """Grow-on-demand scratch buffer (one per model)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device='cuda'):
        n = (int(nbytes) + 3) // 4 + 16
        if self.buf is None or self.buf.numel() < n:
            self.buf = None
            self.buf = torch.empty(n, dtype=torch.float32, device=device)
        return self.buf


_default_ws = Workspace()
_post_fusion = __import__('os').environ.get('PCNN_POST_FUSION', '1') != '0'      # developer switch: 0 = every activation backward as its own pass


def set_post_fusion(on):
    global _post_fusion
    _post_fusion = bool(on)


_stage_fusion = __import__('os').environ.get('PCNN_STAGE_FUSION', '1') != '0'    # developer switch: 0 = a narrow resnet stage as three launches


def set_stage_fusion(on):
    global _stage_fusion
    _stage_fusion = bool(on)


class KernelTimer:
    """Live per-kernel timing for bench.py: HIP events (torch.cuda.Event on the launch stream) around every launch of the
    MFMA kernels, with the algorithmic FLOP count of each launch."""

    def __init__(self):
        self.records = []   # (kind, flops, start_event, end_event, algorithmic bytes)

    def launch(self, kind, flops, fn, nbytes=0.0):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.append((kind, flops, s, e, nbytes))

    def total_bytes(self, kind):
        return sum(r[4] for r in self.records if r[0] == kind)

    def select(self, pred):
        """(flops, bytes, seconds, launches) over the records for which pred(kind, flops, bytes) holds."""
        torch.cuda.synchronize()
        sel = [r for r in self.records if pred(r[0], r[1], r[4])]
        return sum(r[1] for r in sel), sum(r[4] for r in sel), sum(r[2].elapsed_time(r[3]) for r in sel) * 1e-3, len(sel)

    def totals(self, kind):
        torch.cuda.synchronize()
        sel = [r for r in self.records if r[0] == kind]
        return sum(r[1] for r in sel), sum(r[2].elapsed_time(r[3]) for r in sel) * 1e-3, len(sel)


_timer = None


def set_kernel_timer(t):
    global _timer
    _timer = t


def _launch(kind, flops, fn, nbytes=0.0):
    if _timer is None:
        fn()
    else:
        _timer.launch(kind, flops, fn, nbytes)


# ----------------------------------------------------------------------------- convolution
def conv_desc(x_shape, ldx, w_shape, out_hw, ldy, pad_top, pad_left, pad_mode='CONSTANT', pad_value=0.0, act='linear', ld_res=0, ld_act=0):
    N, H, W, Cin = x_shape
    kh, kw, ci, Cout = w_shape
    assert ci == Cin, (ci, Cin)
    return ConvDesc(N, H, W, Cin, ldx, out_hw[0], out_hw[1], Cout, ldy, kh, kw, pad_top, pad_left, PAD_MODES[pad_mode.upper()],
                    float(pad_value), ACTS[act], LEAKY_ALPHA, ld_res, ld_act)


def conv2d_fwd(x, w, bias=None, *, pad_top, pad_left, out_hw=None, pad_mode='CONSTANT', pad_value=0.0, act='linear',
               bn_scale=None, bn_shift=None, residual=None, out=None, act_out=None, y_absmax=None, w_version=0):
    """w_version: 0 (default) - nothing is known about w's contents; else the caller's weights version (filter_version()): the spectral route may
    reuse the spectrum it computed for this very tensor under the same version."""
    N, H, W, Cin = x.shape
    kh, kw, ci, Cout = w.shape
    assert w.is_contiguous()
    Ho, Wo = out_hw if out_hw is not None else (H, W)
    if out is None:
        out = empty((N, Ho, Wo, Cout), x.device)
    d = conv_desc(x.shape, _ld(x), w.shape, (Ho, Wo), _ld(out), pad_top, pad_left, pad_mode, pad_value, act,
                  _ld(residual) if residual is not None else 0, _ld(act_out) if act_out is not None else 0)
    def run():
        h = handle()
        with _FilterVersion(h, w_version):
            h.call('pcnn_conv2d_fwd_absmax', byref(d), _p(x), _p(w), _p(bias), _p(bn_scale), _p(bn_shift), _p(residual), _p(out), _p(act_out), _p(y_absmax))
    _launch('conv_fwd', 2.0 * N * Ho * Wo * kh * kw * Cin * Cout, run,
            4.0 * (N * H * W * Cin + N * Ho * Wo * Cout * (1 + (residual is not None) + (act_out is not None)) + kh * kw * Cin * Cout))
    return out


def conv2d_dgrad_post(dz, wf, *, pad_top, pad_left, out_hw, residual, post):
    """The data gradient of a NARROW layer (vector-ALU route) with the producer's activation backward in its epilogue (pcnn_conv2d_dgrad_post): returns dx
    (already multiplied by act') and sets post.applied / post.raw - or returns None when the library declines (not the narrow route, odd channel counts,
    unaligned views, an inference-mode BatchNormalization in the offer): the caller then runs conv2d_fwd and the separate pass."""
    if post is None or post.bn_scale is not None or not _post_fusion:
        return None
    N, H, W, Cin = dz.shape
    kh, kw, _, Cout = wf.shape
    Ho, Wo = out_hw
    out = empty((N, Ho, Wo, Cout), dz.device)
    dg = conv_desc(dz.shape, _ld(dz), wf.shape, (Ho, Wo), _ld(out), pad_top, pad_left, 'CONSTANT', 0.0, 'linear', _ld(residual) if residual is not None else 0, 0)
    raw = empty((N, Ho, Wo, Cout), dz.device) if post.want_raw else None
    pd = PostDesc(post.a.data_ptr(), _ld(post.a), ACTS[post.act], LEAKY_ALPHA, post.dbias.data_ptr() if post.dbias is not None else None,
                  raw.data_ptr() if raw is not None else None, Cout)
    h = handle()
    if tuple(post.a.shape) != (N, Ho, Wo, Cout) or not h.lib.pcnn_conv2d_dgrad_post_eligible(h._h, byref(dg), _p(dz), _p(residual), _p(out), byref(pd)):
        return None
    _launch('conv_fwd_post', 2.0 * N * Ho * Wo * kh * kw * Cin * Cout,                  # (its own kind: a convolution launch that also does the producer's epilogue pass)
            lambda: h.call('pcnn_conv2d_dgrad_post', byref(dg), _p(dz), _p(wf), _p(residual), _p(out), byref(pd)),
            4.0 * (N * H * W * Cin + N * Ho * Wo * Cout * (1 + (residual is not None)) + kh * kw * Cin * Cout))      # the convolution's own bytes, as conv2d_fwd counts them
    post.applied, post.raw = True, raw
    return out


def flip_transpose_weights(w, out=None):
    kh, kw, ci, co = w.shape
    wt = out if out is not None else empty((kh, kw, co, ci), w.device)
    handle().call('pcnn_conv2d_flip_transpose_weights', _p(w), _p(wt), c_int(kh), c_int(kw), c_int(ci), c_int(co))
    return wt


def conv2d_wgrad(x, dz, w_shape, *, pad_top, pad_left, pad_mode='CONSTANT', pad_value=0.0, out=None, ws=None, x_absmax=None, dz_absmax=None):
    """dw (kh,kw,Cin,Cout) of the forward conv that maps x -> dz's shape."""
    N, Ho, Wo, Cout = dz.shape
    d = conv_desc(x.shape, _ld(x), w_shape, (Ho, Wo), _ld(dz), pad_top, pad_left, pad_mode, pad_value)
    lib = _lib.load()
    nbytes = lib.pcnn_conv2d_wgrad_workspace(byref(d))
    wsb = (ws or _default_ws).get(nbytes, x.device)
    dw = out if out is not None else empty(tuple(w_shape), x.device)
    assert dw.is_contiguous()
    _launch('conv_wgrad', 2.0 * N * Ho * Wo * w_shape[0] * w_shape[1] * w_shape[2] * w_shape[3],
            lambda: handle().call('pcnn_conv2d_wgrad_hint', byref(d), _p(x), _p(dz), _p(dw), _p(wsb), c_size_t(wsb.numel() * 4), _p(x_absmax), _p(dz_absmax)),
            4.0 * (x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] + N * Ho * Wo * Cout + w_shape[0] * w_shape[1] * w_shape[2] * w_shape[3]))
    return dw


class PostDesc(ctypes.Structure):
    """include/pcnn.h pcnn_post_desc"""
    _fields_ = [('act_out', c_void_p), ('ld_act_out', c_int), ('act', c_int), ('act_alpha', c_float), ('dbias', c_void_p), ('raw_out', c_void_p), ('ld_raw', c_int)]


class Post:
    """The activation backward of the layer that produced a convolution's input, offered to that convolution's data-gradient kernel for fusion
    (pcnn_conv2d_bwd_spectral_post): a = the producer's saved activation output, act its activation, dbias its bias gradient (or None),
    want_raw: the un-multiplied gradient is needed as well (a skip connection branches off here).  After the call `applied` says whether the
    kernel took it (then `raw` holds the un-multiplied gradient if it was asked for)."""

    def __init__(self, a, act, dbias=None, want_raw=False, bn_scale=None, s_dy_a=None, s_dy=None):
        self.a, self.act, self.dbias, self.want_raw = a, act, dbias, want_raw
        self.bn_scale, self.s_dy_a, self.s_dy = bn_scale, s_dy_a, s_dy   # an inference-mode BatchNormalization behind the activation (fold route only: pad_fold_bwd_post)
        self.applied, self.raw = False, None


def conv2d_bwd_fused(x, dz, w_shape, wf, *, pad_top, pad_left, pad_mode='CONSTANT', pad_value=0.0, dw, residual=None, post=None, w_version=0):
    """Both gradients of the fused pad+conv in one call when the layer takes the spectral route (pcnn_conv2d_bwd_spectral: dz's spectrum is
    computed once for the data gradient and the weight gradient).  wf: the flipped / transposed filter (kh,kw,Cout,Cin).  Fills dw and returns
    the data-gradient convolution's output - dx for CONSTANT padding (+ `residual` if given), the gradient on the padded domain otherwise
    (fold it with pad_fold_bwd) - or None when the layer is not eligible (the caller then uses conv2d_wgrad / conv2d_fwd)."""
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w_shape
    mode = pad_mode.upper()
    d = conv_desc(x.shape, _ld(x), w_shape, (dz.shape[1], dz.shape[2]), _ld(dz), pad_top, pad_left, mode, pad_value)
    if mode == 'CONSTANT':
        out_hw, dpt, dpl = (H, W), kh - 1 - pad_top, kw - 1 - pad_left
    else:
        out_hw, dpt, dpl = (H + kh - 1, W + kw - 1), kh - 1, kw - 1
        residual = None
    ldo = Cin
    dg = conv_desc(dz.shape, _ld(dz), (kh, kw, Cout, Cin), out_hw, ldo, dpt, dpl, 'CONSTANT', 0.0, 'linear', _ld(residual) if residual is not None else 0, 0)
    h = handle()
    if not h.lib.pcnn_conv2d_bwd_spectral_eligible(h._h, byref(d), byref(dg)):
        return None
    out = empty((N, out_hw[0], out_hw[1], Cin), x.device)
    flops = 2.0 * N * dz.shape[1] * dz.shape[2] * kh * kw * Cin * Cout
    nbytes = 4.0 * (2 * N * H * W * Cin + N * dz.shape[1] * dz.shape[2] * Cout + 2 * kh * kw * Cin * Cout)
    if post is not None and post.bn_scale is None and mode == 'CONSTANT' and _post_fusion and h.lib.pcnn_conv2d_bwd_spectral_post_eligible(h._h, byref(d), byref(dg)):
        post.raw = empty((N, H, W, Cin), x.device) if post.want_raw else None
        pd = PostDesc(post.a.data_ptr(), _ld(post.a), ACTS[post.act], LEAKY_ALPHA, post.dbias.data_ptr() if post.dbias is not None else None,
                      post.raw.data_ptr() if post.raw is not None else None, Cin)
        def run_post():
            with _FilterVersion(h, w_version):
                h.call('pcnn_conv2d_bwd_spectral_post', byref(d), byref(dg), _p(x), _p(dz), _p(wf), _p(residual), _p(out), _p(dw), byref(pd))
        _launch('conv_bwd_fused', 2.0 * flops, run_post, nbytes)
        post.applied = True
        return out
    def run():
        with _FilterVersion(h, w_version):
            h.call('pcnn_conv2d_bwd_spectral', byref(d), byref(dg), _p(x), _p(dz), _p(wf), _p(residual), _p(out), _p(dw))
    _launch('conv_bwd_fused', 2.0 * flops, run, nbytes)
    return out


def resnet3_eligible(C, act):
    """A narrow resnet stage (3 x 3, C -> C three times, zero padding) that pcnn_resnet3_fwd runs as one launch."""
    h = handle()
    return _stage_fusion and bool(h.lib.pcnn_resnet3_fwd_eligible(h._h, c_int(C), c_int(ACTS[act])))


def resnet3_fwd(x, w0, b0, w1, b1, w2, b2, *, act, training, out=None):
    """blocks/resnet.py:29-39 as ONE launch (pcnn_resnet3_fwd).  Returns (y, o0, a1, o1): the three intermediates the unfused chain leaves for
    the backward pass (None when training is False - they then never reach HBM)."""
    N, H, W, C = x.shape
    assert x.is_contiguous() and w0.shape == (3, 3, C, C) and w1.shape == (3, 3, C, C) and w2.shape == (3, 3, C, C)
    if out is None:
        out = empty((N, H, W, C), x.device)
    assert out.is_contiguous() and out.shape == x.shape
    o0, a1, o1 = (empty((N, H, W, C), x.device) for _ in range(3)) if training else (None, None, None)
    T = 4.0 * N * H * W * C
    # algorithmic bytes of the UNFUSED layers (what the three conv2d_fwd launches count: conv1 reads the skip input and, in training, writes a1 too)
    nbytes = (8.0 if training else 7.0) * T + 3 * 4.0 * 9 * C * C
    _launch('conv_stage', 3 * 2.0 * N * H * W * 9 * C * C,
            lambda: handle().call('pcnn_resnet3_fwd', c_int(N), c_int(H), c_int(W), c_int(C), c_int(ACTS[act]), c_float(LEAKY_ALPHA), _p(x), _p(w0), _p(b0),
                                  _p(w1), _p(b1), _p(w2), _p(b2), _p(o0), _p(a1), _p(o1), _p(out)), nbytes)
    return out, o0, a1, o1


def epilogue_bwd(dy, a, *, act='linear', bn_scale=None, dz=None, dbias=None, s_dy_a=None, s_dy=None, ws=None, dz_absmax=None):
    """dz_absmax: optional 1-element device tensor that receives max|dz| (a free by-product; conv2d_wgrad takes it as a hint)."""
    N, H, W, C = dy.shape
    lib = _lib.load()
    wsb = (ws or _default_ws).get(lib.pcnn_colsum_workspace(c_int(C)), dy.device)
    handle().call('pcnn_conv2d_epilogue_bwd_absmax', c_int64(N * H * W), c_int(C), _p(dy), c_int(_ld(dy)), _p(a), c_int(_ld(a) if a is not None else 0),
                  _p(bn_scale), c_int(ACTS[act]), c_float(LEAKY_ALPHA), _p(dz), c_int(_ld(dz) if dz is not None else 0),
                  _p(dbias), _p(s_dy_a), _p(s_dy), _p(dz_absmax), _p(wsb), c_size_t(wsb.numel() * 4))
    return dz


def pad_fold_bwd(gp, out_hw, pads, pad_mode, out=None, accumulate=False):
    N, Hp, Wp, C = gp.shape
    H, W = out_hw
    (pt, pb), (pl, pr) = pads
    assert Hp == H + pt + pb and Wp == W + pl + pr
    gx = out if out is not None else empty((N, H, W, C), gp.device)
    handle().call('pcnn_pad_fold_bwd', c_int(N), c_int(H), c_int(W), c_int(C), c_int(pt), c_int(pb), c_int(pl), c_int(pr),
                  c_int(PAD_MODES[pad_mode.upper()]), _p(gp), c_int(_ld(gp)), _p(gx), c_int(_ld(gx)), c_int(1 if accumulate else 0))
    return gx


_fold_post = __import__('os').environ.get('PCNN_FOLD_POST', '1') != '0'          # developer switch (A/B timing): 0 = fold and activation backward as two passes


def pad_fold_bwd_post(gp, out_hw, pads, pad_mode, post, add_to=None, ws=None):
    """pad_fold_bwd AND the producer's activation backward in one pass (pcnn_pad_fold_bwd_post): returns dz = (fold(gp) [+ add_to]) * act'(post.a), sets
    post.applied / post.raw (the un-multiplied gradient, if post.want_raw) and writes post.dbias - or returns None (nothing done) where the shape is not
    eligible or the fusion is switched off (PCNN_POST_FUSION=0): the caller then folds and lets the producer run its own epilogue pass."""
    if not _post_fusion or not _fold_post:
        return None
    N, Hp, Wp, C = gp.shape
    H, W = out_hw
    (pt, pb), (pl, pr) = pads
    assert Hp == H + pt + pb and Wp == W + pl + pr
    lib = _lib.load()
    dz = empty((N, H, W, C), gp.device)
    raw = empty((N, H, W, C), gp.device) if post.want_raw else None
    pd = PostDesc(post.a.data_ptr(), _ld(post.a), ACTS[post.act], LEAKY_ALPHA, post.dbias.data_ptr() if post.dbias is not None else None,
                  raw.data_ptr() if raw is not None else None, C)
    if add_to is not None and not (add_to.dim() == 4 and tuple(add_to.shape) == (N, H, W, C)):
        return None
    ld_add = _ld(add_to) if add_to is not None else 0
    if not lib.pcnn_pad_fold_bwd_post_eligible(c_int(C), c_int(_ld(gp)), c_int(ld_add), _p(gp), _p(add_to), byref(pd), c_int(_ld(dz)), _p(dz)):
        return None
    wsb = (ws or _default_ws).get(lib.pcnn_colsum_workspace(c_int(C)), gp.device)
    handle().call('pcnn_pad_fold_bwd_post', c_int(N), c_int(H), c_int(W), c_int(C), c_int(pt), c_int(pb), c_int(pl), c_int(pr), c_int(PAD_MODES[pad_mode.upper()]),
                  _p(gp), c_int(_ld(gp)), _p(add_to), c_int(ld_add), byref(pd), _p(post.bn_scale), _p(post.s_dy_a), _p(post.s_dy), _p(dz), c_int(_ld(dz)),
                  _p(wsb), c_size_t(wsb.numel() * 4))
    post.raw, post.applied = raw, True
    return dz


def bn_fold(gamma, beta, mean, var, scale, shift, eps=BN_EPS):
    handle().call('pcnn_bn_fold', c_int(gamma.numel()), _p(gamma), _p(beta), _p(mean), _p(var), c_float(eps), _p(scale), _p(shift))


def bn_fold_bwd(s1, s2, mean, var, dgamma, dbeta, eps=BN_EPS):
    handle().call('pcnn_bn_fold_bwd', c_int(s1.numel()), _p(s1), _p(s2), _p(mean), _p(var), c_float(eps), _p(dgamma), _p(dbeta))


def channel_affine(x, scale, shift, residual=None, out=None):
    N, H, W, C = x.shape
    y = out if out is not None else empty((N, H, W, C), x.device)
    handle().call('pcnn_channel_affine', c_int64(N * H * W), c_int(C), _p(x), c_int(_ld(x)), _p(scale), _p(shift), _p(residual),
                  c_int(_ld(residual) if residual is not None else 0), _p(y), c_int(_ld(y)))
    return y


def bn_train_forward(a, gamma, beta, moving_mean, moving_var, residual=None, out=None, momentum=0.99, eps=BN_EPS, ws=None):
    """Batch statistics of `a` (N,H,W,C), moving-average update, y = BN(a) (+ residual).  Returns (y, (mean, inv_std, scale))."""
    N, H, W, C = a.shape
    st = empty((6, C), a.device)
    sum_a2, sum_a, mean, inv_std, scale, shift = st[0], st[1], st[2], st[3], st[4], st[5]
    epilogue_bwd(a, a, act='linear', s_dy_a=sum_a2, s_dy=sum_a, ws=ws)      # s_dy_a = sum a*a, s_dy = sum a
    handle().call('pcnn_bn_train_finalize', c_int(C), c_int64(N * H * W), _p(sum_a), _p(sum_a2), _p(gamma), _p(beta), c_float(eps), c_float(momentum),
                  _p(moving_mean), _p(moving_var), _p(mean), _p(inv_std), _p(scale), _p(shift))
    y = channel_affine(a, scale, shift, residual=residual, out=out)
    return y, (mean, inv_std, scale)


def bn_train_backward(dy, a, stats, dgamma, dbeta, ws=None):
    """da for y = BN_train(a); writes dgamma / dbeta."""
    N, H, W, C = dy.shape
    mean, inv_std, scale = stats
    tmp = empty((4, C), dy.device)
    epilogue_bwd(dy, a, act='linear', s_dy_a=tmp[0], s_dy=tmp[1], ws=ws)
    da = empty((N, H, W, C), dy.device)
    handle().call('pcnn_bn_train_bwd', c_int64(N * H * W), c_int(C), _p(dy), c_int(_ld(dy)), _p(a), c_int(_ld(a)), _p(scale), _p(mean), _p(inv_std),
                  _p(tmp[0]), _p(tmp[1]), _p(dgamma), _p(dbeta), _p(tmp[2:]), _p(da), c_int(C))
    return da


# ----------------------------------------------------------------------------- pooling / deconv / resize
def pool_out(n, f):
    return -(-n // f)


def pool2d_fwd(x, f, kind='average', out=None):
    N, H, W, C = x.shape
    y = out if out is not None else empty((N, pool_out(H, f), pool_out(W, f), C), x.device)
    handle().call('pcnn_pool2d_fwd', c_int(POOLS[kind.lower()]), c_int(N), c_int(H), c_int(W), c_int(C), c_int(f), _p(x), c_int(_ld(x)), _p(y), c_int(_ld(y)))
    return y


def pool2d_bwd(x, dy, f, kind='average', dx=None, accumulate=False):
    N, H, W, C = x.shape
    if dx is None:
        dx = empty((N, H, W, C), x.device)
        accumulate = False
    handle().call('pcnn_pool2d_bwd', c_int(POOLS[kind.lower()]), c_int(N), c_int(H), c_int(W), c_int(C), c_int(f), _p(x), c_int(_ld(x)),
                  c_void_p(0), c_int(0), _p(dy), c_int(_ld(dy)), _p(dx), c_int(_ld(dx)), c_int(1 if accumulate else 0))
    return dx


def _deconv_same_pads(k, hc, wc, H, W, f):
    """tf.nn.conv2d_transpose(padding='SAME') is the adjoint of the SAME forward convolution: pad_before = max((in - 1) f + k - out, 0) // 2."""
    kh, kw = k.shape[0], k.shape[1]
    if hc != -(-H // f) or wc != -(-W // f):
        raise ValueError('SAME transposed convolution needs input = ceil(output / stride): got %s for output %s at stride %d' % ((hc, wc), (H, W), f))
    return max((hc - 1) * f + kh - H, 0) // 2, max((wc - 1) * f + kw - W, 0) // 2


def _deconv_general_fwd(x, k, bias, out_hw, f, alpha, beta, out):
    """kernel_size != upsample_ratio (layers/deconvupscale.py:100-106 allows it; no shipped config uses it): zero insertion (the adjoint of
    the strided sub-sampling) followed by the stride-1 correlation with the flipped / transposed kernel - the fused pad+conv kernels."""
    N, hc, wc, Cin = x.shape
    kh, kw, Cout, _ = k.shape
    H, W = out_hw
    pt, pl = _deconv_same_pads(k, hc, wc, H, W, f)
    x_up = subsample_bwd(x, ((hc - 1) * f + 1, (wc - 1) * f + 1), f)
    y = conv2d_fwd(x_up, flip_transpose_weights(k.contiguous()), bias, pad_top=kh - 1 - pt, pad_left=kw - 1 - pl, out_hw=(H, W))
    if out is None:
        return y if alpha == 1.0 else axpby(alpha, y, 0.0, y)
    return axpby(alpha, y, beta, out)


def deconv_fwd(x, k, bias, out_hw, f, *, alpha=1.0, beta=0.0, out=None):
    N, hc, wc, Cin = x.shape
    Cout = k.shape[2]
    H, W = out_hw
    if k.shape[0] != f or k.shape[1] != f:
        return _deconv_general_fwd(x, k, bias, out_hw, f, alpha, beta, out)
    y = out if out is not None else empty((N, H, W, Cout), x.device)
    # algorithmic bytes: x read once, y written once (+ read once when the branch merge accumulates in place, beta != 0), filter
    _launch('deconv_fwd', 2.0 * N * H * W * Cin * Cout,
            lambda: handle().call('pcnn_deconv_fwd', c_int(N), c_int(hc), c_int(wc), c_int(Cin), c_int(H), c_int(W), c_int(Cout), c_int(f), _p(x), c_int(_ld(x)),
                                  _p(k), _p(bias), c_float(alpha), c_float(beta), _p(y), c_int(_ld(y))),
            4.0 * (N * hc * wc * Cin + N * H * W * Cout * (2 if beta != 0.0 else 1) + f * f * Cin * Cout))
    return y


def deconv_bwd_data(dy, k, coarse_hw, f, *, alpha=1.0, out=None):
    N, H, W, Cout = dy.shape
    Cin = k.shape[3]
    hc, wc = coarse_hw
    if k.shape[0] != f or k.shape[1] != f:               # the SAME strided convolution of dy with k: stride-1 launch + sub-sampling
        pt, pl = _deconv_same_pads(k, hc, wc, H, W, f)
        full = conv2d_fwd(dy, k.contiguous(), None, pad_top=pt, pad_left=pl, out_hw=((hc - 1) * f + 1, (wc - 1) * f + 1))
        dx_ = subsample(full, f)
        if alpha != 1.0:
            axpby(alpha, dx_, 0.0, dx_)
        if out is None:
            return dx_
        return axpby(1.0, dx_, 0.0, out)
    dx = out if out is not None else empty((N, hc, wc, Cin), dy.device)
    handle().call('pcnn_deconv_bwd_data', c_int(N), c_int(hc), c_int(wc), c_int(Cin), c_int(H), c_int(W), c_int(Cout), c_int(f), _p(dy), c_int(_ld(dy)),
                  _p(k), c_float(alpha), _p(dx), c_int(_ld(dx)))
    return dx


def deconv_bwd_filter(x, dy, f, *, alpha=1.0, dk=None, dbias=None, ws=None, kernel_size=None):
    N, hc, wc, Cin = x.shape
    _, H, W, Cout = dy.shape
    if kernel_size is not None and tuple(kernel_size) != (f, f):   # weight gradient of the zero-insertion + correlation form, flipped back
        kh, kw = kernel_size
        pt, pl = max((hc - 1) * f + kh - H, 0) // 2, max((wc - 1) * f + kw - W, 0) // 2
        x_up = subsample_bwd(x, ((hc - 1) * f + 1, (wc - 1) * f + 1), f)
        dwf = conv2d_wgrad(x_up, dy, (kh, kw, Cin, Cout), pad_top=kh - 1 - pt, pad_left=kw - 1 - pl, ws=ws)
        g = flip_transpose_weights(dwf)                              # (kh, kw, Cout, Cin)
        if dk is None:
            dk = g if alpha == 1.0 else axpby_flat(alpha, g, 0.0, g)
        else:
            axpby_flat(alpha, g, 0.0, dk)
        if dbias is not None:
            epilogue_bwd(dy, None, dbias=dbias, ws=ws)
            if alpha != 1.0:
                axpby_flat(alpha, dbias, 0.0, dbias)
        return dk
    lib = _lib.load()
    nbytes = lib.pcnn_deconv_wgrad_workspace(c_int(N), c_int(hc), c_int(wc), c_int(Cin), c_int(Cout), c_int(f))
    wsb = (ws or _default_ws).get(nbytes, x.device)
    if dk is None:
        dk = empty((f, f, Cout, Cin), x.device)
    handle().call('pcnn_deconv_bwd_filter', c_int(N), c_int(hc), c_int(wc), c_int(Cin), c_int(H), c_int(W), c_int(Cout), c_int(f), _p(x), c_int(_ld(x)),
                  _p(dy), c_int(_ld(dy)), c_float(alpha), _p(dk), _p(dbias), _p(wsb), c_size_t(wsb.numel() * 4))
    return dk


def upload(a, device, dtype=None):
    """numpy array -> device tensor through pinned staging and an asynchronous copy on the current stream.  torch.tensor(a, device=...) copies from
    pageable memory and blocks the host until everything queued on the stream before it has run - once per small table of a NEW grid shape (resize
    tables, pyramid bins, quadrature vectors: ~16 per shape), i.e. every step of the shipped training workload (profiles/r06_train_shipped.txt)."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    dev = torch.device(device)
    if dev.type != 'cuda':
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


_table_cache = {}


def resize_tables(method, n_in, n_out, device):
    key = (method, n_in, n_out, str(device))
    t = _table_cache.get(key)
    if t is None:
        idx = np.zeros((n_out, 4), dtype=np.int32)
        wt = np.zeros((n_out, 4), dtype=np.float32)
        rc = _lib.load().pcnn_resize_tables(c_int(RESIZE[method.lower()]), c_int(n_in), c_int(n_out),
                                           idx.ctypes.data_as(c_void_p), wt.ctypes.data_as(c_void_p))
        if rc != 0:
            raise RuntimeError('pcnn_resize_tables failed (%d)' % rc)
        t = (upload(idx, device), upload(wt, device))
        _table_cache[key] = t
    return t


def resize_fwd(x, out_hw, method, *, alpha=1.0, beta=0.0, out=None):
    N, hc, wc, C = x.shape
    Ho, Wo = out_hw
    iy, wy = resize_tables(method, hc, Ho, x.device)
    ix, wx = resize_tables(method, wc, Wo, x.device)
    y = out if out is not None else empty((N, Ho, Wo, C), x.device)
    _launch('resize_fwd', 0.0,
            lambda: handle().call('pcnn_resize_fwd', c_int(N), c_int(hc), c_int(wc), c_int(C), c_int(Ho), c_int(Wo), _p(x), c_int(_ld(x)), _p(iy), _p(wy), _p(ix), _p(wx),
                                  c_float(alpha), c_float(beta), _p(y), c_int(_ld(y))),
            4.0 * (N * hc * wc * C + N * Ho * Wo * C * (2 if beta != 0.0 else 1)))
    return y


class ResizeSrc(ctypes.Structure):
    """include/pcnn.h pcnn_resize_src"""
    _fields_ = [('x', c_void_p), ('hc', c_int), ('wc', c_int), ('ldx', c_int), ('idx_y', c_void_p), ('wt_y', c_void_p), ('idx_x', c_void_p), ('wt_x', c_void_p)]


_resize_multi = __import__('os').environ.get('PCNN_RESIZE_MULTI', '1') != '0'      # developer switch (A/B): 0 = one resize_fwd per branch


def resize_fwd_multi(xs, out_hw, methods, *, alpha, beta, out):
    """out = beta out + alpha (resize(xs[0]) + resize(xs[1]) [+ resize(xs[2])]) with ONE read-modify-write pass over `out` (pcnn_resize_fwd_multi): bit-identical to
    len(xs) consecutive resize_fwd calls with beta, 1, 1.  Returns out, or None when the library declines (channel count, alignment, more than three sources)."""
    if not _resize_multi or not 2 <= len(xs) <= 3:
        return None
    N, _, _, C = xs[0].shape
    Ho, Wo = out_hw
    arr = (ResizeSrc * len(xs))()
    keep = []
    nbytes = 4.0 * N * Ho * Wo * C * (2 if beta != 0.0 else 1)
    for k, (x, m) in enumerate(zip(xs, methods)):
        if x.shape[0] != N or x.shape[3] != C:
            return None
        iy, wy = resize_tables(m, x.shape[1], Ho, x.device)
        ix, wx = resize_tables(m, x.shape[2], Wo, x.device)
        keep += [iy, wy, ix, wx]
        arr[k] = ResizeSrc(x.data_ptr(), x.shape[1], x.shape[2], _ld(x), iy.data_ptr(), wy.data_ptr(), ix.data_ptr(), wx.data_ptr())
        nbytes += 4.0 * x.numel()
    lib = _lib.load()
    if not lib.pcnn_resize_fwd_multi_eligible(c_int(N), c_int(C), c_int(Ho), c_int(Wo), c_int(len(xs)), arr, _p(out), c_int(_ld(out))):
        return None
    _launch('resize_fwd', 0.0, lambda: handle().call('pcnn_resize_fwd_multi', c_int(N), c_int(C), c_int(Ho), c_int(Wo), c_int(len(xs)), arr, c_float(alpha), c_float(beta),
                                                     _p(out), c_int(_ld(out))), nbytes)
    return out


def resize_bwd(dy, coarse_hw, method, *, alpha=1.0, out=None):
    N, Ho, Wo, C = dy.shape
    hc, wc = coarse_hw
    iy, wy = resize_tables(method, hc, Ho, dy.device)
    ix, wx = resize_tables(method, wc, Wo, dy.device)
    tmp = empty((N, hc, Wo, C), dy.device)
    dx = out if out is not None else empty((N, hc, wc, C), dy.device)
    handle().call('pcnn_resize_bwd', c_int(N), c_int(hc), c_int(wc), c_int(C), c_int(Ho), c_int(Wo), _p(dy), c_int(_ld(dy)), _p(iy), _p(wy), _p(ix), _p(wx),
                  c_float(alpha), _p(tmp), _p(dx), c_int(_ld(dx)))
    return dx


# ----------------------------------------------------------------------------- dense / spp
def dense_fwd(x, w, b, act='linear', out=None):
    N, In = x.shape
    Out = w.shape[1]
    y = out if out is not None else empty((N, Out), x.device)
    handle().call('pcnn_dense_fwd', c_int(N), c_int(In), c_int(Out), _p(x), _p(w), _p(b), c_int(ACTS[act]), c_float(LEAKY_ALPHA), _p(y))
    return y


def dense_bwd(x, w, y, dy, act, dw, db, need_dx=True):
    """dw, db are ACCUMULATED into."""
    N, In = x.shape
    Out = w.shape[1]
    dx = empty((N, In), x.device) if need_dx else None
    handle().call('pcnn_dense_bwd', c_int(N), c_int(In), c_int(Out), _p(x), _p(w), _p(y), _p(dy), c_int(ACTS[act]), c_float(LEAKY_ALPHA), _p(dx), _p(dw), _p(db))
    return dx


def layernorm_fwd(x, gamma, beta, eps=1e-3):
    """tf.keras.layers.LayerNormalization() on (N, F): returns (y, (mean, rstd))."""
    N, F = x.shape
    y = empty((N, F), x.device)
    st = empty((2, N), x.device)
    handle().call('pcnn_layernorm_fwd', c_int(N), c_int(F), _p(x), _p(gamma), _p(beta), c_float(eps), _p(y), _p(st[0]), _p(st[1]))
    return y, st


def layernorm_bwd(x, gamma, stats, dy, dgamma, dbeta):
    N, F = x.shape
    dx = empty((N, F), x.device)
    handle().call('pcnn_layernorm_bwd', c_int(N), c_int(F), _p(x), _p(gamma), _p(stats[0]), _p(stats[1]), _p(dy), _p(dx), _p(dgamma), _p(dbeta))
    return dx


def softmax_fwd(x):
    N, F = x.shape
    y = empty((N, F), x.device)
    handle().call('pcnn_softmax_fwd', c_int(N), c_int(F), _p(x), _p(y))
    return y


def softmax_bwd(y, dy):
    N, F = y.shape
    dx = empty((N, F), y.device)
    handle().call('pcnn_softmax_bwd', c_int(N), c_int(F), _p(y), _p(dy), _p(dx))
    return dx


def spp_max_fwd(x, bins):
    N, H, W, C = x.shape
    assert x.is_contiguous()
    nb = bins.shape[0]
    out = empty((N, nb), x.device)
    arg = torch.empty((N, nb), dtype=torch.int32, device=x.device)
    handle().call('pcnn_spp_max_fwd', c_int(N), c_int(H), c_int(W), c_int(C), c_int(nb), _p(bins), _p(x), _p(out), _p(arg))
    return out, arg


def spp_max_bwd(arg, dout, x_shape):
    N, H, W, C = x_shape
    dx = empty((N, H, W, C), dout.device)
    handle().call('pcnn_spp_max_bwd', c_int(N), c_int(H), c_int(W), c_int(C), c_int(arg.shape[1]), _p(arg), _p(dout), _p(dx))
    return dx


# ----------------------------------------------------------------------------- elementwise
def assemble_input(rhs_nhw, use_pos=True):
    N, H, W = rhs_nhw.shape
    C = 3 if use_pos else 1
    out = empty((N, H, W, C), rhs_nhw.device)
    handle().call('pcnn_assemble_input', c_int(N), c_int(H), c_int(W), _p(rhs_nhw), c_int(1 if use_pos else 0), _p(out), c_int(C))
    return out


def axpby(alpha, x, beta, y):
    """y = alpha*x + beta*y on NHWC tensors (channel slices allowed)."""
    N, H, W, C = x.shape
    handle().call('pcnn_axpby', c_int64(N * H * W), c_int(C), c_float(alpha), _p(x), c_int(_ld(x)), c_float(beta), _p(y), c_int(_ld(y)))
    return y


def subsample(x, stride):
    """x[:, ::s, ::s, :] (the strided form of the fused pad+conv keeps every s-th output of the stride-1 result)."""
    N, H, W, C = x.shape
    y = empty((N, -(-H // stride), -(-W // stride), C), x.device)
    handle().call('pcnn_subsample', c_int(N), c_int(H), c_int(W), c_int(C), c_int(stride), _p(x), c_int(_ld(x)), _p(y), c_int(_ld(y)), c_int(0))
    return y


def subsample_bwd(dy, full_hw, stride):
    """Adjoint of subsample: the coarse gradient scattered into a zero full-resolution tensor."""
    N, Ho, Wo, C = dy.shape
    H, W = full_hw
    dx = empty((N, H, W, C), dy.device)
    handle().call('pcnn_subsample', c_int(N), c_int(H), c_int(W), c_int(C), c_int(stride), _p(dy), c_int(_ld(dy)), _p(dx), c_int(_ld(dx)), c_int(1))
    return dx


def axpby_flat(alpha, x, beta, y):
    n = x.numel()
    handle().call('pcnn_axpby', c_int64(1), c_int(n), c_float(alpha), _p(x), c_int(n), c_float(beta), _p(y), c_int(n))
    return y


def channel_scale_fwd(x, s, out=None):
    N, H, W, C = x.shape
    y = out if out is not None else empty((N, H, W, C), x.device)
    handle().call('pcnn_channel_scale_fwd', c_int(N), c_int64(H * W), c_int(C), _p(x), c_int(_ld(x)), _p(s), _p(y), c_int(_ld(y)))
    return y


def channel_scale_bwd(x, s, dy, ws=None):
    N, H, W, C = x.shape
    lib = _lib.load()
    wsb = (ws or _default_ws).get(lib.pcnn_channel_scale_workspace(c_int(N), c_int64(H * W), c_int(C)), x.device)
    dx = empty((N, H, W, C), x.device)
    ds = empty((N, C), x.device)
    handle().call('pcnn_channel_scale_bwd', c_int(N), c_int64(H * W), c_int(C), _p(x), c_int(_ld(x)), _p(s), _p(dy), c_int(_ld(dy)), _p(dx), c_int(_ld(dx)),
                  _p(ds), _p(wsb), c_size_t(wsb.numel() * 4))
    return dx, ds


def channel_scale_bwd_post(x, s, dy, post, ws=None):
    """channel_scale_bwd + the activation backward of the layer whose saved activation output IS x (pcnn_channel_scale_bwd_post): returns (dz, ds) with
    dz = dy s act'(x) and post.dbias written, post.applied set - or None when the offer does not fit (x is not the offered activation, an inference-mode
    BatchNormalization or a raw copy is part of it, fusion switched off)."""
    if post is None or not _post_fusion or post.bn_scale is not None or post.want_raw or post.a.data_ptr() != x.data_ptr() or tuple(post.a.shape) != tuple(x.shape):
        return None
    N, H, W, C = x.shape
    lib = _lib.load()
    wsb = (ws or _default_ws).get(2 * lib.pcnn_channel_scale_workspace(c_int(N), c_int64(H * W), c_int(C)), x.device)
    dz = empty((N, H, W, C), x.device)
    ds = empty((N, C), x.device)
    handle().call('pcnn_channel_scale_bwd_post', c_int(N), c_int64(H * W), c_int(C), _p(x), c_int(_ld(x)), _p(s), _p(dy), c_int(_ld(dy)), _p(dz), c_int(_ld(dz)),
                  _p(ds), c_int(ACTS[post.act]), c_float(LEAKY_ALPHA), _p(post.dbias), _p(wsb), c_size_t(wsb.numel() * 4))
    post.applied = True
    return dz, ds


def sample_scale_fwd(x, g):
    N = x.shape[0]
    y = torch.empty_like(x)
    handle().call('pcnn_sample_scale_fwd', c_int(N), c_int64(x.numel() // N), _p(x), _p(g), _p(y))
    return y


def sample_scale_bwd(x, g, dy):
    N = x.shape[0]
    dx = torch.empty_like(x)
    dg = empty((N,), x.device)
    handle().call('pcnn_sample_scale_bwd', c_int(N), c_int64(x.numel() // N), _p(x), _p(g), _p(dy), _p(dx), _p(dg))
    return dx, dg


def bc_ring_fwd(x, neumann):
    N, H, W = x.shape[0], x.shape[1], x.shape[2]
    y = torch.empty_like(x)
    handle().call('pcnn_bc_ring_fwd', c_int(N), c_int(H), c_int(W), c_int(1 if neumann else 0), _p(x), _p(y))
    return y


def bc_ring_bwd(dy, neumann):
    N, H, W = dy.shape[0], dy.shape[1], dy.shape[2]
    dx = torch.empty_like(dy)
    handle().call('pcnn_bc_ring_bwd', c_int(N), c_int(H), c_int(W), c_int(1 if neumann else 0), _p(dy), _p(dx))
    return dx


def jacobi_sweep(u, rhs, dx2):
    N, H, W = u.shape[0], u.shape[1], u.shape[2]
    out = torch.empty_like(u)
    handle().call('pcnn_jacobi_sweep', c_int(N), c_int(H), c_int(W), _p(u), _p(rhs), _p(dx2), _p(out))
    return out


def jacobi_sweep_bwd(dout, dx2):
    N, H, W = dout.shape[0], dout.shape[1], dout.shape[2]
    du = torch.empty_like(dout)
    handle().call('pcnn_jacobi_sweep_bwd', c_int(N), c_int(H), c_int(W), _p(dout), _p(dx2), _p(du))
    return du


# ----------------------------------------------------------------------------- loss / optimizer
def loss_partials(pred, target, G, lp_power=2.0):
    N = pred.shape[0]
    out = empty((N, 4), pred.device)
    handle().call('pcnn_loss_partials_p', c_int(N), c_int64(pred.numel() // N), _p(pred), _p(target), _p(G), c_float(lp_power), _p(out))
    return out


def loss_bwd(pred, target, G, c_mae, c_mse, c_int_, out=None, lp_power=2.0):
    N = pred.shape[0]
    d = out if out is not None else torch.empty_like(pred)
    handle().call('pcnn_loss_bwd_p', c_int(N), c_int64(pred.numel() // N), _p(pred), _p(target), _p(G), _p(c_mae), _p(c_mse), _p(c_int_), c_float(lp_power),
                  _p(d))
    return d


def pi_loss_partials(pred, rhs, kern):
    N, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]   # (N,1,H,W) or (N,H,W)
    out = empty((N,), pred.device)
    handle().call('pcnn_pi_loss_partials', c_int(N), c_int(H), c_int(W), c_int(kern.shape[-1]), _p(pred), _p(rhs), _p(kern), _p(out))
    return out


def pi_loss_bwd(pred, rhs, kern, coef, dpred):
    N, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]   # (N,1,H,W) or (N,H,W)
    handle().call('pcnn_pi_loss_bwd', c_int(N), c_int(H), c_int(W), c_int(kern.shape[-1]), _p(pred), _p(rhs), _p(kern), _p(coef), _p(dpred))
    return dpred


def adam_step(w, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, vhat=None):
    weights_changed()                                                 # a raw-pointer write torch does not count: cached filter spectra go stale
    handle().call('pcnn_adam_amsgrad_step', c_int64(w.numel()), _p(w), _p(g), _p(m), _p(v), _p(vhat), c_float(lr), c_float(beta1), c_float(beta2),
                  c_float(eps), c_int(step), c_float(grad_scale))


def sgd_step(w, g, lr, grad_scale=1.0):
    weights_changed()                                                 # a raw-pointer write torch does not count: cached filter spectra go stale
    handle().call('pcnn_sgd_step', c_int64(w.numel()), _p(w), _p(g), c_float(lr), c_float(grad_scale))


def sgd_momentum_step(w, g, v, lr, momentum, nesterov, grad_scale=1.0):
    weights_changed()                                                 # a raw-pointer write torch does not count: cached filter spectra go stale
    handle().call('pcnn_sgd_momentum_step', c_int64(w.numel()), _p(w), _p(g), _p(v), c_float(lr), c_float(momentum), c_int(1 if nesterov else 0), c_float(grad_scale))


# ----------------------------------------------------------------------------- Dirichlet_BC_NN_Legacy_2 / Poisson_CNN_Legacy
def dbc_assemble_input(bc_nl):
    """(N, L) boundary values -> (N, 1, L, 3) NHWC [bc, 1, cos(pi y/(L-1))] (models/Dirichlet_BC_NN_Legacy.py:136-139)."""
    N, Lh = bc_nl.shape
    out = empty((N, 1, Lh, 3), bc_nl.device)
    handle().call('pcnn_dbc_assemble_input', c_int(N), c_int(Lh), _p(bc_nl), _p(out), c_int(3))
    return out


def spp_avg_fwd(x, bins):
    N, H, W, C = x.shape
    nb = bins.shape[0]
    out = empty((N, nb), x.device)
    handle().call('pcnn_spp_avg_fwd', c_int(N), c_int(H), c_int(W), c_int(C), c_int(_ld(x)), c_int(nb), _p(bins), _p(x), _p(out))
    return out


def spp_avg_bwd(bins, dout, x_shape):
    N, H, W, C = x_shape
    dx = empty((N, H, W, C), dout.device)
    handle().call('pcnn_spp_avg_bwd', c_int(N), c_int(H), c_int(W), c_int(C), c_int(C), c_int(bins.shape[0]), _p(bins), _p(dout), _p(dx))
    return dx


def dbc_expand_fwd(f, sinh_table, d):
    """f (N,1,L,M) boundary features, sinh_table (M,X), d (N,M) -> (N,X,L,M+2) (einsum + positional embeddings)."""
    N, _, Lh, M = f.shape
    X = sinh_table.shape[1]
    out = empty((N, X, Lh, M + 2), f.device)
    handle().call('pcnn_dbc_expand_fwd', c_int(N), c_int(X), c_int(Lh), c_int(M), _p(f), c_int(_ld(f)), _p(sinh_table), _p(d), _p(out), c_int(M + 2))
    return out


def dbc_expand_bwd(dout, f, sinh_table, d, ws=None):
    N, _, Lh, M = f.shape
    X = sinh_table.shape[1]
    lib = _lib.load()
    wsb = (ws or _default_ws).get(lib.pcnn_dbc_expand_bwd_workspace(c_int(N), c_int(Lh), c_int(M)), f.device)
    df = empty((N, 1, Lh, M), f.device)
    dd = empty((N, M), f.device)
    handle().call('pcnn_dbc_expand_bwd', c_int(N), c_int(X), c_int(Lh), c_int(M), _p(dout), c_int(_ld(dout)), _p(f), c_int(_ld(f)), _p(sinh_table),
                  _p(d), _p(df), c_int(M), _p(dd), _p(wsb), c_size_t(wsb.numel() * 4))
    return df, dd


def set_max_magnitude_fwd(x, target=1.0):
    """Returns (x * target / max|x| per sample, factors); x (N, ...) contiguous, not modified."""
    N = x.shape[0]
    y = x.clone()
    t = torch.full((N,), float(target), dtype=torch.float32, device=x.device)
    fac = empty((N,), x.device)
    handle().call('pcnn_set_max_magnitude', c_int(N), c_int64(x.numel() // N), _p(t), _p(y), _p(fac))
    return y, fac


def set_max_magnitude_bwd(x, dy, target=1.0):
    N = x.shape[0]
    t = torch.full((N,), float(target), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    handle().call('pcnn_set_max_magnitude_bwd', c_int(N), c_int64(x.numel() // N), _p(t), _p(x), _p(dy), _p(dx))
    return dx


def set_first_row(y_nxl, bc_nl=None):
    """In place: y[n, 0, :] = bc[n, :] (or zeros)."""
    N, X, Lh = y_nxl.shape
    handle().call('pcnn_set_first_row', c_int(N), c_int(X), c_int(Lh), _p(bc_nl), _p(y_nxl))
    return y_nxl


def flip_rotate(x_nhw, transpose=False, flip_y=False, flip_x=False, alpha=None, out=None, accumulate=False):
    """out = reverse(transpose?(x)) per sample (dataset/utils/flip_and_rotate_tensor.py), optionally * alpha[n] and accumulated."""
    N, H, W = x_nhw.shape
    Ho, Wo = (W, H) if transpose else (H, W)
    if out is None:
        out = empty((N, Ho, Wo), x_nhw.device)
        accumulate = False
    assert tuple(out.shape) == (N, Ho, Wo) and out.is_contiguous() and x_nhw.is_contiguous()
    handle().call('pcnn_flip_rotate', c_int(N), c_int(Ho), c_int(Wo), c_int(int(transpose)), c_int(int(flip_y)), c_int(int(flip_x)), _p(x_nhw),
                  _p(alpha), c_int(int(accumulate)), _p(out))
    return out


# ----------------------------------------------------------------------------- per-sample-filter ("metalearning") convolutions
def _row_stride(t):
    """floats between consecutive samples of a (N, F) parameter matrix emitted by a hyper-network (a view of it may start at a column offset)"""
    assert t.dim() == 2 and t.stride(1) == 1
    return int(t.stride(0))


def grouped_conv2d_fwd(x, w_all, w_shape, bias_all=None, *, pad_top, pad_left, out_hw, pad_mode='CONSTANT', pad_value=0.0, act='linear', flip_transpose=False,
                       out=None):
    """ONE launch for the batch: sample n is convolved with ITS filter w_all[n] (a row of the hyper-network's output matrix, HWIO w_shape) and
    bias bias_all[n] (layers/metalearning_conv.py:148-169).  flip_transpose: w_shape is then the FORWARD filter's (kh, kw, Cout_here, Cin_here)
    and the kernel reads it flipped and transposed - the data gradient of the same layer."""
    N, H, W, Cin = x.shape
    kh, kw, a, b = w_shape
    Cout = a if flip_transpose else b
    assert (b if flip_transpose else a) == Cin, (w_shape, Cin, flip_transpose)
    Ho, Wo = out_hw
    y = out if out is not None else empty((N, Ho, Wo, Cout), x.device)
    d = conv_desc(x.shape, _ld(x), (kh, kw, Cin, Cout), (Ho, Wo), _ld(y), pad_top, pad_left, pad_mode, pad_value, act)
    if _grouped_per_sample(d, 0):
        # wide layers (beyond the grouped kernels' 32 output channels / 31 taps, or wide enough that the ordinary kernels - spectral route, 32-wide
        # implicit GEMM - are the faster machine): one ordinary launch per sample with that sample's filter.  ADVICE r3: no channel count fails.
        nk = kh * kw * Cin * Cout
        for n in range(N):
            wn = w_all[n, :nk].view(kh, kw, a, b)
            wn = flip_transpose_weights(wn) if flip_transpose else wn
            conv2d_fwd(x[n:n + 1], wn, bias_all[n] if bias_all is not None else None, pad_top=pad_top, pad_left=pad_left, out_hw=(Ho, Wo), pad_mode=pad_mode,
                       pad_value=pad_value, act=act, out=y[n:n + 1])
        return y
    handle().call('pcnn_grouped_conv2d_fwd', byref(d), _p(x), _p(w_all), ctypes.c_longlong(_row_stride(w_all)), _p(bias_all),
                  ctypes.c_longlong(_row_stride(bias_all) if bias_all is not None else 0), c_int(1 if flip_transpose else 0), _p(y))
    return y


def grouped_conv2d_wgrad(x, dz, w_shape, dw_all, *, pad_top, pad_left, pad_mode='CONSTANT', pad_value=0.0, ws=None):
    """dw_all[n] (a row of the (N, F) gradient matrix) = filter gradient of sample n, one launch (+ a fixed-order reduction)."""
    kh, kw, Cin, Cout = w_shape
    d = conv_desc(x.shape, _ld(x), w_shape, (dz.shape[1], dz.shape[2]), _ld(dz), pad_top, pad_left, pad_mode, pad_value)
    if _grouped_per_sample(d, 1):
        nk = kh * kw * Cin * Cout
        for n in range(x.shape[0]):
            conv2d_wgrad(x[n:n + 1], dz[n:n + 1], w_shape, pad_top=pad_top, pad_left=pad_left, pad_mode=pad_mode, pad_value=pad_value,
                         out=dw_all[n, :nk].view(kh, kw, Cin, Cout), ws=ws)
        return dw_all
    lib = _lib.load()
    lib.pcnn_grouped_conv2d_wgrad_workspace.restype = c_size_t
    wsb = (ws or _default_ws).get(lib.pcnn_grouped_conv2d_wgrad_workspace(byref(d)), x.device)
    handle().call('pcnn_grouped_conv2d_wgrad', byref(d), _p(x), _p(dz), _p(dw_all), ctypes.c_longlong(_row_stride(dw_all)), _p(wsb))
    return dw_all


def _grouped_per_sample(d, what):
    """True when a per-sample-filter layer leaves the grouped kernels for one ordinary launch per sample: shapes beyond the grouped vector-ALU
    kernels' limits (csrc/grouped_conv.hip: <= 32 output channels and <= 31 taps forward; Cin Cout <= 4096 and a 64 KB staging tile in the
    filter gradient), and layers that are not eligible for the matrix-core grouped kernel but wide enough (Cin Cout >= 256) that the
    ordinary kernels' matrix-core / spectral routes beat ~6 TFLOP/s of scalar FMAs (tools/bench_grouped.py: 7 x 7, 32 -> 32 at 10 x 200^2:
    6.3 ms grouped)."""
    if _lib.load().pcnn_grouped_conv2d_uses_mfma(byref(d), c_int(what)):
        return False
    if d.Cin * d.Cout >= 256 and not __import__('os').environ.get('PCNN_GROUPED_VALU'):     # (the developer switch keeps everything it can on the vector-ALU kernels)
        return True
    if what == 0:
        return d.Cout > 32 or d.kh > 31 or d.kw > 31
    return d.Cin * d.Cout > 4096 or ((128 + d.kw - 1) * d.Cin + 128 * d.Cout) * 4 > 64 * 1024


def grouped_uses_mfma(x_shape, w_shape, out_hw, what='fwd'):
    """True when pcnn_grouped_conv2d_fwd (what = 'fwd': forward / data gradient) or _wgrad ('wgrad') takes the matrix-core route
    (v_mfma_f32_4x4x1_16B_f32 grouped implicit GEMM) for this shape, False for the vector-ALU kernels (PCNN_GROUPED_VALU=1 forces those)."""
    d = conv_desc(x_shape, x_shape[3], w_shape, out_hw, w_shape[3], 0, 0, 'CONSTANT', 0.0)
    return bool(_lib.load().pcnn_grouped_conv2d_uses_mfma(byref(d), c_int(0 if what == 'fwd' else 1)))


def grouped_bias_grad(dz, dbias_all):
    N, H, W, C = dz.shape
    handle().call('pcnn_grouped_bias_grad', c_int(N), ctypes.c_longlong(H * W), c_int(C), _p(dz), c_int(_ld(dz)), _p(dbias_all), ctypes.c_longlong(_row_stride(dbias_all)))
    return dbias_all


def grouped_deconv_fwd(x, k_all, k_shape, bias_all, out_hw, f, out=None):
    """Per-sample conv2d_transpose with kernel = stride = f (layers/metalearning_deconvupscale.py:104-137), k_shape (f, f, Cout, Cin)."""
    N, hc, wc, Cin = x.shape
    assert x.is_contiguous() and k_shape[0] == f and k_shape[1] == f and k_shape[3] == Cin
    Cout = k_shape[2]
    y = out if out is not None else empty((N, out_hw[0], out_hw[1], Cout), x.device)
    assert y.is_contiguous()
    handle().call('pcnn_grouped_deconv_fwd', c_int(N), c_int(hc), c_int(wc), c_int(Cin), c_int(out_hw[0]), c_int(out_hw[1]), c_int(Cout), c_int(f), _p(x), _p(k_all),
                  ctypes.c_longlong(_row_stride(k_all)), _p(bias_all), ctypes.c_longlong(_row_stride(bias_all) if bias_all is not None else 0), _p(y))
    return y


def grouped_deconv_bwd_data(dy, k_all, k_shape, coarse_hw, f, out=None):
    N, H, W, Cout = dy.shape
    Cin = k_shape[3]
    assert dy.is_contiguous()
    dx = out if out is not None else empty((N, coarse_hw[0], coarse_hw[1], Cin), dy.device)
    handle().call('pcnn_grouped_deconv_bwd_data', c_int(N), c_int(coarse_hw[0]), c_int(coarse_hw[1]), c_int(Cin), c_int(H), c_int(W), c_int(Cout), c_int(f), _p(dy), _p(k_all),
                  ctypes.c_longlong(_row_stride(k_all)), _p(dx))
    return dx


def grouped_deconv_bwd_filter(x, dy, f, dk_all, dbias_all=None):
    N, hc, wc, Cin = x.shape
    _, H, W, Cout = dy.shape
    assert x.is_contiguous() and dy.is_contiguous()
    handle().call('pcnn_grouped_deconv_bwd_filter', c_int(N), c_int(hc), c_int(wc), c_int(Cin), c_int(H), c_int(W), c_int(Cout), c_int(f), _p(x), _p(dy), _p(dk_all),
                  ctypes.c_longlong(_row_stride(dk_all)), _p(dbias_all), ctypes.c_longlong(_row_stride(dbias_all) if dbias_all is not None else 0))
    return dk_all
