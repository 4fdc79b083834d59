"""Thin Python wrappers over the libpcnn C-ABI.  Tensors are torch CUDA float32 in NHWC with an arbitrary channel
stride (so channel slices of wider buffers are accepted).  torch is used for allocation and streams only."""
from ctypes import byref, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

from . import _lib
from ._lib import ConvDesc

PAD_MODES = {'CONSTANT': 0, 'SYMMETRIC': 1, 'REFLECT': 2}
ACTS = {'linear': 0, 'leaky_relu': 1, 'tanh': 2, 'relu': 3}
LEAKY_ALPHA = 0.2   # tf.nn.leaky_relu default

_handles = {}


def handle():
    """Handle bound to the current device and torch's current stream."""
    dev = torch.cuda.current_device()
    st = torch.cuda.current_stream().cuda_stream
    h = _handles.get(dev)
    if h is None:
        h = _lib.Handle(dev, st)
        h._stream = st
        _handles[dev] = h
    elif h._stream != st:
        h.set_stream(st)
        h._stream = st
    return h


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _chk(t, name='tensor'):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError('%s must be a CUDA float32 tensor' % name)
    if t.stride(-1) != 1:
        raise ValueError('%s must have unit channel stride (NHWC)' % name)
    return t


def _ld(t):
    """channel stride (floats between consecutive pixels) of an NHWC tensor; checks pixel-linear strides."""
    _chk(t)
    N, H, W, C = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else max(C, 1))
    if (W > 1 and t.stride(2) != ld) or (H > 1 and t.stride(1) != W * ld) or (N > 1 and t.stride(0) != H * W * ld):
        raise ValueError('tensor is not pixel-linear NHWC (strides %s for shape %s)' % (t.stride(), tuple(t.shape)))
    return ld


def workspace(nbytes, device):
    return torch.empty((max(int(nbytes), 4) + 3) // 4, dtype=torch.float32, device=device)


def conv2d_fwd(x, w, bias=None, *, pad_top, pad_left, out_hw=None, pad_mode='CONSTANT', pad_value=0.0, act='linear',
               bn_scale=None, bn_shift=None, residual=None, out=None, act_out=None):
    N, H, W, Cin = x.shape
    kh, kw, ci, Cout = w.shape
    assert ci == Cin and w.is_contiguous()
    Ho, Wo = out_hw if out_hw is not None else (H, W)
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    d = ConvDesc(N, H, W, Cin, _ld(x), Ho, Wo, Cout, _ld(out), kh, kw, pad_top, pad_left, PAD_MODES[pad_mode.upper()],
                 float(pad_value), ACTS[act], LEAKY_ALPHA, _ld(residual) if residual is not None else 0,
                 _ld(act_out) if act_out is not None else 0)
    handle().call('pcnn_conv2d_fwd', byref(d), _p(x), _p(w), _p(bias), _p(bn_scale), _p(bn_shift), _p(residual), _p(out), _p(act_out))
    return out


def flip_transpose_weights(w):
    kh, kw, ci, co = w.shape
    wt = torch.empty((kh, kw, co, ci), dtype=torch.float32, device=w.device)
    handle().call('pcnn_conv2d_flip_transpose_weights', _p(w), _p(wt), c_int(kh), c_int(kw), c_int(ci), c_int(co))
    return wt
