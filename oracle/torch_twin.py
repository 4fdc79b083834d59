"""CPU oracle, autograd twin: the same op namespace as ``oracle.np_ops`` written with fp64 PyTorch-CPU
primitives so that ``oracle.hpnn.forward(torch_twin, ...)`` is differentiable.  Used ONLY to obtain
reference gradients (dL/dW, dL/dx) for the backward kernels; its forward is itself checked against the
numpy oracle (tests/test_oracle_ops.py), so the gradients are autograd's derivative of a verified forward.

TEST INFRASTRUCTURE ONLY - see oracle/np_ops.py header.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import np_ops

DT = torch.float64   # bench.py's cpu_baseline leg switches this to float32 for timing (set_dtype)


def set_dtype(dt):
    global DT
    DT = dt


def asarray(x):
    if isinstance(x, torch.Tensor):
        return x.to(DT)
    return torch.as_tensor(np.asarray(x), dtype=DT)


def concat(xs, axis):
    return torch.cat([asarray(x) for x in xs], dim=axis)


def activation(x, name):
    name = np_ops.canonical_activation(name)
    if name == 'linear':
        return x
    if name == 'leaky_relu':
        return torch.where(x > 0, x, np_ops.LEAKY_ALPHA * x)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'relu':
        return torch.clamp_min(x, 0)
    if name == 'softmax':                                   # Keras: over the last axis
        return torch.softmax(x, dim=-1)
    raise ValueError(name)


def _pad_index(n, before, after, mode):
    idx = np.arange(-before, n + after)
    if mode == 'SYMMETRIC':
        idx = np.where(idx < 0, -idx - 1, idx)
        idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
    else:  # REFLECT
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
    return torch.as_tensor(idx, dtype=torch.long)


def pad2d(x, pads, mode, value=0.0):
    mode = mode.upper()
    (t, b), (l, r) = pads
    if mode == 'CONSTANT':
        return F.pad(x, (l, r, t, b), mode='constant', value=float(value))
    iy = _pad_index(x.shape[2], t, b, mode)
    ix = _pad_index(x.shape[3], l, r, mode)
    return x.index_select(2, iy).index_select(3, ix)


_FFT_CONV = {'min_pixels': None, 'tile': 256}


def set_fft_conv(min_pixels=None, tile=256):
    """min_pixels = n: stride-1 convolutions on images of at least n pixels are evaluated by conv2d_valid_fft (overlap-save tiles of `tile`
    points, torch.fft in the working precision) instead of F.conv2d - PyTorch's fp64 CPU convolution unfolds the image into a (k k Cin) x (H W)
    matrix: 60 GB for ONE 15 x 15 x 32 layer at 1024^2, 5.5 minutes for its backward pass.  None (default): F.conv2d everywhere.  Used by
    tests/golden/make_atsize_golden.py for the 1024^2 training-step fixture; tests/test_oracle_ops.py pins the two evaluations to each
    other (values and gradients) to 1e-12."""
    _FFT_CONV['min_pixels'], _FFT_CONV['tile'] = min_pixels, int(tile)


def conv2d_valid_fft(x, w, bias=None, tile=256):
    """VALID cross-correlation y[n,o,p,q] = b[o] + sum_{c,i,j} x[n,c,p+i,q+j] w[i,j,c,o] by overlap-save in the frequency domain: the same
    numbers as F.conv2d to rounding (fp64: ~1e-15 relative), differentiable (torch.fft has autograd), memory = one tile's spectra + the filter
    spectrum.  x (N,C,H,W), w (kh,kw,C,O)."""
    w = asarray(w)
    kh, kw, C, O = w.shape
    N, _, H, W = x.shape
    Ho, Wo = H - kh + 1, W - kw + 1
    T = max(int(tile), 2 * max(kh, kw))
    Vy, Vx = T - kh + 1, T - kw + 1
    wh = torch.fft.rfft2(w.permute(3, 2, 0, 1), s=(T, T)).conj()                  # (O, C, T, T/2+1): correlation = product with the conjugate
    rows = []
    for y0 in range(0, Ho, Vy):
        cols = []
        for x0 in range(0, Wo, Vx):
            xt = x[:, :, y0:y0 + T, x0:x0 + T]
            xh = torch.fft.rfft2(xt, s=(T, T))                                     # zero-padded at the image's far edges
            # channel contraction per frequency, eight output channels at a time (an einsum of the same contraction is 8x slower on complex128;
            # the whole (O, C, T, T/2+1) product at once is a 0.5 GB temporary per tile)
            yh = torch.cat([(xh.unsqueeze(1) * wh[o:o + 8].unsqueeze(0)).sum(2) for o in range(0, O, 8)], 1)
            yt = torch.fft.irfft2(yh, s=(T, T))
            cols.append(yt[:, :, :min(Vy, Ho - y0), :min(Vx, Wo - x0)])
        rows.append(torch.cat(cols, 3))
    y = torch.cat(rows, 2)
    return y if bias is None else y + asarray(bias)[None, :, None, None]


def conv2d_valid(x, w, bias=None, stride=1):
    if stride == 1 and _FFT_CONV['min_pixels'] is not None and x.shape[2] * x.shape[3] >= _FFT_CONV['min_pixels']:
        return conv2d_valid_fft(x, w, bias, _FFT_CONV['tile'])
    w = asarray(w).permute(3, 2, 0, 1)
    return F.conv2d(x, w, None if bias is None else asarray(bias), stride=stride)


def padded_conv2d(x, w, bias, padding_mode='CONSTANT', constant_padding_value=0.0, act='linear', stride=1):
    kh, kw = w.shape[:2]
    xp = pad2d(x, (np_ops.advanced_pad_amounts(kh), np_ops.advanced_pad_amounts(kw)), padding_mode, constant_padding_value)
    return activation(conv2d_valid(xp, w, bias, stride), act)


def same_conv2d(x, w, bias, act='linear'):
    kh, kw = w.shape[:2]
    pads = (((kh - 1) // 2, kh - 1 - (kh - 1) // 2), ((kw - 1) // 2, kw - 1 - (kw - 1) // 2))
    return activation(conv2d_valid(pad2d(x, pads, 'CONSTANT', 0.0), w, bias), act)


def conv2d_transpose_same(x, k, bias, out_hw, stride, act='linear'):
    kh, kw = k.shape[:2]
    h, w_ = x.shape[2:]
    H, W = out_hw
    pb_y = max((h - 1) * stride + kh - H, 0) // 2
    pb_x = max((w_ - 1) * stride + kw - W, 0) // 2
    full = F.conv_transpose2d(x, asarray(k).permute(3, 2, 0, 1), None, stride=stride)
    full = F.pad(full, (0, max(pb_x + W - full.shape[3], 0), 0, max(pb_y + H - full.shape[2], 0)))        # kh < stride: positions no tap reaches stay 0
    out = full[:, :, pb_y:pb_y + H, pb_x:pb_x + W]
    if bias is not None:
        out = out + asarray(bias)[None, :, None, None]
    return activation(out, act)


def _pool_matrix(n, f):
    out, pb = np_ops._same_pool_geometry(n, f)
    M = np.zeros((out, n))
    for o in range(out):
        a, b = max(o * f - pb, 0), min(o * f - pb + f, n)
        M[o, a:b] = 1.0 / (b - a)
    return torch.as_tensor(M, dtype=DT)


def pool2d_same(x, f, kind='average'):
    if kind.lower().startswith('av'):
        return torch.einsum('oh,nchw,pw->ncop', _pool_matrix(x.shape[2], f), x, _pool_matrix(x.shape[3], f))
    H, W = x.shape[2:]
    Ho, pby = np_ops._same_pool_geometry(H, f)
    Wo, pbx = np_ops._same_pool_geometry(W, f)
    xp = F.pad(x, (pbx, Wo * f - W - pbx, pby, Ho * f - H - pby), value=float('-inf'))
    return F.max_pool2d(xp, f, f)


def resize2d(x, out_hw, method, half_pixel=True, align_corners=False):
    Ry = torch.as_tensor(np_ops.resize_matrix(x.shape[2], out_hw[0], method, half_pixel, align_corners), dtype=DT)
    Rx = torch.as_tensor(np_ops.resize_matrix(x.shape[3], out_hw[1], method, half_pixel, align_corners), dtype=DT)
    return torch.einsum('oh,nchw,pw->ncop', Ry, x, Rx)


def batchnorm_inference(x, gamma, beta, mean, var, eps=np_ops.BN_EPS):
    gamma, beta, mean, var = map(asarray, (gamma, beta, mean, var))
    s = gamma / torch.sqrt(var + eps)
    return x * s[None, :, None, None] + (beta - mean * s)[None, :, None, None]


def batchnorm_training(x, gamma, beta, eps=np_ops.BN_EPS):
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    return batchnorm_inference(x, gamma, beta, mean, var, eps), mean, var


def dense(x, w, b, act='linear'):
    y = x @ asarray(w)
    return activation(y if b is None else y + asarray(b), act)


def spatial_pyramid_pool(x, levels, kind='max'):
    N = x.shape[0]
    feats = []
    for lv in levels:
        lv = [lv, lv] if isinstance(lv, int) else (list(lv) * 2 if len(lv) == 1 else list(lv))
        iy = np_ops.split_indices(x.shape[2], lv[0])
        ix = np_ops.split_indices(x.shape[3], lv[1])
        for by in range(lv[0]):
            for bx in range(lv[1]):
                b = x[:, :, iy[by]:iy[by + 1], ix[bx]:ix[bx + 1]].reshape(N, -1)
                feats.append(b.max(dim=1).values if kind.lower() == 'max' else b.mean(dim=1))
    return torch.stack(feats, dim=1)


def bc_ring(x, mode):
    return pad2d(x[:, :, 1:-1, 1:-1], ((1, 1), (1, 1)), mode, 0.0)


def jacobi_iterations(guess, rhs, dx, n_iterations, stencil_sizes=(3, 3), orders=(2, 2)):
    coeff = np_ops.build_fd_coefficients(list(stencil_sizes), list(orders), 2)
    c = tuple(s // 2 for s in stencil_sizes)
    diag = torch.as_tensor(coeff[(Ellipsis,) + c].copy(), dtype=DT)
    lu = coeff.copy()
    lu[(Ellipsis,) + c] = 0.0
    lu = torch.as_tensor(lu, dtype=DT)
    dx = asarray(dx)
    rhs = asarray(rhs)
    dxp = (1.0 / dx) ** torch.as_tensor(list(orders), dtype=DT)
    kern = torch.einsum('dij,bd->bij', lu, dxp)
    dinv = 1.0 / (dxp @ diag)
    x = guess
    py, px = c
    for _ in range(n_iterations):
        cr = torch.cat([F.conv2d(x[b:b + 1], kern[b][None, None]) for b in range(x.shape[0])], 0)
        inner = dinv[:, None, None, None] * (rhs[:, :, py:-py, px:-px] - cr)
        mask = torch.zeros_like(x)
        mask[:, :, py:-py, px:-px] = 1.0
        x = x * (1 - mask) + F.pad(inner, (px, px, py, py))
    return x


# ---- additions for Dirichlet_BC_NN_Legacy_2 / Poisson_CNN_Legacy (oracle/dbcnn.py)
def einsum(eq, *xs):
    return torch.einsum(eq, *[asarray(x) for x in xs])


def set_max_magnitude_in_batch(x, target=1.0, return_factors=False):
    f = target / x.reshape(x.shape[0], -1).abs().amax(dim=1)      # amax splits its gradient evenly over ties, like tf.reduce_max
    y = x * f.reshape((-1,) + (1,) * (x.dim() - 1))
    return (y, f) if return_factors else y


def flip(x, axes):
    return torch.flip(x, dims=tuple(axes)) if len(axes) else x


def transpose(x, perm):
    return x.permute(*perm)


def zeros_like(x):
    return torch.zeros_like(x)
