"""CPU oracle: fp64 numpy restatement of the TensorFlow-2.4 op semantics the reference's hot path uses.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the product
package ``poisson_cnn_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED for the TF-op arithmetic: TensorFlow is absent from the build
environment (``import tensorflow`` -> ModuleNotFoundError) and the reference ships no
tests or golden vectors for conv / pool / resize / BN.  Each op
below restates the TF 2.4 kernel semantics from knowledge of TF, and is cross-checked
against independent PyTorch-CPU implementations where semantics coincide
(tests/test_oracle_ops.py).  The reference-authored known answers that DO exist
(FD stencils, split_indices, the integral-loss known answer) are pinned in
tests/test_oracle_golden.py.

All tensors are channels_first (N, C, H, W) float64, like the reference configs
(experiments/hpnn.json:5).  Citations are relative to /root/reference/poisson_CNN/.
"""
import numpy as np

LEAKY_ALPHA = 0.2  # tf.nn.leaky_relu default alpha
BN_EPS = 1e-3      # tf.keras.layers.BatchNormalization default epsilon


# ----------------------------------------------------------------------------- activations
def activation(x, name):
    """'tf.nn.leaky_relu' (alpha 0.2), 'tf.nn.tanh', 'linear' (utils/convert_tf_object_names.py:13-18
    turns these JSON strings into TF callables)."""
    name = canonical_activation(name)
    if name == 'linear':
        return x
    if name == 'leaky_relu':
        return np.where(x > 0, x, LEAKY_ALPHA * x)
    if name == 'tanh':
        return np.tanh(x)
    if name == 'relu':
        return np.maximum(x, 0)
    if name == 'softmax':                                   # Keras: over the last axis
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return e / e.sum(axis=-1, keepdims=True)
    raise ValueError(name)


def canonical_activation(name):
    if name is None:
        return 'linear'
    if not isinstance(name, str):
        raise ValueError('activation must be a string name, got %r' % (name,))
    n = name.lower()
    for pref in ('tf.nn.', 'tf.keras.activations.', 'tf.math.'):
        if n.startswith(pref):
            n = n[len(pref):]
    if n in ('linear', 'leaky_relu', 'tanh', 'relu', 'softmax'):
        return n
    raise ValueError('unsupported activation %r' % (name,))


# ----------------------------------------------------------------------------- padding
def advanced_pad_amounts(k):
    """utils/apply_advanced_padding_and_call_conv_layer.py:9-10: before = k//2, after = k//2 - (1 - k%2)."""
    return k // 2, k // 2 - (1 - k % 2)


def pad2d(x, pads, mode, value=0.0):
    """tf.pad on the two spatial dims (utils/apply_advanced_padding_and_call_conv_layer.py:18).
    pads = ((top, bottom), (left, right)); mode CONSTANT / SYMMETRIC (edge-inclusive mirror) / REFLECT."""
    mode = mode.upper()
    pw = ((0, 0), (0, 0), tuple(pads[0]), tuple(pads[1]))
    if mode == 'CONSTANT':
        return np.pad(x, pw, mode='constant', constant_values=value)
    for (a, b), n in zip(pads, x.shape[2:]):
        lim = n if mode == 'SYMMETRIC' else n - 1
        if a > lim or b > lim:  # tf.pad raises for these
            raise ValueError('pad %s exceeds what tf.pad %s allows for size %d' % ((a, b), mode, n))
    if mode == 'SYMMETRIC':
        return np.pad(x, pw, mode='symmetric')
    if mode == 'REFLECT':
        return np.pad(x, pw, mode='reflect')
    raise ValueError(mode)


# ----------------------------------------------------------------------------- convolution
def conv2d_valid(x, w, bias=None, stride=1):
    """tf.nn.conv2d VALID (cross-correlation).  x (N,Cin,H,W); w HWIO (kh,kw,Cin,Cout)."""
    N, C, H, W = x.shape
    kh, kw, ci, co = w.shape
    assert ci == C, (ci, C)
    Ho = (H - kh) // stride + 1
    Wo = (W - kw) // stride + 1
    xl = np.ascontiguousarray(x.transpose(0, 2, 3, 1))  # NHWC
    out = np.zeros((N, Ho, Wo, co), dtype=np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xl[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            out += patch @ w[i, j]
    if bias is not None:
        out += bias
    return out.transpose(0, 3, 1, 2)


def padded_conv2d(x, w, bias, padding_mode='CONSTANT', constant_padding_value=0.0, act='linear', stride=1):
    """pad_and_apply_convolution (utils/apply_advanced_padding_and_call_conv_layer.py:16-20)."""
    kh, kw = w.shape[:2]
    xp = pad2d(x, (advanced_pad_amounts(kh), advanced_pad_amounts(kw)), padding_mode, constant_padding_value)
    return activation(conv2d_valid(xp, w, bias, stride), act)


def same_conv2d(x, w, bias, act='linear'):
    """Keras Conv2D(padding='same', strides=1): zero pad total k-1, before=(k-1)//2
    (models/Homogeneous_Poisson_NN_Legacy.py:71,75,95; layers/Scaling.py:28)."""
    kh, kw = w.shape[:2]
    pads = (((kh - 1) // 2, kh - 1 - (kh - 1) // 2), ((kw - 1) // 2, kw - 1 - (kw - 1) // 2))
    return activation(conv2d_valid(pad2d(x, pads, 'CONSTANT', 0.0), w, bias), act)


def conv2d_transpose_same(x, k, bias, out_hw, stride, act='linear'):
    """tf.nn.conv2d_transpose(x, k, output_shape, strides=f, padding='SAME') + bias + act
    (layers/deconvupscale.py:103-108).  k is (kh, kw, Cout, Cin) (deconvupscale.py:58).
    It is the adjoint of the SAME forward conv: out[i] += x[o] k[t] for o*f + t - pad_before = i,
    pad_total = max((h-1) f + kh - H, 0), pad_before = pad_total // 2."""
    N, Ci, h, w_ = x.shape
    kh, kw, Co, ci2 = k.shape
    assert ci2 == Ci
    H, W = out_hw
    assert h == -(-H // stride) and w_ == -(-W // stride), 'SAME transpose needs in = ceil(out/stride)'
    pb_y = max((h - 1) * stride + kh - H, 0) // 2
    pb_x = max((w_ - 1) * stride + kw - W, 0) // 2
    full = np.zeros((N, Co, max((h - 1) * stride + kh, pb_y + H), max((w_ - 1) * stride + kw, pb_x + W)))   # kh < stride: rows no tap reaches stay 0
    for ty in range(kh):
        for tx in range(kw):
            contrib = np.einsum('nchw,oc->nohw', x, k[ty, tx])
            full[:, :, ty:ty + (h - 1) * stride + 1:stride, tx:tx + (w_ - 1) * stride + 1:stride] += contrib
    out = full[:, :, pb_y:pb_y + H, pb_x:pb_x + W]
    if bias is not None:
        out = out + bias[None, :, None, None]
    return activation(out, act)


# ----------------------------------------------------------------------------- pooling
def _same_pool_geometry(n, f):
    out = -(-n // f)
    pad_total = max((out - 1) * f + f - n, 0)
    return out, pad_total // 2


def pool2d_same(x, f, kind='average'):
    """tf.keras.layers.{Average,Max}Pooling2D(pool_size=f, strides=f, padding='same')
    (utils/get_pooling_method.py:3-6; blocks/bottleneck_block.py:36-37).  Window o covers
    [o f - pad_before, o f - pad_before + f) clipped to the image; the average divides by the
    number of valid (un-padded) elements."""
    N, C, H, W = x.shape
    Ho, pby = _same_pool_geometry(H, f)
    Wo, pbx = _same_pool_geometry(W, f)
    out = np.empty((N, C, Ho, Wo))
    for oy in range(Ho):
        y0, y1 = max(oy * f - pby, 0), min(oy * f - pby + f, H)
        for ox in range(Wo):
            x0, x1 = max(ox * f - pbx, 0), min(ox * f - pbx + f, W)
            win = x[:, :, y0:y1, x0:x1]
            out[:, :, oy, ox] = win.mean(axis=(2, 3)) if kind.lower().startswith('av') else win.max(axis=(2, 3))
    return out


# ----------------------------------------------------------------------------- resize (tf.image.resize, antialias=False)
_BICUBIC_TABLE = 1024


def _bicubic_table(a):
    """tensorflow/core/kernels/image/resize_bicubic_op.cc InitCoeffsTable, float32 arithmetic."""
    t = np.zeros((_BICUBIC_TABLE + 1) * 2, dtype=np.float32)
    a = np.float32(a)
    for i in range(_BICUBIC_TABLE + 1):
        x = np.float32(i) / np.float32(_BICUBIC_TABLE)
        t[2 * i] = ((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + np.float32(1)
        x = x + np.float32(1)
        t[2 * i + 1] = ((a * x - np.float32(5) * a) * x + np.float32(8) * a) * x - np.float32(4) * a
    return t


def resize_matrix(n_in, n_out, method, half_pixel=True, align_corners=False):
    """Dense (n_out, n_in) interpolation matrix of one axis for the TF resize kernels.
    half_pixel=True  -> tf.image.resize v2 (layers/Upsample.py:56): bilinear / nearest / bicubic (Keys a=-0.5,
                        1024-entry table, out-of-range taps zeroed and weights renormalised).
    align_corners    -> tf.compat.v1.image.resize_images(align_corners=True) (dataset/utils/image_resize.py:20):
                        legacy bicubic a=-0.75 with clamped indices."""
    method = method.lower()
    M = np.zeros((n_out, n_in))
    if align_corners and n_out > 1:
        scale = np.float32(n_in - 1) / np.float32(n_out - 1)
    else:
        scale = np.float32(n_in) / np.float32(n_out)
    for o in range(n_out):
        if half_pixel:
            src = (np.float32(o) + np.float32(0.5)) * scale - np.float32(0.5)
        else:
            src = np.float32(o) * scale
        if method == 'nearest':
            if half_pixel:
                idx = int(np.floor((np.float32(o) + np.float32(0.5)) * scale))
            else:
                idx = int(np.round(np.float32(o) * scale)) if align_corners else int(np.floor(np.float32(o) * scale))
            M[o, min(max(idx, 0), n_in - 1)] = 1.0
        elif method == 'bilinear':
            f = np.floor(src)
            lo = int(max(f, 0))
            hi = int(min(np.ceil(src), n_in - 1))
            lerp = float(np.float32(src - f))
            M[o, lo] += 1.0 - lerp
            M[o, hi] += lerp
        elif method == 'bicubic':
            a = -0.5 if half_pixel else -0.75
            tab = _bicubic_table(a)
            loc = int(np.floor(src))
            delta = np.float32(src - np.float32(loc))
            off = int(np.rint(delta * np.float32(_BICUBIC_TABLE)))  # lrintf: round-half-even
            ws = [tab[off * 2 + 1], tab[off * 2], tab[(_BICUBIC_TABLE - off) * 2], tab[(_BICUBIC_TABLE - off) * 2 + 1]]
            idxs = [loc - 1, loc, loc + 1, loc + 2]
            if half_pixel:  # use_keys_cubic: zero the out-of-range taps, renormalise
                cl = [min(max(i, 0), n_in - 1) for i in idxs]
                ws = [np.float32(wt if c == i else 0.0) for wt, c, i in zip(ws, cl, idxs)]
                s = np.float32(ws[0] + ws[1] + ws[2] + ws[3])
                if abs(s) >= 1000.0 * np.finfo(np.float32).tiny:
                    inv = np.float32(1.0) / s
                    ws = [wt * inv for wt in ws]
                idxs = cl
            else:
                idxs = [min(max(i, 0), n_in - 1) for i in idxs]
            for i, wt in zip(idxs, ws):
                M[o, i] += float(wt)
        else:
            raise ValueError(method)
    return M


def resize2d(x, out_hw, method, half_pixel=True, align_corners=False):
    """Separable resize of the two spatial axes of (N,C,H,W)."""
    Ry = resize_matrix(x.shape[2], out_hw[0], method, half_pixel, align_corners)
    Rx = resize_matrix(x.shape[3], out_hw[1], method, half_pixel, align_corners)
    return np.einsum('oh,nchw,pw->ncop', Ry, x, Rx, optimize=True)


# ----------------------------------------------------------------------------- normalisation / dense / misc
def batchnorm_inference(x, gamma, beta, mean, var, eps=BN_EPS):
    """BatchNormalization(axis=1) in inference mode (models/Homogeneous_Poisson_NN_Legacy.py:55,
    blocks/resnet.py:26-27; mode discussion: SURVEY.md row H4)."""
    s = gamma / np.sqrt(var + eps)
    return x * s[None, :, None, None] + (beta - mean * s)[None, :, None, None]


def batchnorm_training(x, gamma, beta, eps=BN_EPS):
    """Training-mode BN: biased batch statistics over (N,H,W)."""
    mean = x.mean(axis=(0, 2, 3))
    var = x.var(axis=(0, 2, 3))
    return batchnorm_inference(x, gamma, beta, mean, var, eps), mean, var


def dense(x, w, b, act='linear'):
    return activation(x @ w + b, act)


def split_indices(n, sections):
    """dataset/utils/split_indices.py:4-26: first n%sections bins get one extra element."""
    per, extra = divmod(int(n), int(sections))
    sizes = [0] + [per + 1] * extra + [per] * (sections - extra)
    return np.cumsum(sizes)


def spatial_pyramid_pool(x, levels, kind='max'):
    """layers/SpatialPyramidPool.py:35-66 with equal_split_tensor_slice
    (dataset/utils/equal_split_tensor_slice.py:40-57): for every bin the reduction runs over
    channels AND the spatial bin (tf.map_fn over the batch of pooling_func, SpatialPyramidPool.py:43-44)."""
    N = x.shape[0]
    feats = []
    for lv in levels:
        lv = [lv, lv] if isinstance(lv, int) else (list(lv) * 2 if len(lv) == 1 else list(lv))
        iy = split_indices(x.shape[2], lv[0])
        ix = split_indices(x.shape[3], lv[1])
        for by in range(lv[0]):
            for bx in range(lv[1]):
                b = x[:, :, iy[by]:iy[by + 1], ix[bx]:ix[bx + 1]].reshape(N, -1)
                feats.append(b.max(axis=1) if kind.lower() == 'max' else b.mean(axis=1))
    return np.stack(feats, axis=1)


def bc_ring(x, mode):
    """out = tf.pad(out[..., 1:-1, 1:-1], 1, mode) (models/Homogeneous_Poisson_NN_Legacy.py:106-113,251):
    CONSTANT 0 (Dirichlet) or SYMMETRIC (Neumann: ring := adjacent interior value)."""
    return pad2d(x[:, :, 1:-1, 1:-1], ((1, 1), (1, 1)), mode, 0.0)


def get_fd_coefficients(stencil_positions, order):
    """dataset/utils/get_fd_coefficients.py:4-19 (Vandermonde inverse)."""
    from math import factorial
    pos = np.array(sorted(stencil_positions), dtype=np.float64)
    V = np.array([pos ** k for k in range(len(pos))])
    rhs = np.zeros(len(pos))
    rhs[order] = factorial(order)
    return np.linalg.solve(V, rhs)


def build_fd_coefficients(stencil_size, orders, ndims=2):
    """dataset/utils/build_fd_coefficients.py:5-42: (ndims, s0, s1) cross-shaped stencils."""
    if isinstance(stencil_size, int):
        stencil_size = [stencil_size] * ndims
    if isinstance(orders, int):
        orders = [orders] * ndims
    ss = np.array(stencil_size)
    assert np.all(ss % 2 == 1)
    coeff = np.zeros([ndims] + list(ss))
    for d in range(ndims):
        sl = [d] + list(ss // 2)
        sl[d + 1] = slice(0, ss[d])
        coeff[tuple(sl)] += get_fd_coefficients(list(range(-(ss[d] // 2), ss[d] // 2 + 1)), orders[d])
    return coeff


def jacobi_iterations(guess, rhs, dx, n_iterations, stencil_sizes=(3, 3), orders=(2, 2)):
    """layers/JacobiIterationLayer.py:7-66: weighted-Jacobi sweeps of the FD Laplacian with a
    per-sample kernel; boundary ring of width stencil//2 is kept.  dx is (N, 2)."""
    coeff = build_fd_coefficients(list(stencil_sizes), list(orders), 2)
    c = tuple(s // 2 for s in stencil_sizes)
    diag = coeff[(Ellipsis,) + c].copy()          # (ndims,)
    lu = coeff.copy()
    lu[(Ellipsis,) + c] = 0.0
    dxp = (1.0 / dx) ** np.array(orders, dtype=np.float64)   # (N, 2)
    kern = np.einsum('dij,bd->bij', lu, dxp)
    dinv = 1.0 / (dxp @ diag)
    x = guess
    py, px = c
    for _ in range(n_iterations):
        new = x.copy()
        for b in range(x.shape[0]):
            cr = conv2d_valid(x[b:b + 1], kern[b][:, :, None, None])
            new[b:b + 1, :, py:-py, px:-px] = dinv[b] * (rhs[b:b + 1, :, py:-py, px:-px] - cr)
        x = new
    return x


def concat(xs, axis):
    return np.concatenate(xs, axis=axis)


def asarray(x):
    return np.asarray(x, dtype=np.float64)


# ---- additions for Dirichlet_BC_NN_Legacy_2 / Poisson_CNN_Legacy (oracle/dbcnn.py)
def einsum(eq, *xs):
    return np.einsum(eq, *[np.asarray(x, dtype=np.float64) for x in xs])


def set_max_magnitude_in_batch(x, target=1.0, return_factors=False):
    """dataset/utils/set_max_magnitude.py:4-50: per sample, x * target / max|x|."""
    x = np.asarray(x, dtype=np.float64)
    f = target / np.abs(x.reshape(x.shape[0], -1)).max(axis=1)
    y = x * f.reshape((-1,) + (1,) * (x.ndim - 1))
    return (y, f) if return_factors else y


def flip(x, axes):
    return np.flip(x, axis=tuple(axes)) if len(axes) else x


def transpose(x, perm):
    return np.transpose(x, perm)


def zeros_like(x):
    return np.zeros_like(x)
