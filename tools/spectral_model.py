"""Numpy model of the tiled spectral convolution (poisson_cnn_amd/csrc/spectral_*.hip): the exact index conventions of the kernels,
checked here against direct correlation.  Developer tool - not imported by the product or the tests.

Tile T = 32.  A window xw[y][x][c] (32 x 32 pixels, channels last) is transformed by two real matrix products per axis (the DFT as a
GEMM - on the GPU these are fp32 MFMAs with the channel index in the N / lane dimension, so every global access is a pixel's 128-byte
channel row):
  x axis (real -> half complex):  U[y][s][c] = sum_x G[s][x] xw[y][x][c],  s <= 16: Re(fx = s), s >= 17: Im(fx = s - 16)
  y axis: columns fx = 0 and 16 are real  -> half complex again (G);  columns fx = 1..15 are complex -> full complex DFT (Fc).
Spectrum rows R (1024 per tile, each a channel row):  R in [0,32): fx = 0;  [32,64): fx = 16;  64 + 64 (fx-1) + {fy | 32 + fy}: Re | Im.
"""
import numpy as np

T = 32


def G_fwd():
    x = np.arange(T)
    G = np.zeros((T, T))
    for s in range(T):
        G[s] = np.cos(2 * np.pi * s * x / T) if s <= 16 else -np.sin(2 * np.pi * (s - 16) * x / T)
    return G


def G_inv():
    """x[n] = sum_s Ginv[n][s] h[s] for a half-complex vector h (rows as in G_fwd)."""
    n = np.arange(T)
    Gi = np.zeros((T, T))
    for s in range(T):
        if s == 0:
            Gi[:, s] = 1.0
        elif s == 16:
            Gi[:, s] = (-1.0) ** n
        elif s < 16:
            Gi[:, s] = 2 * np.cos(2 * np.pi * s * n / T)
        else:
            Gi[:, s] = -2 * np.sin(2 * np.pi * (s - 16) * n / T)
    return Gi / T


def Fc():
    a = np.arange(T)
    th = 2 * np.pi * np.outer(a, a) / T
    C, S = np.cos(th), np.sin(th)
    return np.block([[C, S], [-S, C]])          # [out_r; out_i] = Fc [in_r; in_i]  (forward, e^{-i theta})


def forward_tile(xw):
    """xw (T, T, C) -> spectrum rows (1024, C)."""
    U = np.einsum('sx,yxc->ysc', G_fwd(), xw)
    out = np.zeros((T * T, xw.shape[2]))
    out[0:32] = np.einsum('sy,yc->sc', G_fwd(), U[:, 0])
    out[32:64] = np.einsum('sy,yc->sc', G_fwd(), U[:, 16])
    F = Fc()
    for fx in range(1, 16):
        out[64 + 64 * (fx - 1):64 + 64 * fx] = F @ np.concatenate([U[:, fx], U[:, 16 + fx]], 0)
    return out


def inverse_tile(sp):
    """spectrum rows (1024, C) -> window (T, T, C)."""
    C = sp.shape[1]
    U = np.zeros((T, T, C))
    U[:, 0] = G_inv() @ sp[0:32]
    U[:, 16] = G_inv() @ sp[32:64]
    Fi = Fc().T / T
    for fx in range(1, 16):
        u = Fi @ sp[64 + 64 * (fx - 1):64 + 64 * fx]
        U[:, fx], U[:, 16 + fx] = u[:32], u[32:]
    return np.einsum('xs,ysc->yxc', G_inv(), U)


def slots():
    """The 512 mixing slots: (row of the real part, row of the imaginary part, kind, fy, fx).  kind 0: complex frequency (fy, fx);
    kind 1: two real frequencies packed as one slot: 're' row = (0, fx), 'im' row = (16, fx)."""
    sl = []
    for base, fx in ((0, 0), (32, 16)):
        sl.append((base, base + 16, 1, 0, fx))
        for fy in range(1, 16):
            sl.append((base + fy, base + 16 + fy, 0, fy, fx))
    for fx in range(1, 16):
        for fy in range(32):
            sl.append((64 + 64 * (fx - 1) + fy, 64 + 64 * (fx - 1) + 32 + fy, 0, fy, fx))
    return sl


def weight_spectrum(w):
    """w (kh, kw, Ci, Co) -> complex W[fy][fx] (T, T, Ci, Co) of the zero-padded kernel (numpy FFT: the model's ground truth), and the
    same through forward_tile (what the GPU does: the filter is transformed as a 1-tile image with Ci*Co channels)."""
    kh, kw, Ci, Co = w.shape
    wp = np.zeros((T, T, Ci * Co))
    wp[:kh, :kw] = w.reshape(kh, kw, Ci * Co)
    return forward_tile(wp)                       # (1024, Ci*Co) rows


def mix_matrices(wsp, Ci, Co, conj=True):
    """From the filter's spectrum rows build the 512 real (2Ci x 2Co) matrices M with [Yr | Yi] = [Xr | Xi] M  (H = conj(W): correlation)."""
    Ms = []
    for (rr, ri, kind, fy, fx) in slots():
        Wr, Wi = wsp[rr].reshape(Ci, Co), wsp[ri].reshape(Ci, Co)
        if kind == 1:                              # rows rr / ri hold two REAL frequencies
            M = np.block([[Wr, np.zeros((Ci, Co))], [np.zeros((Ci, Co)), Wi]])
        else:
            Hr, Hi = Wr, (-Wi if conj else Wi)
            M = np.block([[Hr, Hi], [-Hi, Hr]])
        Ms.append(M)
    return Ms


def conv_forward(x, w, pad_top, pad_left, Ho, Wo):
    """y[o] = sum_t xpad[o + t - pad] w[t] (zero padding) through the tiled spectral path.  x (H, W, Ci), w (kh, kw, Ci, Co)."""
    H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    Vy, Vx = T - kh + 1, T - kw + 1
    Ms = mix_matrices(weight_spectrum(w), Ci, Co)
    sl = slots()
    y = np.zeros((Ho, Wo, Co))
    for y0 in range(0, Ho, Vy):
        for x0 in range(0, Wo, Vx):
            xw = np.zeros((T, T, Ci))
            for r in range(T):
                for c in range(T):
                    sy, sx = y0 - pad_top + r, x0 - pad_left + c
                    if 0 <= sy < H and 0 <= sx < W:
                        xw[r, c] = x[sy, sx]
            X = forward_tile(xw)
            Y = np.zeros((T * T, Co))
            for (rr, ri, kind, fy, fx), M in zip(sl, Ms):
                o = np.concatenate([X[rr], X[ri]]) @ M
                Y[rr], Y[ri] = o[:Co], o[Co:]
            yw = inverse_tile(Y)
            vy, vx = min(Vy, Ho - y0), min(Vx, Wo - x0)
            y[y0:y0 + vy, x0:x0 + vx] = yw[:vy, :vx]
    return y


def conv_wgrad(x, dz, kh, kw, pad_top, pad_left):
    """dw[t] = sum_o xpad[o + t - pad] dz[o]."""
    H, W, Ci = x.shape
    Ho, Wo, Co = dz.shape
    Vy, Vx = T - kh + 1, T - kw + 1
    sl = slots()
    Csp = np.zeros((T * T, Ci * Co))
    for y0 in range(0, Ho, Vy):
        for x0 in range(0, Wo, Vx):
            xw = np.zeros((T, T, Ci))
            dw_ = np.zeros((T, T, Co))
            for r in range(T):
                for c in range(T):
                    sy, sx = y0 - pad_top + r, x0 - pad_left + c
                    if 0 <= sy < H and 0 <= sx < W:
                        xw[r, c] = x[sy, sx]
                    if r < Vy and c < Vx and y0 + r < Ho and x0 + c < Wo:
                        dw_[r, c] = dz[y0 + r, x0 + c]
            X, D = forward_tile(xw), forward_tile(dw_)
            for (rr, ri, kind, fy, fx) in sl:
                P = np.outer(np.concatenate([X[rr], X[ri]]), np.concatenate([D[rr], D[ri]]))   # (2Ci, 2Co)
                P11, P12, P21, P22 = P[:Ci, :Co], P[:Ci, Co:], P[Ci:, :Co], P[Ci:, Co:]
                if kind == 1:
                    Csp[rr] += P11.reshape(-1)
                    Csp[ri] += P22.reshape(-1)
                else:
                    Csp[rr] += (P11 + P22).reshape(-1)
                    Csp[ri] += (P21 - P12).reshape(-1)
    c = inverse_tile(Csp)
    return c[:kh, :kw].reshape(kh, kw, Ci, Co)


def direct(x, w, pad_top, pad_left, Ho, Wo):
    H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    xp = np.zeros((Ho + kh - 1 + 64, Wo + kw - 1 + 64, Ci))
    xp[pad_top:pad_top + H, pad_left:pad_left + W] = x
    y = np.zeros((Ho, Wo, Co))
    for i in range(kh):
        for j in range(kw):
            y += np.einsum('hwc,co->hwo', xp[i:i + Ho, j:j + Wo], w[i, j])
    return y


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    xw = rng.standard_normal((T, T, 3))
    sp = forward_tile(xw)
    ref = np.fft.fft2(xw, axes=(0, 1))
    assert np.allclose(sp[0:17], ref[0:17, 0].real) and np.allclose(sp[17:32], ref[1:16, 0].imag)
    assert np.allclose(sp[64 + 64 * 4:64 + 64 * 4 + 32], ref[:, 5].real) and np.allclose(sp[64 + 64 * 4 + 32:64 + 64 * 5], ref[:, 5].imag)
    assert np.allclose(inverse_tile(sp), xw)
    H, W, Ci, Co, k = 45, 50, 3, 2, 7
    x = rng.standard_normal((H, W, Ci))
    w = rng.standard_normal((k, k, Ci, Co))
    y = conv_forward(x, w, 3, 3, H, W)
    yr = direct(x, w, 3, 3, H, W)
    print('forward rel err', np.linalg.norm(y - yr) / np.linalg.norm(yr))
    dz = rng.standard_normal((H, W, Co))
    dw = conv_wgrad(x, dz, k, k, 3, 3)
    xp = np.zeros((H + k - 1, W + k - 1, Ci)); xp[3:3 + H, 3:3 + W] = x
    dwr = np.stack([np.stack([np.einsum('hwc,hwo->co', xp[i:i + H, j:j + W], dz) for j in range(k)]) for i in range(k)])
    print('wgrad rel err', np.linalg.norm(dw - dwr) / np.linalg.norm(dwr))
    print('slots', len(slots()))
