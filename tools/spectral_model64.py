"""Numpy model of the 64-point tiled spectral convolution (poisson_cnn_amd/csrc/spectral64.hip): the spectrum-row conventions are those
of tools/spectral_model.py scaled to T = 64, the transforms are decomposed exactly as the kernels decompose them (one radix-2 step per
axis, the remaining 32-point transforms as real matrix products = MFMA GEMMs), and every table the kernels hold is built here by the same
formulas as build_tables64() and checked against numpy's FFT.  Developer tool - not imported by the product or the tests.

Spectrum rows R (T*T = 4096 per tile, each a channel row), T2 = T/2 = 32:
  R in [0, T)        column fx = 0   (real in x): half-complex in y - rows 0..T2: Re(fy), rows T2+1..T-1: Im(fy = R - T2)
  R in [T, 2T)       column fx = T2  (real in x): the same
  2T + 2T (fx - 1) + {fy | T + fy}   complex column fx = 1..T2-1: Re | Im of fy = 0..T-1
Mixing slots: T*T/2 (tools/spectral_model.py slots() with T = 64).

Forward kernel (spec64_fwd_kernel), one item = (tile, 32 channels, x parity px):
  u[y][x] = w[y][x] + w[y][x+32] (px = 0) or w[y][x] - w[y][x+32] (px = 1), x < 32           (vector ALU, in-lane)
  D[y][rho] = sum_x GX[px][rho][x] u[y][x], rho = 16 part + j: pair j of this parity           (16 MFMAs 32x32x2 per window row)
     px = 0: pair 0 = the two real columns (part 0: fx = 0, part 1: fx = 32), pair j >= 1: fx = 2j (part 0: Re, part 1: Im)
     px = 1: pair j: fx = 2j + 1
  y axis, wave parity h (output fy = 2m + h), v[y] = D[y] + D[y+32] (h = 0) or D[y] - D[y+32] (h = 1), y < 32:
     complex pair:  Zr[m] = sum_y CY[h][m][y] vr[y] + SY[h][m][y] vi[y],   Zi[m] = sum_y CY[h][m][y] vi[y] - SY[h][m][y] vr[y]
     real pair:     H[rho] = sum_y RY[h][rho][y] v[y]  for each of the two real columns (half-complex rows, see ry_row())
Inverse kernel (spec64_inv_kernel), one item = (tile, 16 channels): the mirror image (decimation in time), see inverse_tile_kernel().
"""
import numpy as np

T, T2 = 64, 32
TP = 2 * np.pi / T


# ----------------------------------------------------------------------------------------------------------------- conventions (any T)
def G_fwd(t=T):
    x = np.arange(t)
    G = np.zeros((t, t))
    for s in range(t):
        G[s] = np.cos(2 * np.pi * s * x / t) if s <= t // 2 else -np.sin(2 * np.pi * (s - t // 2) * x / t)
    return G


def G_inv(t=T):
    n = np.arange(t)
    Gi = np.zeros((t, t))
    for s in range(t):
        if s == 0:
            Gi[:, s] = 1.0
        elif s == t // 2:
            Gi[:, s] = (-1.0) ** n
        elif s < t // 2:
            Gi[:, s] = 2 * np.cos(2 * np.pi * s * n / t)
        else:
            Gi[:, s] = -2 * np.sin(2 * np.pi * (s - t // 2) * n / t)
    return Gi / t


def Fc(t=T):
    a = np.arange(t)
    th = 2 * np.pi * np.outer(a, a) / t
    C, S = np.cos(th), np.sin(th)
    return np.block([[C, S], [-S, C]])


def forward_tile(xw):
    """Reference: xw (T, T, C) -> spectrum rows (T*T, C), direct matrices."""
    t = xw.shape[0]
    h = t // 2
    U = np.einsum('sx,yxc->ysc', G_fwd(t), xw)
    out = np.zeros((t * t, xw.shape[2]))
    out[0:t] = G_fwd(t) @ U[:, 0]
    out[t:2 * t] = G_fwd(t) @ U[:, h]
    F = Fc(t)
    for fx in range(1, h):
        out[2 * t + 2 * t * (fx - 1):2 * t + 2 * t * fx] = F @ np.concatenate([U[:, fx], U[:, h + fx]], 0)
    return out


def inverse_tile(sp):
    t = int(round(np.sqrt(sp.shape[0])))
    h = t // 2
    C = sp.shape[1]
    U = np.zeros((t, t, C))
    U[:, 0] = G_inv(t) @ sp[0:t]
    U[:, h] = G_inv(t) @ sp[t:2 * t]
    Fi = Fc(t).T / t
    for fx in range(1, h):
        u = Fi @ sp[2 * t + 2 * t * (fx - 1):2 * t + 2 * t * fx]
        U[:, fx], U[:, h + fx] = u[:t], u[t:]
    return np.einsum('xs,ysc->yxc', G_inv(t), U)


def slots(t=T):
    h = t // 2
    sl = []
    for base, fx in ((0, 0), (t, h)):
        sl.append((base, base + h, 1, 0, fx))
        for fy in range(1, h):
            sl.append((base + fy, base + h + fy, 0, fy, fx))
    for fx in range(1, h):
        for fy in range(t):
            sl.append((2 * t + 2 * t * (fx - 1) + fy, 2 * t + 2 * t * (fx - 1) + t + fy, 0, fy, fx))
    return sl


# ----------------------------------------------------------------------------------------------------------------- kernel tables (T = 64)
def pair_fx(px, j):
    """x frequency of pair j of parity px; pair (0, 0) is the real pseudo-pair (fx = 0 | 32)."""
    return 2 * j + px


def GX(px):
    """[rho = 16 part + j][x < 32]: x-axis real -> half-complex transform of u = w[x] +- w[x + 32]."""
    x = np.arange(T2)
    g = np.zeros((T2, T2))
    for part in range(2):
        for j in range(16):
            fx = pair_fx(px, j)
            if px == 0 and j == 0:
                g[16 * part + j] = np.ones(T2) if part == 0 else (-1.0) ** x            # Re X[0], Re X[32]
            else:
                g[16 * part + j] = np.cos(TP * fx * x) if part == 0 else -np.sin(TP * fx * x)
    return g


def CY(h):
    m, y = np.arange(T2)[:, None], np.arange(T2)[None, :]
    return np.cos(TP * (2 * m + h) * y)


def SY(h):
    m, y = np.arange(T2)[:, None], np.arange(T2)[None, :]
    return np.sin(TP * (2 * m + h) * y)


def ry_freq(h, rho):
    """(fy, is_imag) of output row rho of the real-column transform of parity h: h = 0: rho <= 16: Re fy = 2 rho; rho >= 17: Im fy = 2 (rho - 16);
    h = 1: rho < 16: Re fy = 2 rho + 1; rho >= 16: Im fy = 2 (rho - 16) + 1."""
    if h == 0:
        return (2 * rho, 0) if rho <= 16 else (2 * (rho - 16), 1)
    return (2 * rho + 1, 0) if rho < 16 else (2 * (rho - 16) + 1, 1)


def ry_row(h, rho):
    """spectrum row (inside a real column's 64 half-complex rows) of output row rho."""
    fy, im = ry_freq(h, rho)
    return T2 + fy if im else fy


def RY(h):
    y = np.arange(T2)
    r = np.zeros((T2, T2))
    for rho in range(T2):
        fy, im = ry_freq(h, rho)
        r[rho] = -np.sin(TP * fy * y) if im else np.cos(TP * fy * y)
    return r


def forward_tile_kernel(xw):
    """The forward kernel's algebra: xw (64, 64, C) -> rows (4096, C)."""
    C = xw.shape[2]
    out = np.zeros((T * T, C))
    for px in range(2):
        u = xw[:, :T2] + (xw[:, T2:] if px == 0 else -xw[:, T2:])                  # (64, 32, C)
        D = np.einsum('rx,yxc->yrc', GX(px), u)                                     # (64, 32, C): the stage buffer over all stages
        for h in range(2):
            v = D[:T2] + (D[T2:] if h == 0 else -D[T2:])                            # (32, 32, C)
            for j in range(16):
                vr, vi = v[:, j], v[:, 16 + j]
                if px == 0 and j == 0:
                    for col, vv in ((0, vr), (1, vi)):                              # the two real columns: fx = 0 -> rows [0, 64), fx = 32 -> [64, 128)
                        H = RY(h) @ vv
                        for rho in range(T2):
                            out[T * col + ry_row(h, rho)] = H[rho]
                    continue
                Zr = CY(h) @ vr + SY(h) @ vi
                Zi = CY(h) @ vi - SY(h) @ vr
                base = 2 * T + 2 * T * (pair_fx(px, j) - 1)
                for m in range(T2):
                    out[base + 2 * m + h] = Zr[m]
                    out[base + T + 2 * m + h] = Zi[m]
    return out


# inverse kernel tables -------------------------------------------------------------------------------------------------------------
def CIY(h):
    """[y < 32][m]: E (h = 0) / O (h = 1) of the decimation-in-time inverse along y: e^{+i theta}, theta = TP (2m + h) y, scaled 1 / T."""
    y, m = np.arange(T2)[:, None], np.arange(T2)[None, :]
    return np.cos(TP * (2 * m + h) * y) / T


def SIY(h):
    y, m = np.arange(T2)[:, None], np.arange(T2)[None, :]
    return np.sin(TP * (2 * m + h) * y) / T


def RIY(h):
    """[y < 32][rho]: real-column inverse of the half-complex entries of parity h (K order rho as ry_freq)."""
    y = np.arange(T2)
    r = np.zeros((T2, T2))
    for rho in range(T2):
        fy, im = ry_freq(h, rho)
        w = 1.0 if (fy == 0 or fy == T2) else 2.0
        r[:, rho] = (-w * np.sin(TP * fy * y) if im else w * np.cos(TP * fy * y)) / T
    return r


def GIX(px):
    """[x < 32][rho = 16 part + j]: x-axis half-complex -> real inverse, the E (px = 0) / O (px = 1) halves: out[x] = E + O, out[x+32] = E - O."""
    x = np.arange(T2)
    g = np.zeros((T2, T2))
    for part in range(2):
        for j in range(16):
            fx = pair_fx(px, j)
            if px == 0 and j == 0:
                g[:, 16 * part + j] = (np.ones(T2) if part == 0 else (-1.0) ** x) / T
            else:
                g[:, 16 * part + j] = (2 * np.cos(TP * fx * x) if part == 0 else -2 * np.sin(TP * fx * x)) / T
    return g


def inverse_tile_kernel(sp):
    """The inverse kernel's algebra: rows (4096, C) -> window (64, 64, C)."""
    C = sp.shape[1]
    EO = [np.zeros((T, T2, C)), np.zeros((T, T2, C))]                             # E_x / O_x [y][x < 32]
    for px in range(2):
        for j in range(16):
            acc = [None, None]                                                       # E / O of the y axis, each (Dr, Di) on y < 32
            for h in range(2):
                if px == 0 and j == 0:
                    ha = np.stack([sp[ry_row(h, rho)] for rho in range(T2)])        # column fx = 0
                    hb = np.stack([sp[T + ry_row(h, rho)] for rho in range(T2)])    # column fx = 32
                    acc[h] = (RIY(h) @ ha, RIY(h) @ hb)
                else:
                    base = 2 * T + 2 * T * (pair_fx(px, j) - 1)
                    Zr = np.stack([sp[base + 2 * m + h] for m in range(T2)])
                    Zi = np.stack([sp[base + T + 2 * m + h] for m in range(T2)])
                    acc[h] = (CIY(h) @ Zr - SIY(h) @ Zi, SIY(h) @ Zr + CIY(h) @ Zi)
            Dr = np.concatenate([acc[0][0] + acc[1][0], acc[0][0] - acc[1][0]])      # y < 32: E + O; y >= 32: E - O
            Di = np.concatenate([acc[0][1] + acc[1][1], acc[0][1] - acc[1][1]])
            g = GIX(px)
            EO[px] += np.einsum('x,yc->yxc', g[:, j], Dr) + np.einsum('x,yc->yxc', g[:, 16 + j], Di)
    return np.concatenate([EO[0] + EO[1], EO[0] - EO[1]], axis=1)


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    xw = rng.standard_normal((T, T, 3))
    sp = forward_tile(xw)
    ref = np.fft.fft2(xw, axes=(0, 1))
    assert np.allclose(sp[0:T2 + 1], ref[0:T2 + 1, 0].real) and np.allclose(sp[T2 + 1:T], ref[1:T2, 0].imag)
    assert np.allclose(sp[T:T + T2 + 1], ref[0:T2 + 1, T2].real) and np.allclose(sp[T + T2 + 1:2 * T], ref[1:T2, T2].imag)
    assert np.allclose(sp[2 * T + 2 * T * 4:2 * T + 2 * T * 4 + T], ref[:, 5].real) and np.allclose(sp[2 * T + 2 * T * 4 + T:2 * T + 2 * T * 5], ref[:, 5].imag)
    assert np.allclose(inverse_tile(sp), xw)
    spk = forward_tile_kernel(xw)
    print('forward kernel algebra vs matrices', np.abs(spk - sp).max())
    assert np.allclose(spk, sp)
    xk = inverse_tile_kernel(sp)
    print('inverse kernel algebra vs window  ', np.abs(xk - xw).max())
    assert np.allclose(xk, xw)
    print('slots', len(slots()))
