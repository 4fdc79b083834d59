"""Compares the two forms of the 64-point forward transform (PCNN_FWD64_RADIX = 2: round 3's parity form, 4: the second radix-2 step) row by row on
one random 64 x 64 x 32 window, against numpy's FFT for the rows whose meaning is stated in spectral_common.h."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops
from poisson_cnn_amd.ops import _p
g = torch.Generator(device='cuda').manual_seed(0)
C = 32
x = torch.randn(64, 64, C, device='cuda', generator=g)
def spec(radix, ylim=64, xlim=64):
    os.environ['PCNN_FWD64_RADIX'] = str(radix)
    out = torch.zeros(4096, 32, device='cuda')
    ops.handle().call('pcnn_debug_tile_spectrum64', ctypes.c_int(64), ctypes.c_int(64), ctypes.c_int(C), _p(x), ctypes.c_int(ylim), ctypes.c_int(xlim), _p(out))
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64)
a, b = spec(2), spec(4)
X = np.fft.fft2(x.cpu().numpy().astype(np.float64), axes=(0, 1))        # X[fy][fx][c]
err = np.abs(a - b).max(axis=1) / np.abs(a).max()
bad = np.nonzero(err > 1e-5)[0]
print('rows differing between radix 2 and 4: %d of 4096' % len(bad))
def where(row):
    if row < 64: return 'fx=0 ' + ('Re fy=%d' % row if row <= 32 else 'Im fy=%d' % (row - 32))
    if row < 128: r = row - 64; return 'fx=32 ' + ('Re fy=%d' % r if r <= 32 else 'Im fy=%d' % (r - 32))
    fx = 1 + (row - 128) // 128; r = (row - 128) % 128
    return 'fx=%d %s fy=%d' % (fx, 'Re' if r < 64 else 'Im', r % 64)
def ref(row):
    if row < 128:
        fx = 0 if row < 64 else 32; r = row % 64
        return X[r, fx, 0].real if r <= 32 else X[r - 32, fx, 0].imag
    fx = 1 + (row - 128) // 128; r = (row - 128) % 128
    return X[r % 64, fx, 0].real if r < 64 else X[r % 64, fx, 0].imag
for r in bad[:12]:
    print(r, where(r), 'radix2 %.4f radix4 %.4f numpy %.4f' % (a[r, 0], b[r, 0], ref(r)))
# decompose: what does radix 4 produce in terms of the classes?  W_re / W_im contributions for the real column fx = 0, fy = 1
R = x.cpu().numpy().astype(np.float64)[:, :, 0].sum(axis=1)        # x-frequency 0 of channel 0: sum over x of row y
y = np.arange(16)
for h2 in (1, 3):
    D = [R[y + 16 * q] for q in range(4)]
    Wre = D[0] - D[2]; Wim = (-D[1] + D[3]) if h2 == 1 else (D[1] - D[3])
    for fy in (h2, 4 + h2):
        th = 2 * np.pi * fy * y / 64
        print('class %d fy %d: Re %.4f = a %.4f + b %.4f ; Im %.4f' % (h2, fy, (np.cos(th) * Wre + np.sin(th) * Wim).sum(), (np.cos(th) * Wre).sum(), (np.sin(th) * Wim).sum(), (np.cos(th) * Wim - np.sin(th) * Wre).sum()))
# radix 2 against numpy (sanity of the row map)
fx, fy = 5, 9
print('check map: row', 128 + 128 * (fx - 1) + fy, a[128 + 128 * (fx - 1) + fy, 0], X[fy, fx, 0].real, a[128 + 128 * (fx - 1) + 64 + fy, 0], X[fy, fx, 0].imag)

print('kernel radix4 fy=1 Re: %.4f, fy=5: %.4f, fy=3: %.4f' % (b[1, 0], b[5, 0], b[3, 0]))
D = [R[y + 16 * q] for q in range(4)]
import itertools
for fy, h2 in ((1, 1), (5, 1), (3, 3)):
    th = 2 * np.pi * fy * y / 64
    best = []
    for sa in itertools.product((-1, 0, 1), repeat=4):
        for sb in itertools.product((-1, 0, 1), repeat=4):
            Wre = sum(c * d for c, d in zip(sa, D)); Wim = sum(c * d for c, d in zip(sb, D))
            for name, val in (('cos*Wa+sin*Wb', (np.cos(th) * Wre + np.sin(th) * Wim).sum()), ('cos*Wa-sin*Wb', (np.cos(th) * Wre - np.sin(th) * Wim).sum())):
                if abs(val - b[fy, 0]) < 2e-3 * max(1, abs(b[fy, 0])):
                    best.append((sa, sb, name))
    print('fy', fy, 'matches:', best[:6])

bnob = spec(4, ylim=1064)
for fy, h2 in ((1, 1), (5, 1), (3, 3), (7, 3)):
    th = 2 * np.pi * fy * y / 64
    Wre = D[0] - D[2]; Wim = (-D[1] + D[3]) if h2 == 1 else (D[1] - D[3])
    print('fy %d: kernel without pass b %.4f (expected a = %.4f); kernel full %.4f; difference full - nob %.4f (expected b = %.4f)' % (fy, bnob[fy, 0], (np.cos(th) * Wre).sum(), b[fy, 0], b[fy, 0] - bnob[fy, 0], (np.sin(th) * Wim).sum()))

print('--- search for what pass b computes')
for fy, h2 in ((1, 1), (5, 1), (3, 3), (7, 3)):
    target = b[fy, 0] - bnob[fy, 0]
    found = []
    for sbc in itertools.product((-1, 0, 1), repeat=4):
        Wim = sum(c_ * d for c_, d in zip(sbc, D))
        for f2 in range(64):
            th = 2 * np.pi * f2 * y / 64
            for nm, tr in (('sin', np.sin(th)), ('cos', np.cos(th))):
                v = (tr * Wim).sum()
                if abs(v - target) < 1e-3 * max(1, abs(target)): found.append((sbc, f2, nm))
    print(fy, target, found[:5])
# also: all 16 partial products? maybe y index shifted: sum over y with table row y' = y + s
