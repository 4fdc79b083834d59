// In-register FFTs for the tiled spectral convolution (spectral_fft.hip): every routine works IN PLACE on a small array that the caller keeps in
// vector registers (lane = channel: one lane holds a whole row or column of ITS channel, so no butterfly ever crosses lanes).  All loops have
// compile-time bounds and are fully unrolled; every index and every twiddle factor is a compile-time constant after unrolling, so the arrays
// never leave the register file.  fp32 MFMA has no rate advantage over the vector ALUs on gfx950 (both 64 FLOP/clk/SIMD): the DFT-as-GEMM of
// spectral_conv.hip spends T multiply-adds per point where these spend ~2.5 log2 T.
//
// The header is also compiled by the host compiler (tests/native/test_fft_regs.cpp, no GPU needed): the same source is checked against a
// double-precision DFT there.
#pragma once
#if defined(__HIPCC__)
#define FFT_HD __host__ __device__ __forceinline__
#else
#define FFT_HD inline __attribute__((always_inline))
#endif

namespace pcnn_fft {

// cos / sin of 2 pi k / 64
FFT_HD constexpr float cos64(int k) {
  constexpr float t[64] = {1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f, 0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f, 0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f, 0.0f, -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f, -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f, -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f, -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f, -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f, -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f, 0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f, 0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f, 0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f};
  return t[k & 63];
}
FFT_HD constexpr float sin64(int k) {
  constexpr float t[64] = {0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f, 0.3826834261417389f, 0.4713967442512512f, 0.5555702447891235f, 0.6343932747840881f, 0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f, 0.8819212913513184f, 0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f, 1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f, 0.9238795042037964f, 0.8819212913513184f, 0.8314695954322815f, 0.7730104327201843f, 0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f, 0.4713967442512512f, 0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f, 0.0f, -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f, -0.3826834261417389f, -0.4713967442512512f, -0.5555702447891235f, -0.6343932747840881f, -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f, -0.8819212913513184f, -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f, -1.0f, -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f, -0.9238795042037964f, -0.8819212913513184f, -0.8314695954322815f, -0.7730104327201843f, -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f, -0.4713967442512512f, -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f};
  return t[k & 63];
}
// W_N^k = exp(-2 pi i k / N) for N | 64
template <int N> FFT_HD constexpr float tw_re(int k) { return cos64(k * (64 / N)); }
template <int N> FFT_HD constexpr float tw_im(int k) { return -sin64(k * (64 / N)); }

FFT_HD constexpr int bitrev(int k, int N) {
  int r = 0;
  for (int b = 1; b < N; b <<= 1) { r = (r << 1) | (k & 1); k >>= 1; }
  return r;
}

FFT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// (dr + i di) * W, W = W_N^j (SIGN < 0) or its conjugate (SIGN > 0); j in (0, N/2).  The special angles cost adds only / two multiplies.
template <int N, int SIGN> FFT_HD void mul_tw(int j, float dr, float di, float& outr, float& outi) {
  constexpr float R2 = 0.70710678118654752440f;
  if (j == 0) { outr = dr; outi = di; }
  else if (4 * j == N) { if (SIGN < 0) { outr = di; outi = -dr; } else { outr = -di; outi = dr; } }          // -i | +i
  else if (8 * j == N) {                                                                                        // (1 -+ i) / sqrt 2
    if (SIGN < 0) { outr = (dr + di) * R2; outi = (di - dr) * R2; } else { outr = (dr - di) * R2; outi = (di + dr) * R2; }
  } else if (8 * j == 3 * N) {                                                                                  // (-1 -+ i) / sqrt 2
    if (SIGN < 0) { outr = (di - dr) * R2; outi = -(dr + di) * R2; } else { outr = -(dr + di) * R2; outi = (dr - di) * R2; }
  } else {
    const float wr = tw_re<N>(j), wi = SIGN < 0 ? tw_im<N>(j) : -tw_im<N>(j);
    outr = fma_(dr, wr, -(di * wi));
    outi = fma_(dr, wi, di * wr);
  }
}

// Complex FFT, decimation in frequency, in place: natural-order input, output X[k] at index bitrev(k, N).  SIGN < 0: forward (e^{-i}), SIGN > 0:
// the unnormalised inverse (e^{+i}, N times the inverse transform).
template <int N, int SIGN> FFT_HD void cfft_dif(float* re, float* im) {
  if constexpr (N == 2) {
    const float ar = re[0], ai = im[0], br = re[1], bi = im[1];
    re[0] = ar + br; im[0] = ai + bi; re[1] = ar - br; im[1] = ai - bi;
  } else if constexpr (N > 2) {
    constexpr int H = N / 2;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const float ar = re[j], ai = im[j], br = re[j + H], bi = im[j + H];
      re[j] = ar + br; im[j] = ai + bi;
      mul_tw<N, SIGN>(j, ar - br, ai - bi, re[j + H], im[j + H]);
    }
    cfft_dif<H, SIGN>(re, im);
    cfft_dif<H, SIGN>(re + H, im + H);
  }
}

// The exact inverse flow graph of cfft_dif<N, -1>, unnormalised: input X[k] at index bitrev(k, N), output N x[n] in natural order.
template <int N> FFT_HD void cfft_undo_dif(float* re, float* im) {
  if constexpr (N == 2) {
    const float ar = re[0], ai = im[0], br = re[1], bi = im[1];
    re[0] = ar + br; im[0] = ai + bi; re[1] = ar - br; im[1] = ai - bi;
  } else if constexpr (N > 2) {
    constexpr int H = N / 2;
    cfft_undo_dif<H>(re, im);
    cfft_undo_dif<H>(re + H, im + H);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      float tr, ti;
      mul_tw<N, +1>(j, re[j + H], im[j + H], tr, ti);
      const float ar = re[j], ai = im[j];
      re[j] = ar + tr; im[j] = ai + ti; re[j + H] = ar - tr; im[j + H] = ai - ti;
    }
  }
}

// Real FFT of N points, in place, by the real split radix recursion: e[n] = x[n] + x[n + N/2] carries the even bins (a real FFT of N/2 points),
// the odd bins come from ONE complex FFT of N/4 points: X[4m + 1] = FFT_{N/4}( (d[n] - i d[n + N/4]) W_N^n )[m], d[n] = x[n] - x[n + N/2], and
// X[4m + 3] = conj(X[N - 4m - 3]).  No untangling pass, ~2.5 N log2 N / 2 flops.  Where bin f lands: rfft_pos.
template <int N> FFT_HD void rfft_fwd(float* x) {
  if constexpr (N == 2) {
    const float a = x[0], b = x[1];
    x[0] = a + b; x[1] = a - b;
  } else if constexpr (N > 2) {
    constexpr int H = N / 2, Q = N / 4;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      const float lo = x[n], hi = x[n + H];
      x[n] = lo + hi;
      x[n + H] = n < Q ? lo - hi : hi - lo;          // d[n] for n < N/4, -d[n] beyond: c[n] = x[H + n] + i x[H + Q + n]
    }
#pragma unroll
    for (int n = 1; n < Q; ++n) mul_tw<N, -1>(n, x[H + n], x[H + Q + n], x[H + n], x[H + Q + n]);
    if constexpr (Q > 1) cfft_dif<Q, -1>(x + H, x + H + Q);
    rfft_fwd<H>(x);
  }
}

// Index of Re X[f] (imag = false) or Im X[f] (imag = true) in the array rfft_fwd<N> leaves behind, 0 <= f <= N/2; `neg` is set when the stored
// value is the NEGATED imaginary part (bins f = 3 mod 4 are held as their conjugate partners).  Im X[0] and Im X[N/2] do not exist.
FFT_HD constexpr int rfft_pos(int N, int f, bool imag, bool& neg) {
  neg = false;
  while (true) {
    if (N == 1) return 0;
    if (N == 2) return f;                                   // X[0], X[1]: both real
    if ((f & 1) == 0) { N >>= 1; f >>= 1; continue; }        // even bin: bin f/2 of the half-size transform, held in the first half
    const int H = N / 2, Q = N / 4;
    const bool conj = (f & 3) == 3;
    const int m = conj ? (N - f - 1) / 4 : (f - 1) / 4;
    neg = conj && imag;
    // the odd branch sits at offset H of ITS level's array; levels nest in the first half, so offsets simply add up (all zero)
    return H + (imag ? Q : 0) + bitrev(m, Q);
  }
}

// The exact inverse flow graph of rfft_fwd<N>, unnormalised: input laid out as rfft_fwd leaves it (rfft_pos), output N x[n] in natural order.
template <int N> FFT_HD void rfft_inv(float* x) {
  if constexpr (N == 2) {
    const float a = x[0], b = x[1];
    x[0] = a + b; x[1] = a - b;
  } else if constexpr (N > 2) {
    constexpr int H = N / 2, Q = N / 4;
    rfft_inv<H>(x);                                          // x[0..H) = H e[n]
    if constexpr (Q > 1) cfft_undo_dif<Q>(x + H, x + H + Q);  // = Q c[n]
    // undo the twiddle and bring the odd branch to the even branch's scale: H d = 2 Q c conj(W)
#pragma unroll
    for (int n = 0; n < Q; ++n) {
      float tr, ti;
      mul_tw<N, +1>(n, x[H + n], x[H + Q + n], tr, ti);
      x[H + n] = tr + tr; x[H + Q + n] = ti + ti;
    }
#pragma unroll
    for (int n = 0; n < H; ++n) {
      const float e = x[n], d = x[n + H];
      x[n] = n < Q ? e + d : e - d;
      x[n + H] = n < Q ? e - d : e + d;
    }
  }
}

}  // namespace pcnn_fft
