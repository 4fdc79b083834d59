// Per-sample-filter ("metalearning") convolutions: every sample of the batch is convolved with ITS OWN filter, emitted by a hyper-network
// (layers/metalearning_conv.py:148-169: tf.map_fn over the batch of one tf.nn.conv2d per sample; layers/metalearning_deconvupscale.py:104-137
// the same for conv2d_transpose).  The reference serialises the samples (map_fn); here ONE launch covers the batch: the sample index is a
// grid dimension and the filter / bias pointers advance by a per-sample stride, so the launch count of a metalearning layer does not depend
// on N.  The layers have few channels (<= 32) and filters of any size up to 31 taps (the reference example uses 19 x 19 x 3 x 4), 1-D layers are
// kh = 1: exact fp32 FMAs on the vector ALUs over LDS-staged tiles (the filters of a workgroup's sample are wave-uniform: scalar loads).
//   grouped_fwd_kernel      tf.pad + conv2d(VALID) + bias + activation; with flip = 1 the filter is read flipped and transposed, which
//                           makes the same kernel the data gradient
//   grouped_wgrad_kernel    per-sample filter gradient, deterministic (row strips reduced in a fixed order)
//   grouped_deconv_*        conv2d_transpose with kernel = stride: forward, data gradient, filter + bias gradient
#include "pcnn_internal.h"
#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int GTH = 8, GTW = 32;          // output tile of the forward kernel: 8 rows x 32 columns, one thread per pixel

struct GroupedParams {
  const float* x; const float* w; const float* bias; float* y;
  long long w_stride, b_stride;           // floats between the filters / biases of consecutive samples
  int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, kh, kw, pt, pl, pad_mode; float pad_value; int act; float alpha;
  int flip;                               // 1: w'[i][j][ci][co] = w[kh-1-i][kw-1-j][co][ci] (the stored filter is (kh, kw, Cout, Cin): data gradient)
  int tiles_x, tiles_y;
};

template <int CO>
__global__ __launch_bounds__(256) void grouped_fwd_kernel(GroupedParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];                  // [(GTH + kh - 1)][(GTW + kw - 1)][4 input channels]
  const int TR = GTH + p.kh - 1, TC = GTW + p.kw - 1;
  const int n = blockIdx.y;
  const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x % p.tiles_x;
  const int y0 = ty * GTH, x0 = tx * GTW;
  const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
  const float* w = p.w + (int64_t)n * p.w_stride;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  for (int ci0 = 0; ci0 < p.Cin; ci0 += 4) {
    for (int u = threadIdx.x; u < TR * TC; u += 256) {
      const int rr = u / TC, cc = u - rr * TC;
      const int sy = pcnn_pad_index(y0 + rr - p.pt, p.H, p.pad_mode), sx = pcnn_pad_index(x0 + cc - p.pl, p.W, p.pad_mode);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = 0.f;
        if (ci0 + q < p.Cin) v[q] = (sy < 0 || sx < 0) ? p.pad_value : xin[((int64_t)sy * p.W + sx) * p.ldx + ci0 + q];
      }
      *reinterpret_cast<f32x4*>(lds + 4 * u) = (f32x4){v[0], v[1], v[2], v[3]};
    }
    __syncthreads();
    for (int i = 0; i < p.kh; ++i)
      for (int j = 0; j < p.kw; ++j) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(lds + 4 * ((r + i) * TC + c + j));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ci = ci0 + q;
          if (ci < p.Cin) {                                                   // uniform
            // filter row of (i, j, ci): wave-uniform addresses (scalar loads)
            const float* wr = p.flip ? w + ((int64_t)((p.kh - 1 - i) * p.kw + (p.kw - 1 - j)) * p.Cout) * p.Cin + ci
                                     : w + ((int64_t)(i * p.kw + j) * p.Cin + ci) * p.Cout;
            const int ws = p.flip ? p.Cin : 1;
#pragma unroll
            for (int o = 0; o < CO; ++o)
              if (o < p.Cout) acc[o] = fmaf(xv[q], wr[o * ws], acc[o]);
          }
        }
      }
    __syncthreads();
  }
  const int oy = y0 + r, ox = x0 + c;
  if (oy < p.Ho && ox < p.Wo) {
    float* yo = p.y + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy;
    const float* b = p.bias ? p.bias + (int64_t)n * p.b_stride : nullptr;
#pragma unroll
    for (int o = 0; o < CO; ++o)
      if (o < p.Cout) yo[o] = pcnn_act(acc[o] + (b ? b[o] : 0.f), p.act, p.alpha);
  }
}

// ---- per-sample filter gradient: dw[n][i][j][ci][co] = sum over output pixels of xpad[n][y + i - pt][x + j - pl][ci] dz[n][y][x][co]
struct GroupedWgradParams {
  const float* x; const float* dz; float* part; float* dw;
  long long dw_stride;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, kh, kw, pt, pl, pad_mode; float pad_value;
  int S, rows_per_strip, nout;            // row strips per (sample, filter row); nout = kw Cin Cout outputs of a filter row
  int jw, jchunks;                        // filter columns per workgroup (jw Cin Cout <= 4096) and chunks per filter row
};
constexpr int GXC = 128;                  // columns staged at a time
constexpr int GNE = 16;                   // outputs per thread at most (kw Cin Cout <= 4096)

__global__ __launch_bounds__(256) void grouped_wgrad_kernel(GroupedWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];                  // x chunk [(GXC + kw - 1)][Cin] then dz chunk [GXC][Cout]
  float* xs = lds;
  float* zs = lds + (GXC + p.kw - 1) * p.Cin;
  const int s = blockIdx.x, i = blockIdx.y / p.jchunks, jc = blockIdx.y % p.jchunks, n = blockIdx.z;
  const int j0 = jc * p.jw, nloc = min(p.jw, p.kw - j0) * p.Cin * p.Cout;          // this workgroup's filter columns and outputs
  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
  const float* dzn = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
  float acc[GNE];
  int xoff[GNE], zoff[GNE];
#pragma unroll
  for (int e = 0; e < GNE; ++e) {
    acc[e] = 0.f;
    const int o = threadIdx.x + 256 * e;                                        // (j, ci, co)
    const int co = o % p.Cout, ci = (o / p.Cout) % p.Cin, j = j0 + o / (p.Cout * p.Cin);
    xoff[e] = j * p.Cin + ci; zoff[e] = co;
  }
  const int ya = s * p.rows_per_strip, yb = min(p.Ho, ya + p.rows_per_strip);
  for (int y = ya; y < yb; ++y) {
    const int sy = pcnn_pad_index(y + i - p.pt, p.H, p.pad_mode);
    for (int xa = 0; xa < p.Wo; xa += GXC) {
      const int nx = min(GXC, p.Wo - xa);
      for (int u = threadIdx.x; u < (nx + p.kw - 1) * p.Cin; u += 256) {
        const int px = u / p.Cin, ci = u - px * p.Cin;
        const int sx = pcnn_pad_index(xa + px - p.pl, p.W, p.pad_mode);
        xs[u] = (sy < 0 || sx < 0) ? p.pad_value : xin[((int64_t)sy * p.W + sx) * p.ldx + ci];
      }
      for (int u = threadIdx.x; u < nx * p.Cout; u += 256) {
        const int px = u / p.Cout, co = u - px * p.Cout;
        zs[u] = dzn[((int64_t)y * p.Wo + xa + px) * p.lddz + co];
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < GNE; ++e) {
        if (threadIdx.x + 256 * e < nloc) {
          float a = acc[e];
          for (int px = 0; px < nx; ++px) a = fmaf(xs[px * p.Cin + xoff[e]], zs[px * p.Cout + zoff[e]], a);
          acc[e] = a;
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int e = 0; e < GNE; ++e) {
    const int o = threadIdx.x + 256 * e;
    if (o < nloc) p.part[(((int64_t)n * p.S + s) * p.kh + i) * p.nout + (int64_t)j0 * p.Cin * p.Cout + o] = acc[e];
  }
}

__global__ void grouped_wgrad_reduce_kernel(GroupedWgradParams p) {
  const int64_t per = (int64_t)p.kh * p.nout, total = (int64_t)p.N * per;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / per, e = t - n * per;
    float a = 0.f;
    for (int s = 0; s < p.S; ++s) a += p.part[((n * p.S + s) * p.kh) * p.nout + e];       // fixed order: deterministic
    p.dw[n * p.dw_stride + e] = a;
  }
}

// ---- transposed convolution with kernel = stride f, TensorFlow's padding='SAME' (layers/metalearning_deconvupscale.py:13-16,28-30):
//   y[n][Y][X][co] = b[n][co] + sum_ci x[n][(Y + py) / f][(X + px) / f][ci] K[n][(Y + py) % f][(X + px) % f][co][ci],
//   py = (H f - Ho) / 2, px = (W f - Wo) / 2 - the crop offset of conv2d_transpose when the coarse grid overhangs the output (H = ceil(Ho / f)),
//   the same convention as deconv_mfma.hip and oracle/np_ops.py:conv2d_transpose_same.
struct GroupedDeconvParams {
  const float* x; const float* k; const float* bias; const float* dy; float* y; float* dx; float* dk; float* dbias; float* part;
  long long k_stride, b_stride;
  int N, H, W, Cin, Ho, Wo, Cout, f, py, px, S, rows_per_strip;
};

__global__ __launch_bounds__(256) void grouped_deconv_fwd_kernel(GroupedDeconvParams p) {
  const int n = blockIdx.y;
  const int64_t per = (int64_t)p.Ho * p.Wo * p.Cout;
  const float* K = p.k + (int64_t)n * p.k_stride;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < per; t += (int64_t)gridDim.x * blockDim.x) {
    const int co = t % p.Cout; const int64_t pix = t / p.Cout; const int X = pix % p.Wo, Y = pix / p.Wo;
    const int Yp = Y + p.py, Xp = X + p.px;
    const int yy = Yp / p.f, xx = Xp / p.f;
    float a = p.bias ? p.bias[(int64_t)n * p.b_stride + co] : 0.f;
    if (yy < p.H && xx < p.W) {
      const float* xv = p.x + (((int64_t)n * p.H + yy) * p.W + xx) * p.Cin;
      const float* kv = K + (((int64_t)(Yp - yy * p.f) * p.f + (Xp - xx * p.f)) * p.Cout + co) * p.Cin;
      for (int ci = 0; ci < p.Cin; ++ci) a = fmaf(xv[ci], kv[ci], a);
    }
    p.y[(int64_t)n * per + t] = a;
  }
}

// dx[n][y][x][ci] = sum over the f x f taps and co of dy[n][f y + u - py][f x + v - px][co] K[n][u][v][co][ci] (fine pixels outside the output: none)
__global__ __launch_bounds__(256) void grouped_deconv_bwd_data_kernel(GroupedDeconvParams p) {
  const int n = blockIdx.y;
  const int64_t per = (int64_t)p.H * p.W * p.Cin;
  const float* K = p.k + (int64_t)n * p.k_stride;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < per; t += (int64_t)gridDim.x * blockDim.x) {
    const int ci = t % p.Cin; const int64_t pix = t / p.Cin; const int x = pix % p.W, y = pix / p.W;
    float a = 0.f;
    for (int u = 0; u < p.f; ++u)
      for (int v = 0; v < p.f; ++v) {
        const int Y = p.f * y + u - p.py, X = p.f * x + v - p.px;
        if (Y >= 0 && Y < p.Ho && X >= 0 && X < p.Wo) {
          const float* g = p.dy + (((int64_t)n * p.Ho + Y) * p.Wo + X) * p.Cout;
          const float* kv = K + ((int64_t)(u * p.f + v) * p.Cout) * p.Cin + ci;
          for (int co = 0; co < p.Cout; ++co) a = fmaf(g[co], kv[(int64_t)co * p.Cin], a);
        }
      }
    p.dx[(int64_t)n * per + t] = a;
  }
}

// dK[n][u][v][co][ci] = sum over (y, x) of dy[n][f y + u - py][f x + v - px][co] x[n][y][x][ci]; dbias[n][co] = sum of dy[n][.][.][co].
// Workgroup = (tap, strip of coarse rows, sample), threads over the (co, ci) pairs, a strip's pixels walked in a fixed order; the S strip
// partials (and the strips' bias partials, written by the tap-0 workgroups) are summed in a fixed order by the reduce kernel: deterministic.
// part layout: [n][s][f f Cout Cin | Cout].
__global__ __launch_bounds__(256) void grouped_deconv_bwd_filter_kernel(GroupedDeconvParams p) {
  const int n = blockIdx.z, s = blockIdx.y, tap = blockIdx.x, u = tap / p.f, v = tap % p.f;
  const int npair = p.Cout * p.Cin;
  const int64_t rec = (int64_t)p.f * p.f * npair + p.Cout;
  float* out = p.part + ((int64_t)n * p.S + s) * rec;
  const int y0 = s * p.rows_per_strip, y1 = min(p.H, y0 + p.rows_per_strip);
  for (int e = threadIdx.x; e < npair; e += 256) {
    const int co = e / p.Cin, ci = e - co * p.Cin;
    float a = 0.f;
    for (int y = y0; y < y1; ++y) {
      const int Y = p.f * y + u - p.py;
      if (Y < 0 || Y >= p.Ho) continue;
      for (int x = 0; x < p.W; ++x) {
        const int X = p.f * x + v - p.px;
        if (X < 0 || X >= p.Wo) continue;
        a = fmaf(p.dy[(((int64_t)n * p.Ho + Y) * p.Wo + X) * p.Cout + co], p.x[(((int64_t)n * p.H + y) * p.W + x) * p.Cin + ci], a);
      }
    }
    out[(int64_t)tap * npair + e] = a;
  }
  if (tap == 0) {                                  // bias partial of this strip: its fine rows are f y0 - py .. f y1 - py (a partition of 0..Ho)
    __shared__ float red[256];
    const int Y0 = max(0, p.f * y0 - p.py), Y1 = min(p.Ho, p.f * y1 - p.py);
    const int64_t cnt = (int64_t)max(0, Y1 - Y0) * p.Wo;
    const float* g = p.dy + ((int64_t)n * p.Ho + Y0) * p.Wo * p.Cout;
    for (int co = 0; co < p.Cout; ++co) {
      float a = 0.f;
      for (int64_t t = threadIdx.x; t < cnt; t += 256) a += g[t * p.Cout + co];
      red[threadIdx.x] = a;
      __syncthreads();
      for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
      if (threadIdx.x == 0) out[(int64_t)p.f * p.f * npair + co] = red[0];
      __syncthreads();
    }
  }
}

__global__ void grouped_deconv_filter_reduce_kernel(GroupedDeconvParams p) {
  const int64_t nk = (int64_t)p.f * p.f * p.Cout * p.Cin, rec = nk + p.Cout, total = (int64_t)p.N * rec;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / rec, e = t - n * rec;
    float a = 0.f;
    for (int s = 0; s < p.S; ++s) a += p.part[(n * p.S + s) * rec + e];
    if (e < nk) p.dk[n * p.k_stride + e] = a;
    else if (p.dbias) p.dbias[n * p.b_stride + (e - nk)] = a;
  }
}


// =====================================================================================================================================
// Matrix-core form of the per-sample-filter convolutions (round 4): a GROUPED IMPLICIT GEMM on v_mfma_f32_4x4x1_16B_f32 - sixteen independent
// 4 x 4 outer products per instruction, exact fp32 (the same fmaf chain as the vector kernel), one instruction = 4 output channels x 64
// pixels x one (tap, input channel) step.  The metalearning layers have 3...8 channels: a 32- or 16-wide MFMA tile would idle on 75-90 % of
// its columns, this shape wastes nothing at Cout = 4 / 8.  What makes it cheap to feed:
//   * the instruction's A-matrix BROADCAST (cbsz = 4, abid = s): all sixteen blocks take block s of the A register, so ONE vector register
//     holds the weights of sixteen K steps (lane 4 b + r = step b, output channel r) - a K step costs no weight traffic at all, the
//     `abid` immediate walks through the register;
//   * the B operand (lane = pixel) is one ds_read_b32 of the staged halo tile per K step; with the pixel stride of the tile ODD (Cin odd:
//     the natural [pixel][channel] order; Cin even: channels padded to 2 / 4 / 8 / 16 and one float of skew per pixel) the 64 lanes hit 32
//     different banks;
//   * forward / data gradient: the accumulators ARE the NHWC output (lane = pixel, register = channel).  Weight gradient: roles swapped -
//     A = dz (16 pixels x 4 output channels per register, broadcast by abid), B = 64 consecutive (filter column, input channel) entries
//     of the staged input row, so one register quad accumulates dw[i][0..63][4 g + r]; an input row is read ONCE for all the filter rows
//     a wave owns (5 MFMAs per LDS read at 19 taps).
// Filters are re-packed per launch by gm_pack_kernel (flip / transpose for the data gradient, zero padding of channels and K steps) into
// the handle's filter scratch: N x kh x MV x CO4 x 64 floats.
// -------------------------------------------------------------------------------------------------------------------------------------
#define GM_MFMA(S, a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, (S), 0)

struct GmPackParams {
  const float* w; float* wp; long long w_stride;
  int N, kh, kw, Cin, Cout, C, MV, CO4, flip;       // C: channel pitch of the K index m = j C + ci (Cin, or Cin padded to a power of two)
};
// wp[(((n kh + i) MV + v) CO4 + g) 64 + 4 b + r] = w'[n][i][j][ci][4 g + r], m = 16 v + b = j C + ci (zero outside the filter)
__global__ __launch_bounds__(256) void gm_pack_kernel(GmPackParams p) {
  const int64_t total = (int64_t)p.N * p.kh * p.MV * p.CO4 * 64;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int lane = t & 63; int64_t q = t >> 6;
    const int g = q % p.CO4; q /= p.CO4;
    const int v = q % p.MV; q /= p.MV;
    const int i = q % p.kh; const int64_t n = q / p.kh;
    const int m = 16 * v + (lane >> 2), co = 4 * g + (lane & 3);
    const int j = m / p.C, ci = m - j * p.C;
    float val = 0.f;
    if (j < p.kw && ci < p.Cin && co < p.Cout) {
      const float* w = p.w + n * p.w_stride;
      val = p.flip ? w[((int64_t)((p.kh - 1 - i) * p.kw + (p.kw - 1 - j)) * p.Cout + co) * p.Cin + ci]
                   : w[((int64_t)(i * p.kw + j) * p.Cin + ci) * p.Cout + co];
    }
    p.wp[t] = val;
  }
}

struct GmFwdParams {
  const float* x; const float* wp; const float* bias; float* y; long long b_stride;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, kh, kw, pt, pl, pad_mode; float pad_value; int act; float alpha;
  int PS, MV, vstride, RS, tiles_x;                 // LDS tile: pixel stride / row stride in floats, K-step chunks per filter row and their stride
};
constexpr int GMR = 4;                              // output rows per workgroup (one per wave), 64 pixels per row

// CP = 0: pixel stride = Cin (odd), the K index m is the float offset itself; CP = 2 / 4 / 8 / 16: channels padded to CP, pixel stride CP + 1
template <int CP> __device__ __forceinline__ constexpr int gm_off(int s) {
  if constexpr (CP == 0) return s;
  else return (s / CP) * (CP + 1) + s % CP;
}

template <int CP, int CO4>
__global__ __launch_bounds__(256) void gm_fwd_kernel(GmFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];                  // [GMR + kh - 1][RS]
  const int TR = GMR + p.kh - 1, TCp = 64 + p.kw - 1;
  const int n = blockIdx.y;
  const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x - ty * p.tiles_x;
  const int y0 = ty * GMR, x0 = tx * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
  // ---- stage the halo tile: every float of the tile is written exactly once (channel padding and the row slack that the zero-weight
  //      K steps read are zero: 0 x NaN from stale LDS would poison the sums)
  for (int rr = wave; rr < TR; rr += 4) {
    const int sy = pcnn_pad_index(y0 + rr - p.pt, p.H, p.pad_mode);
    float* row = lds + rr * p.RS;
    for (int px = lane; px * p.PS < p.RS; px += 64) {
      const int sx = pcnn_pad_index(x0 + px - p.pl, p.W, p.pad_mode);
      const bool inside = px < TCp, pad = sy < 0 || sx < 0;
      const float* src = xin + ((int64_t)(pad ? 0 : sy) * p.W + (pad ? 0 : sx)) * p.ldx;
      for (int ci = 0; ci < p.PS; ++ci) {
        const int e = px * p.PS + ci;
        if (e < p.RS) row[e] = (inside && ci < p.Cin) ? (pad ? p.pad_value : src[ci]) : 0.f;
      }
    }
  }
  __syncthreads();
  f32x4 acc[CO4];
#pragma unroll
  for (int g = 0; g < CO4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* wq = p.wp + (int64_t)n * p.kh * p.MV * CO4 * 64 + lane;          // chunk q = i MV + v: CO4 registers of 64 floats
  const int nq = p.kh * p.MV;
  float wn[CO4];
#pragma unroll
  for (int g = 0; g < CO4; ++g) wn[g] = wq[g * 64];
  int i = 0, v = 0;
  for (int q = 0; q < nq; ++q) {
    float wv[CO4];
#pragma unroll
    for (int g = 0; g < CO4; ++g) wv[g] = wn[g];
    if (q + 1 < nq) {                                                            // the next chunk's weights fly under this chunk's MFMAs
#pragma unroll
      for (int g = 0; g < CO4; ++g) wn[g] = wq[((int64_t)(q + 1) * CO4 + g) * 64];
    }
    const float* xl = lds + (wave + i) * p.RS + lane * p.PS + v * p.vstride;
#define GM_STEP(S)                                                        \
    {                                                                     \
      const float xv = xl[gm_off<CP>(S)];                                 \
      _Pragma("unroll") for (int g = 0; g < CO4; ++g) acc[g] = GM_MFMA(S, wv[g], xv, acc[g]); \
    }
    GM_STEP(0) GM_STEP(1) GM_STEP(2) GM_STEP(3) GM_STEP(4) GM_STEP(5) GM_STEP(6) GM_STEP(7)
    GM_STEP(8) GM_STEP(9) GM_STEP(10) GM_STEP(11) GM_STEP(12) GM_STEP(13) GM_STEP(14) GM_STEP(15)
#undef GM_STEP
    if (++v == p.MV) { v = 0; ++i; }
  }
  const int oy = y0 + wave, ox = x0 + lane;
  if (oy < p.Ho && ox < p.Wo) {
    float* yo = p.y + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy;
    const float* b = p.bias ? p.bias + (int64_t)n * p.b_stride : nullptr;
#pragma unroll
    for (int g = 0; g < CO4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * g + r;
        if (co < p.Cout) yo[co] = pcnn_act(acc[g][r] + (b ? b[co] : 0.f), p.act, p.alpha);
      }
  }
}

// ---- weight gradient.  Workgroup = 4 waves, wave w owns the NI = ceil(kh / 4) consecutive filter rows i = NI w ... NI w + NI - 1; the
// workgroup walks tiles of GWR output rows x GWC output columns of ONE sample (grid-stride over the sample's tiles, the partial sums stay
// in the accumulators), staging the input halo tile [GWR + kh - 1][(GWC + kw - 1) Cin] and the dz tile [GWR][GWC][4 CO4] in LDS.  A wave
// walks the input rows Y that any of its filter rows touches (NI - 1 + rows of them) and, per 16-pixel chunk, loads the A registers of the
// NI output rows Y - i (ZERO where Y - i falls outside the tile: the MFMA is issued regardless - a branch around an MFMA makes the compiler
// shuffle the accumulator tuples through VGPR copies, measured 10x slower - so (rows + NI - 1) / rows of the issued products are useful);
// then ONE read of 64 consecutive floats of input row Y per pixel feeds all NI of them.
// Results: part[((n S + s) kh + i) nout + m Cout + co] - the layout of grouped_wgrad_reduce_kernel.
constexpr int GWR = 16, GWC = 64;
struct GmWgradParams {
  const float* x; const float* dz; float* part;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, kh, kw, pt, pl, pad_mode; float pad_value;
  int S, nout, XRS, tiles_x, tiles, GR;           // S workgroups per sample; XRS: floats per staged input row; tiles per sample; rows per tile
};

template <int NI, int MC, int CO4>
__global__ __launch_bounds__(256) void gm_wgrad_kernel(GmWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int COP = 4 * CO4;
  const int GR = p.GR, TR = GR + p.kh - 1;          // GR output rows per tile (16, or 8 when the staged rows are long)
  float* xs = lds;                                  // [TR][XRS]
  float* zs = lds + TR * p.XRS;                     // [GR][GWC][COP]
  const int s = blockIdx.x, n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = wave * NI;
  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
  const float* dzn = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
  f32x4 acc[NI][MC][CO4];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MC; ++b)
#pragma unroll
      for (int g = 0; g < CO4; ++g) acc[a][b][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int xw = (GWC + p.kw - 1) * p.Cin;          // meaningful floats of a staged input row
  for (int t = s; t < p.tiles; t += p.S) {
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const int y0 = ty * GR, x0 = tx * GWC;
    const int rows = min(GR, p.Ho - y0);
    __syncthreads();                                // the previous tile's readers are done
    for (int rr = wave; rr < rows + p.kh - 1; rr += 4) {
      const int sy = pcnn_pad_index(y0 + rr - p.pt, p.H, p.pad_mode);
      float* row = xs + rr * p.XRS;
      int px = lane / p.Cin, ci = lane - px * p.Cin;                            // e = lane + 64 k  ->  (pixel, channel), advanced without divisions
      const int dpx = 64 / p.Cin, dci = 64 - dpx * p.Cin;
      for (int e = lane; e < p.XRS; e += 64) {
        float val = 0.f;
        if (e < xw) {
          const int sx = pcnn_pad_index(x0 + px - p.pl, p.W, p.pad_mode);
          val = (sy < 0 || sx < 0) ? p.pad_value : xin[((int64_t)sy * p.W + sx) * p.ldx + ci];
        }
        row[e] = val;
        px += dpx; ci += dci;
        if (ci >= p.Cin) { ci -= p.Cin; ++px; }
      }
    }
    for (int e = threadIdx.x; e < GR * GWC * COP; e += 256) {
      const int co = e % COP, px = (e / COP) % GWC, rr = e / (COP * GWC);
      zs[e] = (rr < rows && x0 + px < p.Wo && co < p.Cout) ? dzn[((int64_t)(y0 + rr) * p.Wo + x0 + px) * p.lddz + co] : 0.f;
    }
    __syncthreads();
    const int Yend = min(i0 + NI - 1, p.kh - 1) + rows;                          // one past the last input row this wave's filter rows touch
    for (int Y = i0; Y < Yend; ++Y) {
      const float* xrow = xs + Y * p.XRS + lane;
      int zoff[NI];
      bool ok[NI];
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        const int ry = Y - (i0 + a);
        ok[a] = i0 + a < p.kh && ry >= 0 && ry < rows;                           // wave-uniform; the load below is unconditional (clamped row)
        zoff[a] = (min(max(ry, 0), GR - 1) * GWC + (lane >> 2)) * COP + (lane & 3);
      }
#pragma unroll 1
      for (int xc = 0; xc < GWC / 16; ++xc) {
        float A[NI][CO4];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
          for (int g = 0; g < CO4; ++g) {
            const float zv = zs[zoff[a] + xc * 16 * COP + 4 * g];
            A[a][g] = ok[a] ? zv : 0.f;
          }
        const float* xl = xrow + xc * 16 * p.Cin;
#define GM_WSTEP(S)                                                                                   \
        {                                                                                             \
          float B[MC];                                                                                \
          _Pragma("unroll") for (int b = 0; b < MC; ++b) B[b] = xl[(S) * p.Cin + 64 * b];              \
          _Pragma("unroll") for (int a = 0; a < NI; ++a)                                              \
            _Pragma("unroll") for (int b = 0; b < MC; ++b)                                            \
              _Pragma("unroll") for (int g = 0; g < CO4; ++g) acc[a][b][g] = GM_MFMA(S, A[a][g], B[b], acc[a][b][g]); \
        }
        GM_WSTEP(0) GM_WSTEP(1) GM_WSTEP(2) GM_WSTEP(3) GM_WSTEP(4) GM_WSTEP(5) GM_WSTEP(6) GM_WSTEP(7)
        GM_WSTEP(8) GM_WSTEP(9) GM_WSTEP(10) GM_WSTEP(11) GM_WSTEP(12) GM_WSTEP(13) GM_WSTEP(14) GM_WSTEP(15)
#undef GM_WSTEP
      }
    }
  }
  const int km = p.kw * p.Cin;
#pragma unroll
  for (int a = 0; a < NI; ++a) {
    const int i = i0 + a;
    if (i < p.kh) {
      float* out = p.part + (((int64_t)n * p.S + s) * p.kh + i) * p.nout;
#pragma unroll
      for (int b = 0; b < MC; ++b) {
        const int m = 64 * b + lane;
        if (m < km) {
#pragma unroll
          for (int g = 0; g < CO4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (4 * g + r < p.Cout) out[(int64_t)m * p.Cout + 4 * g + r] = acc[a][b][g][r];
        }
      }
    }
  }
}

}  // namespace

// ---- route choice of the matrix-core kernels (everything else stays on the vector-ALU kernels above)
struct GmFwdPlan { bool ok; int CP, C, PS, MV, vstride, RS, CO4; size_t lds; };
static GmFwdPlan gm_fwd_plan(const pcnn_conv_desc* d) {
  GmFwdPlan q{};
  q.ok = false;
  if (getenv("PCNN_GROUPED_VALU")) return q;                       // developer switch: the vector-ALU kernels only (A/B timing, tests of both routes)
  if (d->Cout > 16 || d->Cin > 16) return q;
  if (d->Cin & 1) { q.CP = 0; q.C = d->Cin; q.PS = d->Cin; q.vstride = 16; }
  else { q.CP = 2; while (q.CP < d->Cin) q.CP *= 2; q.C = q.CP; q.PS = q.CP + 1; q.vstride = (16 / q.CP) * q.PS; }
  q.MV = pcnn_cdiv(d->kw * q.C, 16);
  q.RS = std::max((64 + d->kw - 1) * q.PS, 63 * q.PS + q.MV * q.vstride);
  q.CO4 = d->Cout <= 4 ? 1 : (d->Cout <= 8 ? 2 : 4);                  // register quads per pixel (12 channels ride in 4 quads, the filter's 4th is zero)
  q.lds = (size_t)(GMR + d->kh - 1) * q.RS * sizeof(float);
  q.ok = q.lds <= 128 * 1024;
  return q;
}
struct GmWgradPlan { bool ok; int NI, MC, CO4, XRS, tiles_x, tiles, S, GR; size_t lds; };
static GmWgradPlan gm_wgrad_plan(const pcnn_conv_desc* d) {
  GmWgradPlan q{};
  q.ok = false;
  if (getenv("PCNN_GROUPED_VALU")) return q;
  if (d->Cout > 16 || d->kh > 20 || d->kw * d->Cin > 192) return q;
  q.NI = pcnn_cdiv(d->kh, 4);
  q.MC = pcnn_cdiv(d->kw * d->Cin, 64); q.CO4 = d->Cout <= 4 ? 1 : (d->Cout <= 8 ? 2 : 4);
  if (q.NI * q.MC * q.CO4 > 20) return q;                            // accumulator quads per wave (80 registers)
  q.XRS = std::max((GWC + d->kw - 1) * d->Cin, (GWC - 1) * d->Cin + 64 * q.MC);
  for (q.GR = GWR; q.GR >= 4; q.GR /= 2) {                          // output rows per tile: as many as fit (the wave's NI - 1 extra input rows amortise over them)
    q.lds = ((size_t)(q.GR + d->kh - 1) * q.XRS + (size_t)q.GR * GWC * 4 * q.CO4) * sizeof(float);
    if (q.lds <= 128 * 1024) break;
  }
  q.tiles_x = pcnn_cdiv(d->Wo, GWC); q.tiles = q.tiles_x * pcnn_cdiv(d->Ho, q.GR);
  q.S = std::min(q.tiles, 64);
  q.ok = q.GR >= 4 && q.lds <= 128 * 1024;
  return q;
}

extern "C" size_t pcnn_grouped_conv2d_wgrad_workspace(const pcnn_conv_desc* d) {
  if (!d) return 0;
  const int S = std::max(std::max(1, std::min(16, d->Ho / 8)), gm_wgrad_plan(d).ok ? gm_wgrad_plan(d).S : 1);
  return (size_t)d->N * S * d->kh * d->kw * d->Cin * d->Cout * sizeof(float);
}

/* 1 when the call takes the matrix-core route (v_mfma_f32_4x4x1_16B_f32), 0 for the vector-ALU kernels; what = 0 forward / data gradient, 1 filter gradient */
extern "C" int pcnn_grouped_conv2d_uses_mfma(const pcnn_conv_desc* d, int what) {
  if (!d) return 0;
  return what ? (gm_wgrad_plan(d).ok ? 1 : 0) : (gm_fwd_plan(d).ok ? 1 : 0);
}

template <int CP>
static void gm_fwd_launch(int CO4, dim3 grid, size_t lds, hipStream_t st, const GmFwdParams& p) {
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gm_fwd_kernel<CP, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gm_fwd_kernel<CP, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gm_fwd_kernel<CP, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (CO4 == 1) hipLaunchKernelGGL((gm_fwd_kernel<CP, 1>), grid, dim3(256), lds, st, p);
  else if (CO4 == 2) hipLaunchKernelGGL((gm_fwd_kernel<CP, 2>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((gm_fwd_kernel<CP, 4>), grid, dim3(256), lds, st, p);
}

extern "C" int pcnn_grouped_conv2d_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, long long w_sample_stride, const float* bias,
                                       long long bias_sample_stride, int flip_transpose, float* y) {
  PCNN_REQUIRE(h, h && d && x && w && y, "pcnn_grouped_conv2d_fwd: null argument");
  PCNN_REQUIRE(h, d->Cout >= 1 && d->Cout <= 32 && d->Cin >= 1 && d->kh >= 1 && d->kw >= 1 && d->kh <= 31 && d->kw <= 31,
               "pcnn_grouped_conv2d_fwd: %d -> %d channels, %d x %d taps unsupported (<= 32 output channels, <= 31 taps)", d->Cin, d->Cout, d->kh, d->kw);
  const GmFwdPlan q = gm_fwd_plan(d);
  if (q.ok) {                                                        // grouped implicit GEMM on the matrix cores
    const size_t need = (size_t)d->N * d->kh * q.MV * q.CO4 * 64 * sizeof(float);
    if (h->scratch_bytes < need) {
      if (h->scratch) { pcnn_release(h, h->scratch); h->scratch = nullptr; h->scratch_bytes = 0; }
      if (hipMalloc(&h->scratch, need) != hipSuccess) PCNN_FAIL(h, "pcnn_grouped_conv2d_fwd: cannot allocate %zu B of filter scratch", need);
      h->scratch_bytes = need;
    }
    GmPackParams k;
    k.w = w; k.wp = static_cast<float*>(h->scratch); k.w_stride = w_sample_stride; k.N = d->N; k.kh = d->kh; k.kw = d->kw; k.Cin = d->Cin; k.Cout = d->Cout;
    k.C = q.C; k.MV = q.MV; k.CO4 = q.CO4; k.flip = flip_transpose ? 1 : 0;
    hipLaunchKernelGGL(gm_pack_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64((int64_t)need / 4, 256), 2048)), dim3(256), 0, h->stream, k);
    GmFwdParams g;
    g.x = x; g.wp = k.wp; g.bias = bias; g.y = y; g.b_stride = bias_sample_stride;
    g.N = d->N; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.ldx = d->ldx; g.Ho = d->Ho; g.Wo = d->Wo; g.Cout = d->Cout; g.ldy = d->ldy;
    g.kh = d->kh; g.kw = d->kw; g.pt = d->pad_top; g.pl = d->pad_left; g.pad_mode = d->pad_mode; g.pad_value = d->pad_value; g.act = d->act; g.alpha = d->act_alpha;
    g.PS = q.PS; g.MV = q.MV; g.vstride = q.vstride; g.RS = q.RS; g.tiles_x = pcnn_cdiv(d->Wo, 64);
    const dim3 grid((unsigned)(g.tiles_x * pcnn_cdiv(d->Ho, GMR)), (unsigned)d->N);
    switch (q.CP) {
      case 0: gm_fwd_launch<0>(q.CO4, grid, q.lds, h->stream, g); break;
      case 2: gm_fwd_launch<2>(q.CO4, grid, q.lds, h->stream, g); break;
      case 4: gm_fwd_launch<4>(q.CO4, grid, q.lds, h->stream, g); break;
      case 8: gm_fwd_launch<8>(q.CO4, grid, q.lds, h->stream, g); break;
      default: gm_fwd_launch<16>(q.CO4, grid, q.lds, h->stream, g); break;
    }
    PCNN_CHECK_LAUNCH(h, "pcnn_grouped_conv2d_fwd (mfma)");
    return 0;
  }
  GroupedParams p;
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.w_stride = w_sample_stride; p.b_stride = bias_sample_stride;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.ldy = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value; p.act = d->act; p.alpha = d->act_alpha;
  p.flip = flip_transpose ? 1 : 0;
  p.tiles_x = pcnn_cdiv(d->Wo, GTW); p.tiles_y = pcnn_cdiv(d->Ho, GTH);
  const size_t lds = (size_t)(GTH + d->kh - 1) * (GTW + d->kw - 1) * 4 * sizeof(float);
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)d->N);
  if (d->Cout <= 4) hipLaunchKernelGGL(grouped_fwd_kernel<4>, grid, dim3(256), lds, h->stream, p);
  else if (d->Cout <= 8) hipLaunchKernelGGL(grouped_fwd_kernel<8>, grid, dim3(256), lds, h->stream, p);
  else if (d->Cout <= 16) hipLaunchKernelGGL(grouped_fwd_kernel<16>, grid, dim3(256), lds, h->stream, p);
  else hipLaunchKernelGGL(grouped_fwd_kernel<32>, grid, dim3(256), lds, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_conv2d_fwd");
  return 0;
}

extern "C" int pcnn_grouped_conv2d_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw, long long dw_sample_stride, void* workspace) {
  PCNN_REQUIRE(h, h && d && x && dz && dw && workspace, "pcnn_grouped_conv2d_wgrad: null argument");
  const int nout = d->kw * d->Cin * d->Cout;
  const GmWgradPlan q = gm_wgrad_plan(d);
  if (q.ok) {                                                        // matrix-core route
    GmWgradParams g;
    g.x = x; g.dz = dz; g.part = static_cast<float*>(workspace);
    g.N = d->N; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.ldx = d->ldx; g.Ho = d->Ho; g.Wo = d->Wo; g.Cout = d->Cout; g.lddz = d->ldy;
    g.kh = d->kh; g.kw = d->kw; g.pt = d->pad_top; g.pl = d->pad_left; g.pad_mode = d->pad_mode; g.pad_value = d->pad_value;
    g.S = q.S; g.nout = nout; g.XRS = q.XRS; g.tiles_x = q.tiles_x; g.tiles = q.tiles; g.GR = q.GR;
    const dim3 grid((unsigned)q.S, (unsigned)d->N);
#define GM_WLAUNCH(NIv, MCv, COv)                                                                                                \
    do {                                                                                                                         \
      if (q.lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gm_wgrad_kernel<NIv, MCv, COv>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds); \
      hipLaunchKernelGGL((gm_wgrad_kernel<NIv, MCv, COv>), grid, dim3(256), q.lds, h->stream, g);                                \
    } while (0)
#define GM_WLAUNCH_MC(NIv, MCv)                                                   \
    do {                                                                          \
      if constexpr ((NIv) * (MCv) * 1 <= 20) { if (q.CO4 == 1) GM_WLAUNCH(NIv, MCv, 1); } \
      if constexpr ((NIv) * (MCv) * 2 <= 20) { if (q.CO4 == 2) GM_WLAUNCH(NIv, MCv, 2); } \
      if constexpr ((NIv) * (MCv) * 4 <= 20) { if (q.CO4 == 4) GM_WLAUNCH(NIv, MCv, 4); } \
    } while (0)
#define GM_WLAUNCH_NI(NIv)                                  \
    do {                                                    \
      if (q.MC == 1) GM_WLAUNCH_MC(NIv, 1);                 \
      else if (q.MC == 2) GM_WLAUNCH_MC(NIv, 2);            \
      else GM_WLAUNCH_MC(NIv, 3);                           \
    } while (0)
    switch (q.NI) {
      case 1: GM_WLAUNCH_NI(1); break;
      case 2: GM_WLAUNCH_NI(2); break;
      case 3: GM_WLAUNCH_NI(3); break;
      case 4: GM_WLAUNCH_NI(4); break;
      default: GM_WLAUNCH_NI(5); break;
    }
#undef GM_WLAUNCH_MC
#undef GM_WLAUNCH_NI
#undef GM_WLAUNCH
    GroupedWgradParams r;
    r.x = x; r.dz = dz; r.part = g.part; r.dw = dw; r.dw_stride = dw_sample_stride; r.N = d->N; r.kh = d->kh; r.S = q.S; r.nout = nout;
    hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64((int64_t)d->N * d->kh * nout, 256), 1024)), dim3(256), 0, h->stream, r);
    PCNN_CHECK_LAUNCH(h, "pcnn_grouped_conv2d_wgrad (mfma)");
    return 0;
  }
  PCNN_REQUIRE(h, d->Cin * d->Cout <= 256 * GNE, "pcnn_grouped_conv2d_wgrad: Cin Cout = %d exceeds %d", d->Cin * d->Cout, 256 * GNE);
  GroupedWgradParams p;
  p.x = x; p.dz = dz; p.part = static_cast<float*>(workspace); p.dw = dw; p.dw_stride = dw_sample_stride;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.lddz = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.S = std::max(1, std::min(16, d->Ho / 8)); p.rows_per_strip = pcnn_cdiv(d->Ho, p.S); p.nout = nout;
  p.jw = std::min(d->kw, (256 * GNE) / (d->Cin * d->Cout)); p.jchunks = pcnn_cdiv(d->kw, p.jw);
  const size_t lds = ((size_t)(GXC + d->kw - 1) * d->Cin + (size_t)GXC * d->Cout) * sizeof(float);
  PCNN_REQUIRE(h, lds <= 64 * 1024, "pcnn_grouped_conv2d_wgrad: %d + %d channels do not fit the staging buffer", d->Cin, d->Cout);
  hipLaunchKernelGGL(grouped_wgrad_kernel, dim3((unsigned)p.S, (unsigned)(d->kh * p.jchunks), (unsigned)d->N), dim3(256), lds, h->stream, p);
  hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64((int64_t)d->N * d->kh * nout, 256), 1024)), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_conv2d_wgrad");
  return 0;
}

static GroupedDeconvParams deconv_params(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, long long ks, long long bs) {
  GroupedDeconvParams p;
  p.x = nullptr; p.k = nullptr; p.bias = nullptr; p.dy = nullptr; p.y = nullptr; p.dx = nullptr; p.dk = nullptr; p.dbias = nullptr;
  p.part = nullptr; p.S = 1; p.rows_per_strip = H;
  p.k_stride = ks; p.b_stride = bs; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.f = f;
  p.py = (H * f - Ho) / 2; p.px = (W * f - Wo) / 2;          // conv2d_transpose 'SAME' crop offset (the geometry is checked by the callers below)
  return p;
}

extern "C" int pcnn_grouped_deconv_fwd(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* x, const float* k,
                                       long long k_sample_stride, const float* bias, long long bias_sample_stride, float* y) {
  PCNN_REQUIRE(h, h && x && k && y && f >= 1, "pcnn_grouped_deconv_fwd: bad argument");
  PCNN_REQUIRE(h, N >= 1 && Cin >= 1 && Cout >= 1 && Ho >= 1 && Wo >= 1 && H == pcnn_cdiv(Ho, f) && W == pcnn_cdiv(Wo, f),
               "%s: padding='SAME' with stride %d needs a coarse grid of ceil(%d / %d) x ceil(%d / %d), got %d x %d", "pcnn_grouped_deconv_fwd", f, Ho, f, Wo, f, H, W);
  GroupedDeconvParams p = deconv_params(N, H, W, Cin, Ho, Wo, Cout, f, k_sample_stride, bias_sample_stride);
  p.x = x; p.k = k; p.bias = bias; p.y = y;
  const int64_t per = (int64_t)Ho * Wo * Cout;
  hipLaunchKernelGGL(grouped_deconv_fwd_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(per, 256), 4096), (unsigned)N), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_deconv_fwd");
  return 0;
}

extern "C" int pcnn_grouped_deconv_bwd_data(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* dy, const float* k,
                                            long long k_sample_stride, float* dx) {
  PCNN_REQUIRE(h, h && dy && k && dx && f >= 1, "pcnn_grouped_deconv_bwd_data: bad argument");
  PCNN_REQUIRE(h, N >= 1 && Cin >= 1 && Cout >= 1 && Ho >= 1 && Wo >= 1 && H == pcnn_cdiv(Ho, f) && W == pcnn_cdiv(Wo, f),
               "%s: padding='SAME' with stride %d needs a coarse grid of ceil(%d / %d) x ceil(%d / %d), got %d x %d", "pcnn_grouped_deconv_bwd_data", f, Ho, f, Wo, f, H, W);
  GroupedDeconvParams p = deconv_params(N, H, W, Cin, Ho, Wo, Cout, f, k_sample_stride, 0);
  p.dy = dy; p.k = k; p.dx = dx;
  const int64_t per = (int64_t)H * W * Cin;
  hipLaunchKernelGGL(grouped_deconv_bwd_data_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(per, 256), 4096), (unsigned)N), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_deconv_bwd_data");
  return 0;
}

extern "C" int pcnn_grouped_deconv_bwd_filter(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* x, const float* dy,
                                              float* dk, long long k_sample_stride, float* dbias, long long bias_sample_stride) {
  PCNN_REQUIRE(h, h && x && dy && dk && f >= 1, "pcnn_grouped_deconv_bwd_filter: bad argument");
  PCNN_REQUIRE(h, N >= 1 && Cin >= 1 && Cout >= 1 && Ho >= 1 && Wo >= 1 && H == pcnn_cdiv(Ho, f) && W == pcnn_cdiv(Wo, f),
               "%s: padding='SAME' with stride %d needs a coarse grid of ceil(%d / %d) x ceil(%d / %d), got %d x %d", "pcnn_grouped_deconv_bwd_filter", f, Ho, f, Wo, f, H, W);
  GroupedDeconvParams p = deconv_params(N, H, W, Cin, Ho, Wo, Cout, f, k_sample_stride, bias_sample_stride);
  p.x = x; p.dy = dy; p.dk = dk; p.dbias = dbias;
  p.S = std::max(1, std::min(32, H / 2)); p.rows_per_strip = pcnn_cdiv(H, p.S); p.S = pcnn_cdiv(H, p.rows_per_strip);
  const int64_t rec = (int64_t)f * f * Cout * Cin + Cout;
  const size_t need = (size_t)N * p.S * rec * sizeof(float);
  if (h->aux_ws_bytes < need) {                              // handle-owned scratch (shared with the two-pass resize; one stream per handle)
    if (h->aux_ws) { pcnn_release(h, h->aux_ws); h->aux_ws = nullptr; h->aux_ws_bytes = 0; }
    if (hipMalloc(&h->aux_ws, need) != hipSuccess) PCNN_FAIL(h, "pcnn_grouped_deconv_bwd_filter: cannot allocate %zu B of scratch", need);
    h->aux_ws_bytes = need;
  }
  p.part = static_cast<float*>(h->aux_ws);
  hipLaunchKernelGGL(grouped_deconv_bwd_filter_kernel, dim3((unsigned)(f * f), (unsigned)p.S, (unsigned)N), dim3(256), 0, h->stream, p);
  hipLaunchKernelGGL(grouped_deconv_filter_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64((int64_t)N * rec, 256), 1024)), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_deconv_bwd_filter");
  return 0;
}

// dbias[n][co] = sum over the pixels of sample n of dz[n][.][co] (one workgroup per sample, fixed summation order)
namespace {
__global__ __launch_bounds__(256) void grouped_bias_grad_kernel(const float* __restrict__ dz, int64_t HW, int C, int lddz, float* __restrict__ dbias, long long b_stride) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  const float* g = dz + (int64_t)n * HW * lddz;
  for (int co = 0; co < C; ++co) {
    float a = 0.f;
    for (int64_t t = threadIdx.x; t < HW; t += 256) a += g[t * lddz + co];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) dbias[(int64_t)n * b_stride + co] = red[0];
    __syncthreads();
  }
}
}  // namespace

extern "C" int pcnn_grouped_bias_grad(pcnn_handle h, int N, long long HW, int C, const float* dz, int lddz, float* dbias, long long bias_sample_stride) {
  PCNN_REQUIRE(h, h && dz && dbias && N >= 1 && C >= 1, "pcnn_grouped_bias_grad: bad argument");
  hipLaunchKernelGGL(grouped_bias_grad_kernel, dim3((unsigned)N), dim3(256), 0, h->stream, dz, (int64_t)HW, C, lddz, dbias, bias_sample_stride);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_bias_grad");
  return 0;
}
