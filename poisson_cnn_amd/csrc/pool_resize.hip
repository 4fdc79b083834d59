// TF-SAME pooling (stride == window) and tf.image.resize (half-pixel, antialias=False) forward/backward, NHWC.
#include <math.h>
#include <vector>
#include "pcnn_internal.h"

namespace {

__device__ __forceinline__ void win(int o, int f, int pb, int n, int& a, int& b) {
  a = o * f - pb; b = a + f;
  if (a < 0) a = 0;
  if (b > n) b = n;
}

// small windows: one thread per output element
// float4 form of the small-window forward (C, the channel strides and the base pointers multiples of 4 floats)
__global__ void pool_fwd_small_vec4_kernel(int kind, int N, int H, int W, int C, int f, int Ho, int Wo, int pby, int pbx,
                                           const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy);
__global__ void pool_fwd_small_kernel(int kind, int N, int H, int W, int C, int f, int Ho, int Wo, int pby, int pbx,
                                      const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; int64_t r = i / C; const int ox = r % Wo; r /= Wo; const int oy = r % Ho; const int n = r / Ho;
    int y0, y1, x0, x1;
    win(oy, f, pby, H, y0, y1); win(ox, f, pbx, W, x0, x1);
    float acc = kind == PCNN_POOL_MAX ? -INFINITY : 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const float v = x[(((int64_t)n * H + yy) * W + xx) * ldx + c];
        acc = kind == PCNN_POOL_MAX ? fmaxf(acc, v) : acc + v;
      }
    if (kind == PCNN_POOL_AVERAGE) acc /= (float)((y1 - y0) * (x1 - x0));
    y[(((int64_t)n * Ho + oy) * Wo + ox) * ldy + c] = acc;
  }
}
__global__ void pool_fwd_small_vec4_kernel(int kind, int N, int H, int W, int C, int f, int Ho, int Wo, int pby, int pbx,
                                           const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy) {
  const int CV = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) << 2; int64_t r = i / CV; const int ox = r % Wo; r /= Wo; const int oy = r % Ho; const int n = r / Ho;
    int y0, y1, x0, x1;
    win(oy, f, pby, H, y0, y1); win(ox, f, pbx, W, x0, x1);
    const float init = kind == PCNN_POOL_MAX ? -INFINITY : 0.f;
    float4 acc = make_float4(init, init, init, init);
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)n * H + yy) * W + xx) * ldx + c);
        if (kind == PCNN_POOL_MAX) { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      }
    if (kind == PCNN_POOL_AVERAGE) { const float d = (float)((y1 - y0) * (x1 - x0)); acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d; }
    *reinterpret_cast<float4*>(y + (((int64_t)n * Ho + oy) * Wo + ox) * ldy + c) = acc;
  }
}

// large windows: one workgroup per output pixel, threads over (window element, channel)
__global__ __launch_bounds__(256) void pool_fwd_large_kernel(int kind, int N, int H, int W, int C, int CP, int f, int Ho, int Wo, int pby, int pbx,
                                                             const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy) {
  __shared__ float red[256];
  int b = blockIdx.x;
  const int ox = b % Wo; b /= Wo; const int oy = b % Ho; const int n = b / Ho;
  const int tid = threadIdx.x, c = tid % CP, r = tid / CP, R = 256 / CP;
  int y0, y1, x0, x1;
  win(oy, f, pby, H, y0, y1); win(ox, f, pbx, W, x0, x1);
  const int ww = x1 - x0, cnt = (y1 - y0) * ww;
  float acc = kind == PCNN_POOL_MAX ? -INFINITY : 0.f;
  if (c < C)
    for (int e = r; e < cnt; e += R) {
      const int yy = y0 + e / ww, xx = x0 + e % ww;
      const float v = x[(((int64_t)n * H + yy) * W + xx) * ldx + c];
      acc = kind == PCNN_POOL_MAX ? fmaxf(acc, v) : acc + v;
    }
  red[tid] = acc;
  __syncthreads();
  if (r == 0 && c < C) {
    float t = red[c];
    for (int q = 1; q < R; ++q) t = kind == PCNN_POOL_MAX ? fmaxf(t, red[q * CP + c]) : t + red[q * CP + c];
    if (kind == PCNN_POOL_AVERAGE) t /= (float)cnt;
    y[(((int64_t)n * Ho + oy) * Wo + ox) * ldy + c] = t;
  }
}

template <int V>   // V channels per thread (4: float4 accesses)
__global__ __launch_bounds__(256) void pool_bwd_avg_kernel(int N, int H, int W, int C, int f, int Ho, int Wo, int pby, int pbx, const float* __restrict__ dy,
                                                           int lddy, float* __restrict__ dx, int lddx, int accumulate) {
  const int CV = C / V;
  const int64_t total = (int64_t)N * H * W * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V; int64_t r = i / CV; const int xx = r % W; r /= W; const int yy = r % H; const int n = r / H;
    const int oy = (yy + pby) / f, ox = (xx + pbx) / f;
    int y0, y1, x0, x1;
    win(oy, f, pby, H, y0, y1); win(ox, f, pbx, W, x0, x1);
    const float cnt = (float)((y1 - y0) * (x1 - x0));
    const float* src = dy + (((int64_t)n * Ho + oy) * Wo + ox) * lddy + c;
    float* dst = dx + (((int64_t)n * H + yy) * W + xx) * lddx + c;
    if (V == 4) {
      const float4 g = *reinterpret_cast<const float4*>(src);
      float4 o = make_float4(g.x / cnt, g.y / cnt, g.z / cnt, g.w / cnt);
      if (accumulate) { const float4 d = *reinterpret_cast<const float4*>(dst); o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
      *reinterpret_cast<float4*>(dst) = o;
    } else {
      const float g = *src / cnt;
      *dst = accumulate ? *dst + g : g;
    }
  }
}

// max: gradient goes to the first maximum of the window (row-major), like tf.nn.max_pool's backprop
__global__ void pool_bwd_max_kernel(int N, int H, int W, int C, int f, int Ho, int Wo, int pby, int pbx, const float* __restrict__ x, int ldx,
                                    const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; int64_t r = i / C; const int ox = r % Wo; r /= Wo; const int oy = r % Ho; const int n = r / Ho;
    int y0, y1, x0, x1;
    win(oy, f, pby, H, y0, y1); win(ox, f, pbx, W, x0, x1);
    float best = -INFINITY; int by = y0, bx = x0;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const float v = x[(((int64_t)n * H + yy) * W + xx) * ldx + c];
        if (v > best) { best = v; by = yy; bx = xx; }
      }
    dx[(((int64_t)n * H + by) * W + bx) * lddx + c] += dy[(((int64_t)n * Ho + oy) * Wo + ox) * lddy + c];   // windows are disjoint
  }
}

__global__ void zero_strided_kernel(int64_t npix, int C, float* __restrict__ y, int ldy) {
  const int64_t total = npix * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) y[(i / C) * ldy + i % C] = 0.f;
}

// ---------------------------------------------------------------- resize
// V = channels per thread (4: float4 accesses when C, the channel strides and the base pointers allow it; 1: any layout)
template <int V> struct ChanVec;
template <> struct ChanVec<1> {
  float v[1];
  __device__ static ChanVec load(const float* p) { ChanVec r; r.v[0] = *p; return r; }
  __device__ void store(float* p) const { *p = v[0]; }
};
template <> struct ChanVec<4> {
  float v[4];
  __device__ static ChanVec load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    ChanVec r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
  }
  __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

template <int V>
__global__ __launch_bounds__(256) void resize_fwd_kernel(int N, int hc, int wc, int C, int Ho, int Wo, const float* __restrict__ x, int ldx,
                                  const int32_t* __restrict__ iy, const float* __restrict__ wy, const int32_t* __restrict__ ix,
                                  const float* __restrict__ wx, float alpha, float beta, float* __restrict__ y, int ldy) {
  const int CV = C / V;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V; int64_t r = i / CV; const int ox = r % Wo; r /= Wo; const int oy = r % Ho; const int n = r / Ho;
    float wxb[4]; int ixb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { wxb[b] = wx[ox * 4 + b]; ixb[b] = ix[ox * 4 + b]; }
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float wya = wy[oy * 4 + a];
      if (wya == 0.f) continue;
      const float* row = x + ((int64_t)n * hc + iy[oy * 4 + a]) * wc * ldx + c;
      float t[V];
#pragma unroll
      for (int j = 0; j < V; ++j) t[j] = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (wxb[b] != 0.f) {
          const ChanVec<V> q = ChanVec<V>::load(row + (int64_t)ixb[b] * ldx);
#pragma unroll
          for (int j = 0; j < V; ++j) t[j] += wxb[b] * q.v[j];
        }
      }
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += wya * t[j];
    }
    float* dst = &y[(((int64_t)n * Ho + oy) * Wo + ox) * ldy + c];
    ChanVec<V> o;
    if (beta == 0.f) {
#pragma unroll
      for (int j = 0; j < V; ++j) o.v[j] = alpha * acc[j];
    } else {
      const ChanVec<V> old = ChanVec<V>::load(dst);
#pragma unroll
      for (int j = 0; j < V; ++j) o.v[j] = beta * old.v[j] + alpha * acc[j];
    }
    o.store(dst);
  }
}

// y pass of the two-pass float4 form (the x pass - resize_x_kernel below - has interpolated the hc coarse rows to the output width): output row
// index uniform per workgroup, thread = (output column, channel quad), a loop over RPB rows whose 4 row taps are scalar loads; two rows per
// pass, all of a pass's loads (2 x 4 taps + the two old values of the accumulate mode) issued before its two stores (a load between stores
// is waited for with vmcnt(0)).  Zero-weight taps carry index 0: fetched, unused.
constexpr int RESIZE_RPB = 8;
__global__ __launch_bounds__(256) void resize_fwd_y_kernel(int N, int hc, int C, int Ho, int Wo, const float* __restrict__ t, const int32_t* __restrict__ iy,
                                                           const float* __restrict__ wy, float alpha, float beta, float* __restrict__ y, int ldy) {
  const int CV = C >> 2;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= Wo * CV) return;
  const int c = (u % CV) << 2, ox = u / CV;
  const int chunks = (Ho + RESIZE_RPB - 1) / RESIZE_RPB;
  const int n = blockIdx.y / chunks, oy0 = (blockIdx.y % chunks) * RESIZE_RPB;
  const float* tn = t + ((int64_t)n * hc * Wo + ox) * C + c;
  const bool rmw = beta != 0.f;
  for (int oy = oy0; oy < oy0 + RESIZE_RPB && oy < Ho; oy += 2) {
    const int oyb = oy + 1 < Ho ? oy + 1 : oy;                 // odd Ho: the second row of the last pass repeats the first (not stored)
    float4 q[2][4], old[2];
    float wya[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = rr ? oyb : oy;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        wya[rr][a] = wy[row * 4 + a];
        q[rr][a] = *reinterpret_cast<const float4*>(tn + (int64_t)iy[row * 4 + a] * Wo * C);
      }
    }
    float* dst[2] = {y + (((int64_t)n * Ho + oy) * Wo + ox) * ldy + c, y + (((int64_t)n * Ho + oyb) * Wo + ox) * ldy + c};
    if (rmw) { old[0] = *reinterpret_cast<const float4*>(dst[0]); old[1] = *reinterpret_cast<const float4*>(dst[1]); }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[0] += wya[rr][a] * q[rr][a].x; acc[1] += wya[rr][a] * q[rr][a].y; acc[2] += wya[rr][a] * q[rr][a].z; acc[3] += wya[rr][a] * q[rr][a].w;
      }
      float4 o;
      if (rmw) o = make_float4(beta * old[rr].x + alpha * acc[0], beta * old[rr].y + alpha * acc[1], beta * old[rr].z + alpha * acc[2], beta * old[rr].w + alpha * acc[3]);
      else o = make_float4(alpha * acc[0], alpha * acc[1], alpha * acc[2], alpha * acc[3]);
      if (rr == 0 || oyb != oy) *reinterpret_cast<float4*>(dst[rr]) = o;
    }
  }
}

// The same row pass for NS sources at once (round 6: the resize branches of the eight-branch merge, models/Homogeneous_Poisson_NN_Legacy.py:215-224): the
// destination is read and written ONCE for all of them.  The accumulation runs in the order and with the roundings of NS consecutive calls of the kernel
// above - y = beta y + alpha r0, then y = 1 y + alpha r1, ... - so the result is bit-identical to them.
struct ResizeYSrc { const float* t; const int32_t* iy; const float* wy; int hc; };
template <int NS>
__global__ __launch_bounds__(256) void resize_fwd_y_multi_kernel(int N, int C, int Ho, int Wo, ResizeYSrc s0, ResizeYSrc s1, ResizeYSrc s2, float alpha, float beta,
                                                                 float* __restrict__ y, int ldy) {
  const ResizeYSrc src[3] = {s0, s1, s2};
  const int CV = C >> 2;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= Wo * CV) return;
  const int c = (u % CV) << 2, ox = u / CV;
  const int chunks = (Ho + RESIZE_RPB - 1) / RESIZE_RPB;
  const int n = blockIdx.y / chunks, oy0 = (blockIdx.y % chunks) * RESIZE_RPB;
  const bool rmw = beta != 0.f;
  for (int oy = oy0; oy < oy0 + RESIZE_RPB && oy < Ho; ++oy) {
    float4 q[NS][4], old;
    float w[NS][4];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const float* tn = src[k].t + ((int64_t)n * src[k].hc * Wo + ox) * C + c;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        w[k][a] = src[k].wy[oy * 4 + a];
        q[k][a] = *reinterpret_cast<const float4*>(tn + (int64_t)src[k].iy[oy * 4 + a] * Wo * C);
      }
    }
    float* dst = y + (((int64_t)n * Ho + oy) * Wo + ox) * ldy + c;
    if (rmw) old = *reinterpret_cast<const float4*>(dst);
    float4 o;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[0] += w[k][a] * q[k][a].x; acc[1] += w[k][a] * q[k][a].y; acc[2] += w[k][a] * q[k][a].z; acc[3] += w[k][a] * q[k][a].w;
      }
      if (k == 0) {
        if (rmw) o = make_float4(beta * old.x + alpha * acc[0], beta * old.y + alpha * acc[1], beta * old.z + alpha * acc[2], beta * old.w + alpha * acc[3]);
        else o = make_float4(alpha * acc[0], alpha * acc[1], alpha * acc[2], alpha * acc[3]);
      } else {
        o = make_float4(1.f * o.x + alpha * acc[0], 1.f * o.y + alpha * acc[1], 1.f * o.z + alpha * acc[2], 1.f * o.w + alpha * acc[3]);
      }
    }
    *reinterpret_cast<float4*>(dst) = o;
  }
}

// x pass of the two-pass form: t[n, yc, ox, c] = sum_b wx[ox][b] x[n, yc, ix[ox][b], c] on the COARSE rows only (hc of them); the row-uniform
// kernel above then needs 4 row taps per output instead of 16 taps - it was bound by the texture path (16 cache-resident loads per 16 bytes
// written), not by HBM
__global__ __launch_bounds__(256) void resize_x_kernel(int64_t rows, int wc, int C, int Wo, const float* __restrict__ x, int ldx, const int32_t* __restrict__ ix,
                                                       const float* __restrict__ wx, float* __restrict__ t) {
  const int CV = C >> 2;
  const int64_t total = rows * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) << 2; int64_t r = i / CV; const int ox = (int)(r % Wo); const int64_t row = r / Wo;
    const float* src = x + row * wc * ldx + c;
    float4 q[4]; float w[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { w[b] = wx[ox * 4 + b]; q[b] = *reinterpret_cast<const float4*>(src + (int64_t)ix[ox * 4 + b] * ldx); }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < 4; ++b) { o.x += w[b] * q[b].x; o.y += w[b] * q[b].y; o.z += w[b] * q[b].z; o.w += w[b] * q[b].w; }
    *reinterpret_cast<float4*>(t + (row * Wo + ox) * C + c) = o;
  }
}

// pass 1 of the adjoint: tmp[n, yc, X, c] = sum_Y Ry[Y, yc] dy[n, Y, X, c]
template <int V>
__global__ __launch_bounds__(256) void resize_bwd_rows_kernel(int N, int hc, int C, int Ho, int Wo, const float* __restrict__ dy, int lddy,
                                       const int32_t* __restrict__ iy, const float* __restrict__ wy, float* __restrict__ tmp) {
  const int CV = C / V;
  const int64_t total = (int64_t)N * hc * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V; int64_t r = i / CV; const int X = r % Wo; r /= Wo; const int yc = r % hc; const int n = r / hc;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    // only output rows whose (<= 4) taps can reach yc: conservative window around yc * (Ho/hc)
    const float sc = (float)Ho / (float)hc;
    int Ya = (int)floorf(((float)yc - 2.5f) * sc) - 1, Yb = (int)ceilf(((float)yc + 3.5f) * sc) + 2;
    if (Ya < 0) Ya = 0;
    if (Yb > Ho) Yb = Ho;
    for (int Y = Ya; Y < Yb; ++Y) {
      float wsum = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) wsum += (iy[Y * 4 + a] == yc) ? wy[Y * 4 + a] : 0.f;
      if (wsum != 0.f) {
        const ChanVec<V> q = ChanVec<V>::load(dy + (((int64_t)n * Ho + Y) * Wo + X) * lddy + c);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += wsum * q.v[j];
      }
    }
    ChanVec<V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = acc[j];
    o.store(tmp + (((int64_t)n * hc + yc) * Wo + X) * C + c);
  }
}

// float4 form of pass 1, streaming: a workgroup owns RB consecutive coarse rows (uniform), thread = (output column X, channel quad) with RB
// accumulators; it walks the fine rows that can reach those coarse rows ONCE, every row one coalesced 16-byte load per thread (two rows in
// flight), the row's four taps being scalar data: w_k = sum_a [iy[Y][a] == yc0 + k] wy[Y][a].  (The per-coarse-row form above walks the
// same window once per coarse row, a chain of dependent, branch-guarded loads: 0.98 ms for 1 GB at 8 x 1024^2.)
template <int RESIZE_RB>
__global__ __launch_bounds__(256) void resize_bwd_rows_block_kernel(int N, int hc, int C, int Ho, int Wo, const float* __restrict__ dy, int lddy,
                                                                    const int32_t* __restrict__ iy, const float* __restrict__ wy, float* __restrict__ tmp) {
  const int CV = C >> 2;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= Wo * CV) return;
  const int c = (u % CV) << 2, X = u / CV;
  const int groups = (hc + RESIZE_RB - 1) / RESIZE_RB;
  const int n = blockIdx.y / groups, yc0 = (blockIdx.y % groups) * RESIZE_RB;
  const float sc = (float)Ho / (float)hc;
  int Ya = (int)floorf(((float)yc0 - 2.5f) * sc) - 1, Yb = (int)ceilf(((float)(yc0 + RESIZE_RB - 1) + 3.5f) * sc) + 2;
  if (Ya < 0) Ya = 0;
  if (Yb > Ho) Yb = Ho;
  float4 acc[RESIZE_RB];
#pragma unroll
  for (int k = 0; k < RESIZE_RB; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* src = dy + ((int64_t)n * Ho * Wo + X) * lddy + c;
  for (int Y = Ya; Y < Yb; Y += 2) {
    const int Y1 = Y + 1 < Yb ? Y + 1 : Y;
    const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)Y * Wo * lddy);
    const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)Y1 * Wo * lddy);
    float w0[RESIZE_RB], w1[RESIZE_RB];
#pragma unroll
    for (int k = 0; k < RESIZE_RB; ++k) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        s0 += (iy[Y * 4 + a] == yc0 + k) ? wy[Y * 4 + a] : 0.f;
        s1 += (iy[Y1 * 4 + a] == yc0 + k) ? wy[Y1 * 4 + a] : 0.f;
      }
      w0[k] = s0; w1[k] = Y1 != Y ? s1 : 0.f;
    }
#pragma unroll
    for (int k = 0; k < RESIZE_RB; ++k) {
      acc[k].x += w0[k] * v0.x; acc[k].y += w0[k] * v0.y; acc[k].z += w0[k] * v0.z; acc[k].w += w0[k] * v0.w;
      acc[k].x += w1[k] * v1.x; acc[k].y += w1[k] * v1.y; acc[k].z += w1[k] * v1.z; acc[k].w += w1[k] * v1.w;
    }
  }
#pragma unroll
  for (int k = 0; k < RESIZE_RB; ++k)
    if (yc0 + k < hc) *reinterpret_cast<float4*>(tmp + (((int64_t)n * hc + yc0 + k) * Wo + X) * C + c) = acc[k];
}

// pass 2: dx[n, yc, xc, c] = alpha * sum_X Rx[X, xc] tmp[n, yc, X, c]
template <int V>
__global__ __launch_bounds__(256) void resize_bwd_cols_kernel(int N, int hc, int wc, int C, int Wo, const float* __restrict__ tmp, const int32_t* __restrict__ ix,
                                       const float* __restrict__ wx, float alpha, float* __restrict__ dx, int lddx) {
  const int CV = C / V;
  const int64_t total = (int64_t)N * hc * wc * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V; int64_t r = i / CV; const int xc = r % wc; r /= wc; const int yc = r % hc; const int n = r / hc;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    const float sc = (float)Wo / (float)wc;
    int Xa = (int)floorf(((float)xc - 2.5f) * sc) - 1, Xb = (int)ceilf(((float)xc + 3.5f) * sc) + 2;
    if (Xa < 0) Xa = 0;
    if (Xb > Wo) Xb = Wo;
    for (int X = Xa; X < Xb; ++X) {
      float wsum = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) wsum += (ix[X * 4 + b] == xc) ? wx[X * 4 + b] : 0.f;
      if (wsum != 0.f) {
        const ChanVec<V> q = ChanVec<V>::load(tmp + (((int64_t)n * hc + yc) * Wo + X) * C + c);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += wsum * q.v[j];
      }
    }
    ChanVec<V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = alpha * acc[j];
    o.store(dx + (((int64_t)n * hc + yc) * wc + xc) * lddx + c);
  }
}

static bool vec4_ok(int C, const void* a, int lda, const void* b, int ldb) {
  return C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

static dim3 grid1d(int64_t total, int block = 256, int maxb = 16384) {
  int64_t b = pcnn_cdiv64(total, block);
  if (b > maxb) b = maxb;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}
static int pow2_ge(int c) { int p = 1; while (p < c) p <<= 1; return p; }

}  // namespace

extern "C" int pcnn_pool2d_fwd(pcnn_handle h, int kind, int N, int H, int W, int C, int f, const float* x, int ldx, float* y, int ldy) {
  PCNN_REQUIRE(h, h && x && y && f >= 1 && (kind == 0 || kind == 1), "pcnn_pool2d_fwd: bad argument");
  const int Ho = pcnn_cdiv(H, f), Wo = pcnn_cdiv(W, f), pby = (Ho * f - H) / 2, pbx = (Wo * f - W) / 2;
  if (f * f <= 16 || C > 256) {
    if (vec4_ok(C, x, ldx, y, ldy))
      hipLaunchKernelGGL(pool_fwd_small_vec4_kernel, grid1d((int64_t)N * Ho * Wo * (C / 4)), dim3(256), 0, h->stream, kind, N, H, W, C, f, Ho, Wo, pby, pbx, x, ldx, y, ldy);
    else
      hipLaunchKernelGGL(pool_fwd_small_kernel, grid1d((int64_t)N * Ho * Wo * C), dim3(256), 0, h->stream, kind, N, H, W, C, f, Ho, Wo, pby, pbx, x, ldx, y, ldy);
  } else {
    hipLaunchKernelGGL(pool_fwd_large_kernel, dim3((unsigned)((int64_t)N * Ho * Wo)), dim3(256), 0, h->stream, kind, N, H, W, C, pow2_ge(C), f, Ho,
                       Wo, pby, pbx, x, ldx, y, ldy);
  }
  PCNN_CHECK_LAUNCH(h, "pcnn_pool2d_fwd");
  return 0;
}

extern "C" int pcnn_pool2d_bwd(pcnn_handle h, int kind, int N, int H, int W, int C, int f, const float* x, int ldx, const float* y, int ldy,
                               const float* dy, int lddy, float* dx, int lddx, int accumulate) {
  PCNN_REQUIRE(h, h && dy && dx && f >= 1 && (kind == 0 || kind == 1), "pcnn_pool2d_bwd: bad argument");
  (void)y; (void)ldy;
  const int Ho = pcnn_cdiv(H, f), Wo = pcnn_cdiv(W, f), pby = (Ho * f - H) / 2, pbx = (Wo * f - W) / 2;
  if (kind == PCNN_POOL_AVERAGE) {
    if (vec4_ok(C, dy, lddy, dx, lddx))
      hipLaunchKernelGGL(pool_bwd_avg_kernel<4>, grid1d((int64_t)N * H * W * (C / 4)), dim3(256), 0, h->stream, N, H, W, C, f, Ho, Wo, pby, pbx, dy, lddy, dx,
                         lddx, accumulate);
    else
      hipLaunchKernelGGL(pool_bwd_avg_kernel<1>, grid1d((int64_t)N * H * W * C), dim3(256), 0, h->stream, N, H, W, C, f, Ho, Wo, pby, pbx, dy, lddy, dx, lddx,
                         accumulate);
  } else {
    PCNN_REQUIRE(h, x, "pcnn_pool2d_bwd: max pooling needs the forward input");
    if (!accumulate) {
      hipLaunchKernelGGL(zero_strided_kernel, grid1d((int64_t)N * H * W * C), dim3(256), 0, h->stream, (int64_t)N * H * W, C, dx, lddx);
    }
    hipLaunchKernelGGL(pool_bwd_max_kernel, grid1d((int64_t)N * Ho * Wo * C), dim3(256), 0, h->stream, N, H, W, C, f, Ho, Wo, pby, pbx, x, ldx, dy, lddy, dx,
                       lddx);
  }
  PCNN_CHECK_LAUNCH(h, "pcnn_pool2d_bwd");
  return 0;
}

// ---- host: interpolation tables in the float32 arithmetic of TF's resize kernels
// (tensorflow/core/kernels/image/resize_{nearest_neighbor,bilinear,bicubic}_op.cc, half_pixel_centers=true)
extern "C" int pcnn_resize_tables(int method, int n_in, int n_out, int32_t* idx, float* wt) {
  if (!idx || !wt || n_in < 1 || n_out < 1) return 1;
  const float scale = (float)n_in / (float)n_out;
  static std::vector<float> table;
  if (method == PCNN_RESIZE_BICUBIC && table.empty()) {
    const float a = -0.5f;
    table.resize((1024 + 1) * 2);
    for (int i = 0; i <= 1024; ++i) {
      float x = (float)i * 1.0f / 1024.0f;
      table[2 * i] = ((a + 2) * x - (a + 3)) * x * x + 1;
      x += 1.0f;
      table[2 * i + 1] = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a;
    }
  }
  for (int o = 0; o < n_out; ++o) {
    int32_t* I = idx + 4 * o; float* Wt = wt + 4 * o;
    for (int k = 0; k < 4; ++k) { I[k] = 0; Wt[k] = 0.f; }
    if (method == PCNN_RESIZE_NEAREST) {
      int i = (int)floorf(((float)o + 0.5f) * scale);
      I[0] = i < 0 ? 0 : (i > n_in - 1 ? n_in - 1 : i); Wt[0] = 1.f;
    } else if (method == PCNN_RESIZE_BILINEAR) {
      const float src = ((float)o + 0.5f) * scale - 0.5f;
      const float fl = floorf(src);
      const int lo = fl > 0.f ? (int)fl : 0;
      const int hi = (int)fminf(ceilf(src), (float)(n_in - 1));
      const float lerp = src - fl;
      I[0] = lo; Wt[0] = 1.0f - lerp; I[1] = hi; Wt[1] = lerp;
    } else if (method == PCNN_RESIZE_BICUBIC) {
      const float src = ((float)o + 0.5f) * scale - 0.5f;
      const int loc = (int)floorf(src);
      const float delta = src - (float)loc;
      const int off = (int)lrintf(delta * 1024.0f);
      float w4[4] = {table[off * 2 + 1], table[off * 2], table[(1024 - off) * 2], table[(1024 - off) * 2 + 1]};
      float sum = 0.f;
      for (int k = 0; k < 4; ++k) {
        const int raw = loc - 1 + k;
        const int cl = raw < 0 ? 0 : (raw > n_in - 1 ? n_in - 1 : raw);
        if (cl != raw) w4[k] = 0.f;
        I[k] = cl; sum += w4[k];
      }
      if (fabsf(sum) >= 1000.0f * 1.17549435e-38f) {
        const float inv = 1.0f / sum;
        for (int k = 0; k < 4; ++k) w4[k] *= inv;
      }
      for (int k = 0; k < 4; ++k) Wt[k] = w4[k];
    } else if (method == PCNN_RESIZE_BICUBIC_LEGACY_ALIGN_CORNERS) {
      // tf.compat.v1.image.resize_images(BICUBIC, align_corners=True): legacy scaler, a = -0.75, clamped taps
      static std::vector<float> legacy;
      if (legacy.empty()) {
        const float a = -0.75f;
        legacy.resize((1024 + 1) * 2);
        for (int i = 0; i <= 1024; ++i) {
          float x = (float)i * 1.0f / 1024.0f;
          legacy[2 * i] = ((a + 2) * x - (a + 3)) * x * x + 1;
          x += 1.0f;
          legacy[2 * i + 1] = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a;
        }
      }
      const float sc = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : (float)n_in / (float)n_out;
      const float src = (float)o * sc;
      const int loc = (int)floorf(src);
      const int off = (int)lrintf((src - (float)loc) * 1024.0f);
      const float w4[4] = {legacy[off * 2 + 1], legacy[off * 2], legacy[(1024 - off) * 2], legacy[(1024 - off) * 2 + 1]};
      for (int k = 0; k < 4; ++k) {
        const int raw = loc - 1 + k;
        I[k] = raw < 0 ? 0 : (raw > n_in - 1 ? n_in - 1 : raw);
        Wt[k] = w4[k];
      }
    } else {
      return 2;
    }
  }
  return 0;
}

extern "C" int pcnn_resize_fwd(pcnn_handle h, int N, int hc, int wc, int C, int Ho, int Wo, const float* x, int ldx, const int32_t* idx_y,
                               const float* wt_y, const int32_t* idx_x, const float* wt_x, float alpha, float beta, float* y, int ldy) {
  PCNN_REQUIRE(h, h && x && y && idx_y && wt_y && idx_x && wt_x, "pcnn_resize_fwd: null argument");
  if (vec4_ok(C, x, ldx, y, ldy) && (int64_t)N * ((Ho + RESIZE_RPB - 1) / RESIZE_RPB) < 65536) {
    // two passes: x interpolation of the hc coarse rows into the handle's scratch, then the row-uniform y pass
    const size_t need = (size_t)N * hc * Wo * C * sizeof(float);
    if (h->aux_ws_bytes < need) {
      if (h->aux_ws) { pcnn_release(h, h->aux_ws); h->aux_ws = nullptr; h->aux_ws_bytes = 0; }
      if (hipMalloc(&h->aux_ws, need) != hipSuccess) PCNN_FAIL(h, "pcnn_resize_fwd: cannot allocate %zu B of scratch", need);
      h->aux_ws_bytes = need;
    }
    float* t = static_cast<float*>(h->aux_ws);
    hipLaunchKernelGGL(resize_x_kernel, grid1d((int64_t)N * hc * Wo * (C / 4)), dim3(256), 0, h->stream, (int64_t)N * hc, wc, C, Wo, x, ldx, idx_x, wt_x, t);
    hipLaunchKernelGGL(resize_fwd_y_kernel, dim3((unsigned)((Wo * (C / 4) + 255) / 256), (unsigned)(N * ((Ho + RESIZE_RPB - 1) / RESIZE_RPB))), dim3(256), 0, h->stream,
                       N, hc, C, Ho, Wo, t, idx_y, wt_y, alpha, beta, y, ldy);
  }
  else if (vec4_ok(C, x, ldx, y, ldy))
    hipLaunchKernelGGL(resize_fwd_kernel<4>, grid1d((int64_t)N * Ho * Wo * (C / 4)), dim3(256), 0, h->stream, N, hc, wc, C, Ho, Wo, x, ldx, idx_y, wt_y, idx_x,
                       wt_x, alpha, beta, y, ldy);
  else
    hipLaunchKernelGGL(resize_fwd_kernel<1>, grid1d((int64_t)N * Ho * Wo * C), dim3(256), 0, h->stream, N, hc, wc, C, Ho, Wo, x, ldx, idx_y, wt_y, idx_x, wt_x,
                       alpha, beta, y, ldy);
  PCNN_CHECK_LAUNCH(h, "pcnn_resize_fwd");
  return 0;
}

// y = beta y + alpha (resize(x_0) + resize(x_1) [+ resize(x_2)]) with ONE pass over y - bit-identical to nsrc calls of pcnn_resize_fwd (beta, then 1, 1).
extern "C" int pcnn_resize_fwd_multi_eligible(int N, int C, int Ho, int Wo, int nsrc, const pcnn_resize_src* src, const float* y, int ldy) {
  if (!src || !y || (nsrc != 2 && nsrc != 3) || C < 4 || C % 4 != 0 || ldy % 4 != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return 0;
  if ((int64_t)N * ((Ho + RESIZE_RPB - 1) / RESIZE_RPB) >= 65536) return 0;
  for (int k = 0; k < nsrc; ++k)
    if (!src[k].x || !src[k].idx_y || !src[k].wt_y || !src[k].idx_x || !src[k].wt_x || src[k].ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(src[k].x) & 15) != 0 ||
        src[k].hc < 1 || src[k].wc < 1) return 0;
  return 1;
}
extern "C" int pcnn_resize_fwd_multi(pcnn_handle h, int N, int C, int Ho, int Wo, int nsrc, const pcnn_resize_src* src, float alpha, float beta, float* y, int ldy) {
  PCNN_REQUIRE(h, h && pcnn_resize_fwd_multi_eligible(N, C, Ho, Wo, nsrc, src, y, ldy), "pcnn_resize_fwd_multi: not eligible (ask pcnn_resize_fwd_multi_eligible first)");
  size_t off[4] = {0, 0, 0, 0};
  for (int k = 0; k < nsrc; ++k) off[k + 1] = off[k] + (((size_t)N * src[k].hc * Wo * C * sizeof(float) + 255) & ~(size_t)255);
  if (h->aux_ws_bytes < off[nsrc]) {
    if (h->aux_ws) { pcnn_release(h, h->aux_ws); h->aux_ws = nullptr; h->aux_ws_bytes = 0; }
    if (hipMalloc(&h->aux_ws, off[nsrc]) != hipSuccess) PCNN_FAIL(h, "pcnn_resize_fwd_multi: cannot allocate %zu B of scratch", off[nsrc]);
    h->aux_ws_bytes = off[nsrc];
  }
  ResizeYSrc ys[3] = {};
  for (int k = 0; k < nsrc; ++k) {
    float* t = reinterpret_cast<float*>(static_cast<char*>(h->aux_ws) + off[k]);
    hipLaunchKernelGGL(resize_x_kernel, grid1d((int64_t)N * src[k].hc * Wo * (C / 4)), dim3(256), 0, h->stream, (int64_t)N * src[k].hc, src[k].wc, C, Wo, src[k].x, src[k].ldx,
                       src[k].idx_x, src[k].wt_x, t);
    ys[k] = ResizeYSrc{t, src[k].idx_y, src[k].wt_y, src[k].hc};
  }
  const dim3 grid((unsigned)((Wo * (C / 4) + 255) / 256), (unsigned)(N * ((Ho + RESIZE_RPB - 1) / RESIZE_RPB)));
  if (nsrc == 2) hipLaunchKernelGGL(resize_fwd_y_multi_kernel<2>, grid, dim3(256), 0, h->stream, N, C, Ho, Wo, ys[0], ys[1], ys[1], alpha, beta, y, ldy);
  else hipLaunchKernelGGL(resize_fwd_y_multi_kernel<3>, grid, dim3(256), 0, h->stream, N, C, Ho, Wo, ys[0], ys[1], ys[2], alpha, beta, y, ldy);
  PCNN_CHECK_LAUNCH(h, "pcnn_resize_fwd_multi");
  return 0;
}

extern "C" int pcnn_resize_bwd(pcnn_handle h, int N, int hc, int wc, int C, int Ho, int Wo, const float* dy, int lddy, const int32_t* idx_y,
                               const float* wt_y, const int32_t* idx_x, const float* wt_x, float alpha, float* tmp, float* dx, int lddx) {
  PCNN_REQUIRE(h, h && dy && dx && tmp && idx_y && wt_y && idx_x && wt_x, "pcnn_resize_bwd: null argument");
  // coarse rows per workgroup: 8 where there are many (the window's margins - 3 coarse rows on either side - are re-read less), 4 otherwise
  const int rb = hc >= 32 ? 8 : 4;
  if (vec4_ok(C, dy, lddy, tmp, 4) && (int64_t)N * ((hc + rb - 1) / rb) < 65536) {
    const dim3 grid((unsigned)((Wo * (C / 4) + 255) / 256), (unsigned)(N * ((hc + rb - 1) / rb)));
    if (rb == 8) hipLaunchKernelGGL(resize_bwd_rows_block_kernel<8>, grid, dim3(256), 0, h->stream, N, hc, C, Ho, Wo, dy, lddy, idx_y, wt_y, tmp);
    else hipLaunchKernelGGL(resize_bwd_rows_block_kernel<4>, grid, dim3(256), 0, h->stream, N, hc, C, Ho, Wo, dy, lddy, idx_y, wt_y, tmp);
  } else if (vec4_ok(C, dy, lddy, tmp, 4))
    hipLaunchKernelGGL(resize_bwd_rows_kernel<4>, grid1d((int64_t)N * hc * Wo * (C / 4)), dim3(256), 0, h->stream, N, hc, C, Ho, Wo, dy, lddy, idx_y, wt_y, tmp);
  else
    hipLaunchKernelGGL(resize_bwd_rows_kernel<1>, grid1d((int64_t)N * hc * Wo * C), dim3(256), 0, h->stream, N, hc, C, Ho, Wo, dy, lddy, idx_y, wt_y, tmp);
  PCNN_CHECK_LAUNCH(h, "pcnn_resize_bwd(rows)");
  if (vec4_ok(C, tmp, 4, dx, lddx))
    hipLaunchKernelGGL(resize_bwd_cols_kernel<4>, grid1d((int64_t)N * hc * wc * (C / 4)), dim3(256), 0, h->stream, N, hc, wc, C, Wo, tmp, idx_x, wt_x, alpha, dx, lddx);
  else
    hipLaunchKernelGGL(resize_bwd_cols_kernel<1>, grid1d((int64_t)N * hc * wc * C), dim3(256), 0, h->stream, N, hc, wc, C, Wo, tmp, idx_x, wt_x, alpha, dx, lddx);
  PCNN_CHECK_LAUNCH(h, "pcnn_resize_bwd(cols)");
  return 0;
}
