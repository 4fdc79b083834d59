// Transposed convolution with kernel == stride as per-tap 32x32 GEMMs on the fp32 matrix cores.
//   fwd      : y[n, yc f+ty-py, xc f+tx-px, co] = beta*y + alpha*(bias[co] + sum_ci x[n,yc,xc,ci] K[ty,tx,co,ci])
//   bwd data : dx[n,yc,xc,ci] = alpha * sum_{ty,tx,co} dy[fine(ty,tx)][co] K[ty,tx,co,ci]
//   bwd filt : dK[ty,tx,co,ci] = alpha * sum_{n,yc,xc} dy[fine(ty,tx)][co] x[n,yc,xc,ci]
// The op is HBM-bound (1024 MAC per output pixel, AI ~ 15 FLOP/B): every kernel streams the full-resolution tensor exactly
// once with 128-byte rows per pixel and keeps the coarse tensor / filter in registers or L2; no LDS staging is needed
// because the MFMA operand fragments are contiguous channel runs of single pixels.
// One wave owns 32 consecutive coarse pixels of one coarse row (fwd / bwd data) so all index arithmetic is wave-uniform.
#include "pcnn_internal.h"

namespace {

struct DeconvParams {
  int N, hc, wc, Cin, H, W, Cout, f, py, px;
  const float* x; int ldx;         // coarse tensor (fwd input / bwd-filter input)
  const float* wp;                 // packed filter
  const float* bias;
  float alpha, beta;
  float* y; int ldy;               // fine tensor (fwd output)
  const float* dy; int lddy;       // fine gradient
  float* dx; int lddx;             // coarse gradient
  float* partial;                  // bwd-filter partials [S][f*f][Cout][Cin]
  int tiles_x, S;
  float* dummy;                    // fwd: 64 floats that absorb the stores (and serve the loads) of lanes / pixels outside the output
};

// K (f,f,Cout,Cin) -> [tap][g][h][co32][4]: element j = K[tap][co][8g+4h+j]            (B operand of fwd: k = ci, n = co)
__global__ void pack_deconv_fwd_kernel(const float* __restrict__ k, float* __restrict__ wp, int taps, int Cin, int Cout, int ng) {
  const int64_t total = (int64_t)taps * ng * 2 * 32 * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = i & 3; int64_t r = i >> 2; const int co = r & 31; r >>= 5; const int h = r & 1; r >>= 1; const int g = r % ng; const int tap = r / ng;
    const int ci = 8 * g + 4 * h + j;
    wp[i] = (ci < Cin && co < Cout) ? k[((int64_t)tap * Cout + co) * Cin + ci] : 0.f;
  }
}
// K (f,f,Cout,Cin) -> [tap][g][h][ci32][4]: element j = K[tap][8g+4h+j][ci]            (B operand of bwd data: k = co, n = ci)
__global__ void pack_deconv_bwd_kernel(const float* __restrict__ k, float* __restrict__ wp, int taps, int Cin, int Cout, int ng) {
  const int64_t total = (int64_t)taps * ng * 2 * 32 * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = i & 3; int64_t r = i >> 2; const int ci = r & 31; r >>= 5; const int h = r & 1; r >>= 1; const int g = r % ng; const int tap = r / ng;
    const int co = 8 * g + 4 * h + j;
    wp[i] = (ci < Cin && co < Cout) ? k[((int64_t)tap * Cout + co) * Cin + ci] : 0.f;
  }
}

// ---------------------------------------------------------------- forward
// The tap loop holds NO lane-dependent control flow: accesses of lanes / pixels outside the output go to a per-lane dummy word instead of being
// predicated (this compiler puts even a single predicated store behind a branch, and behind a branch the number of memory operations is not
// static: every wait then covers everything outstanding).  With a static count the accumulate mode (RMW: the branch merge, beta != 0) can request
// the NEXT tap's sixteen old values BEFORE this tap's stores are issued - the memory counter is in-order, a load issued behind stores is waited for
// until those stores have completed: one exposed write latency per tap in the form that loaded a tap's old values at the top of its own pass.
template <bool RMW>
__global__ __launch_bounds__(256) void deconv_fwd_mfma_kernel(DeconvParams p) {
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  int wid = blockIdx.x * 4 + (threadIdx.x >> 6);               // wave -> (n, yc, x tile, filter row ty): one fine output row per wave, so
  const int total = p.N * p.hc * p.tiles_x * p.f;               // that a 16x up-sampling of a 64^2 grid still fills the chip (16 384 waves)
  if (wid >= total) return;
  const int ty = wid % p.f; wid /= p.f;
  const int tx0 = wid % p.tiles_x; wid /= p.tiles_x;
  const int yc = wid % p.hc, n = wid / p.hc;
  const int x0 = tx0 * 32;
  const int ng = (p.Cin + 7) >> 3;
  const int Y = yc * p.f + ty - p.py;
  if (Y < 0 || Y >= p.H) return;
  // A fragments: this lane's coarse pixel, 4 channels per 8-channel group (lane half selects which 4)
  f32x4 a[4];
  const int xc_l = x0 + col;
  const bool pv = xc_l < p.wc;
  const float* xsrc = p.x + (((int64_t)n * p.hc + yc) * p.wc + (pv ? xc_l : p.wc - 1)) * p.ldx;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ch = 8 * g + 4 * half;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g < ng && pv && ch < p.Cin) v = *reinterpret_cast<const f32x4*>(xsrc + ch);
    a[g] = v;
  }
  const float bias = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
  const f32x4* wl = reinterpret_cast<const f32x4*>(p.wp) + half * 32 + col;
  float* const yrow = p.y + (((int64_t)n * p.H + Y) * p.W) * p.ldy + col;
  // (the dummy word as an OFFSET from the row: a select between two pointers is turned back into a branch around the access, a select between two
  // integer offsets of one base is a v_cndmask)
  const int64_t doff = (p.dummy + lane) - yrow;
  // element i of this lane for tap tx: fine pixel X = xc f + tx - px of coarse pixel xc = x0 + 8 (i >> 2) + 4 half + (i & 3)
  auto where = [&](int i, int tx) -> float* {
    const int xc = x0 + 8 * (i >> 2) + 4 * half + (i & 3);
    const int X = xc * p.f + tx - p.px;
    const bool ok = col < p.Cout && xc < p.wc && X >= 0 && X < p.W;
    int64_t off = ok ? (int64_t)X * p.ldy : doff;
    asm volatile("" : "+v"(off));                              // opaque: otherwise the optimiser re-derives the two pointers and branches around the access
    return yrow + off;
  };
  // the filter fragments of a tap (B operands: 4 channel groups x 16 bytes per lane; groups beyond ng re-read the last one - their A operands are zero)
  auto load_b = [&](int tx, f32x4 (&b)[4]) {
    const int tp = ty * p.f + tx;
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = wl[(int64_t)(tp * ng + min(g, ng - 1)) * 64];
  };
  float old[16];
  f32x4 b[4];
  load_b(0, b);
  if (RMW) {
#pragma unroll
    for (int i = 0; i < 16; ++i) old[i] = *where(i, 0);
  }
  auto tap = [&](int tx) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][j], b[g][j], acc, 0, 0, 0);
    float out[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = p.alpha * (acc[i] + bias);
      out[i] = RMW ? p.beta * old[i] + v : v;
    }
    // the next tap's operands (the last tap re-reads its own): BEFORE this tap's stores.  (A filter load inside the MFMA loop used to be waited for
    // with vmcnt(0) - four times per tap everything outstanding, the previous tap's stores included.)
    const int tn = min(tx + 1, p.f - 1);
    load_b(tn, b);
    if (RMW) {
#pragma unroll
      for (int i = 0; i < 16; ++i) old[i] = *where(i, tn);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) *where(i, tx) = out[i];
  };
  // first tap outside the loop: a pass inside it then always starts from the same memory-counter state
  tap(0);
  for (int tx = 1; tx < p.f; ++tx) tap(tx);
}

// ---------------------------------------------------------------- backward data
// One WORKGROUP owns 32 consecutive coarse pixels of one coarse row; its four waves split the f x f taps (wave w: filter rows ty = w, w + 4, ...) and
// their partial sums are added in a fixed order through LDS - a 128 x up-sampling of an 8 x 8 grid used to be 64 waves of 16 384 taps each on a
// 256-CU chip.  The tap loop is a two-stage software pipeline: the fine-gradient fragments (4 x 16 bytes per lane) and filter fragments of the NEXT
// tap are requested before the current tap's sixteen MFMAs (they used to be loaded one channel group at a time, each waited for where it was issued:
// four exposed latencies per tap); loads of lanes / rows outside the gradient re-read a valid address and are zeroed when consumed.
__global__ __launch_bounds__(256) void deconv_bwd_data_mfma_kernel(DeconvParams p) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int wid = blockIdx.x;
  const int tx0 = wid % p.tiles_x; wid /= p.tiles_x;
  const int yc = wid % p.hc, n = wid / p.hc;
  const int x0 = tx0 * 32;
  const int ng = (p.Cout + 7) >> 3;       // K dimension = co
  const int xc_l = x0 + col;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const f32x4* wl = reinterpret_cast<const f32x4*>(p.wp) + half * 32 + col;
  const float* img = p.dy + (int64_t)n * p.H * p.W * p.lddy;
  // this wave's taps in order: (ty, tx), ty = wave, wave + 4, ...; tx = 0 .. f - 1
  const int nrows = wave < p.f ? (p.f - wave + 3) / 4 : 0, ntap = nrows * p.f;
  auto load_tap = [&](int t, f32x4 (&a)[4], f32x4 (&b)[4], bool& ok) {
    const int ty = wave + 4 * (t / p.f), tx = t % p.f;
    const int Y = yc * p.f + ty - p.py, X = xc_l * p.f + tx - p.px;
    ok = xc_l < p.wc && X >= 0 && X < p.W && Y >= 0 && Y < p.H;
    const float* src = img + ((int64_t)(ok ? Y : 0) * p.W + (ok ? X : 0)) * p.lddy;
    const int tap = ty * p.f + tx;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = min(8 * g + 4 * half, p.Cout - 4);     // (Cout % 4 == 0: checked by the caller)
      a[g] = *reinterpret_cast<const f32x4*>(src + ch);
      b[g] = wl[(int64_t)(tap * ng + min(g, ng - 1)) * 64];
    }
  };
  auto mfma_tap = [&](const f32x4 (&a)[4], const f32x4 (&b)[4], bool ok) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bool use = ok && g < ng && 8 * g + 4 * half < p.Cout;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(use ? a[g][j] : 0.f, b[g][j], acc, 0, 0, 0);
    }
  };
  f32x4 a0[4], b0[4], a1[4], b1[4];
  bool ok0 = false, ok1 = false;
  if (ntap > 0) load_tap(0, a0, b0, ok0);
  for (int t = 0; t < ntap; t += 2) {
    load_tap(min(t + 1, ntap - 1), a1, b1, ok1);
    mfma_tap(a0, b0, ok0);
    load_tap(min(t + 2, ntap - 1), a0, b0, ok0);
    mfma_tap(a1, b1, ok1 && t + 1 < ntap);
  }
  // cross-wave sum in a fixed order
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
  __syncthreads();
  if (wave == 0 && col < p.Cin) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int xc = x0 + 8 * (i >> 2) + 4 * half + (i & 3);
      const float s = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
      if (xc < p.wc) p.dx[(((int64_t)n * p.hc + yc) * p.wc + xc) * p.lddx + col] = p.alpha * s;
    }
  }
}

// ---------------------------------------------------------------- backward filter
// grid (S, f*f); each wave walks coarse rows r = (split*4 + wave), +4S, ... ; M = co, N = ci, K = coarse pixels (2 per MFMA)
__global__ __launch_bounds__(256) void deconv_bwd_filter_mfma_kernel(DeconvParams p) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31, wave = threadIdx.x >> 6;
  const int tap = blockIdx.y, ty = tap / p.f, tx = tap % p.f;
  const int rows = p.N * p.hc;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // (measured and rejected: a flattened two-stage software pipeline over (row, eight-pixel step) - the index arithmetic per four MFMAs costs more than
  // the exposed loads: 0.38 -> 0.52 ms per launch)
  for (int r = blockIdx.x * 4 + wave; r < rows; r += p.S * 4) {
    const int yc = r % p.hc, n = r / p.hc;
    const int Y = yc * p.f + ty - p.py;
    if (Y < 0 || Y >= p.H) continue;
    const float* dyrow = p.dy + ((int64_t)n * p.H + Y) * p.W * p.lddy;
    const float* xrow = p.x + ((int64_t)n * p.hc + yc) * p.wc * p.ldx;
    for (int xc0 = 0; xc0 < p.wc; xc0 += 8) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int xc = xc0 + 2 * u + half;
        const int X = xc * p.f + tx - p.px;
        const bool v = xc < p.wc && X >= 0 && X < p.W;
        a[u] = (v && col < p.Cout) ? dyrow[(int64_t)X * p.lddy + col] : 0.f;
        b[u] = (v && col < p.Cin) ? xrow[(int64_t)xc * p.ldx + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
  }
  // cross-wave sum (fixed order) and store: row = co, col = ci
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
  __syncthreads();
  if (wave == 0) {
    float* dst = p.partial + ((int64_t)blockIdx.x * p.f * p.f + tap) * p.Cout * p.Cin;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float s = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
      const int co = 8 * (i >> 2) + 4 * half + (i & 3);
      if (co < p.Cout && col < p.Cin) dst[co * p.Cin + col] = s;
    }
  }
}

static bool ensure_scratch(pcnn_handle h, size_t need) {
  if (h->scratch_bytes >= need) return true;
  if (h->scratch) { pcnn_release(h, h->scratch); h->scratch = nullptr; h->scratch_bytes = 0; }
  const size_t cap = need < (4u << 20) ? (4u << 20) : need;
  if (hipMalloc(&h->scratch, cap) != hipSuccess) return false;
  h->scratch_bytes = cap;
  return true;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Returns 0 on success, -1 if the shape is outside what the MFMA path covers (caller falls back), >0 on error.
int pcnn_deconv_fwd_mfma(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx, const float* k,
                         const float* bias, float alpha, float beta, float* y, int ldy) {
  if (Cin > 32 || Cout > 32 || Cin % 4 || ldx % 4 || !aligned16(x)) return -1;
  const int ng = (Cin + 7) >> 3;
  const size_t need = (size_t)f * f * ng * 64 * 4 * sizeof(float);                 // packed filter, then 64 dummy floats (deconv_fwd_mfma_kernel)
  if (!ensure_scratch(h, need + 64 * sizeof(float))) PCNN_FAIL(h, "pcnn_deconv_fwd: cannot allocate filter scratch");
  const int64_t tot = (int64_t)f * f * ng * 64 * 4;
  hipLaunchKernelGGL(pack_deconv_fwd_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(tot, 256), 1024)), dim3(256), 0, h->stream, k,
                     static_cast<float*>(h->scratch), f * f, Cin, Cout, ng);
  DeconvParams p{};
  p.N = N; p.hc = hc; p.wc = wc; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.f = f; p.py = (hc * f - H) / 2; p.px = (wc * f - W) / 2;
  p.x = x; p.ldx = ldx; p.wp = static_cast<const float*>(h->scratch); p.bias = bias; p.alpha = alpha; p.beta = beta; p.y = y; p.ldy = ldy;
  p.tiles_x = pcnn_cdiv(wc, 32);
  p.dummy = static_cast<float*>(h->scratch) + need / sizeof(float);
  const int64_t waves = (int64_t)N * hc * p.tiles_x * f;
  if (beta != 0.f) hipLaunchKernelGGL(deconv_fwd_mfma_kernel<true>, dim3((unsigned)pcnn_cdiv64(waves, 4)), dim3(256), 0, h->stream, p);
  else hipLaunchKernelGGL(deconv_fwd_mfma_kernel<false>, dim3((unsigned)pcnn_cdiv64(waves, 4)), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_deconv_fwd(mfma)");
  return 0;
}

int pcnn_deconv_bwd_data_mfma(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* dy, int lddy, const float* k,
                              float alpha, float* dx, int lddx) {
  if (Cin > 32 || Cout > 32 || Cout % 4 || lddy % 4 || !aligned16(dy)) return -1;
  const int ng = (Cout + 7) >> 3;
  const size_t need = (size_t)f * f * ng * 64 * 4 * sizeof(float);
  if (!ensure_scratch(h, need)) PCNN_FAIL(h, "pcnn_deconv_bwd_data: cannot allocate filter scratch");
  const int64_t tot = (int64_t)f * f * ng * 64 * 4;
  hipLaunchKernelGGL(pack_deconv_bwd_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(tot, 256), 1024)), dim3(256), 0, h->stream, k,
                     static_cast<float*>(h->scratch), f * f, Cin, Cout, ng);
  DeconvParams p{};
  p.N = N; p.hc = hc; p.wc = wc; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.f = f; p.py = (hc * f - H) / 2; p.px = (wc * f - W) / 2;
  p.dy = dy; p.lddy = lddy; p.wp = static_cast<const float*>(h->scratch); p.alpha = alpha; p.dx = dx; p.lddx = lddx;
  p.tiles_x = pcnn_cdiv(wc, 32);
  const int64_t groups = (int64_t)N * hc * p.tiles_x;            // one workgroup (four waves sharing the taps) per 32 coarse pixels of a row
  hipLaunchKernelGGL(deconv_bwd_data_mfma_kernel, dim3((unsigned)groups), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_deconv_bwd_data(mfma)");
  return 0;
}

// partial must hold S*f*f*Cout*Cin floats with S = pcnn_deconv_bwd_filter_splits(...)
int pcnn_deconv_bwd_filter_splits(int N, int hc, int f) {
  int S = 1024 / (f * f);
  const int rows4 = (N * hc + 3) / 4;
  if (S > rows4) S = rows4;
  if (S < 1) S = 1;
  if (S > 256) S = 256;
  return S;
}

int pcnn_deconv_bwd_filter_mfma(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx, const float* dy,
                                int lddy, float* partial, int* S_out) {
  if (Cin > 32 || Cout > 32) return -1;
  DeconvParams p{};
  p.N = N; p.hc = hc; p.wc = wc; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.f = f; p.py = (hc * f - H) / 2; p.px = (wc * f - W) / 2;
  p.x = x; p.ldx = ldx; p.dy = dy; p.lddy = lddy; p.partial = partial;
  p.S = pcnn_deconv_bwd_filter_splits(N, hc, f);
  *S_out = p.S;
  hipLaunchKernelGGL(deconv_bwd_filter_mfma_kernel, dim3(p.S, f * f), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_deconv_bwd_filter(mfma)");
  return 0;
}
