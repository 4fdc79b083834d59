// Filter gradient of the fused pad+conv (tf.nn.conv2d backprop-filter) as an fp32 MFMA GEMM:
//   dw[i,j,ci,co] = sum_{n,y,x} xpad[n, y-pt+i, x-pl+j, ci] * dz[n,y,x,co]
// GEMM view per tap (i,j): M = ci, N = co, K = pixels.  v_mfma_f32_32x32x2_f32 consumes 2 pixels per issue; the A
// operand is a 32-channel run of one pixel of x (lane = channel -> contiguous 128 B LDS read, conflict free) and the
// B operand the 32-channel run of the same pixel of dz, so both come from NHWC tiles with no transposition.
//
// Decomposition: grid = (pixel split s, filter row i, column group).  A workgroup walks the output tiles
// s, s+S, ... (8 rows x 32 columns each), stages the x rows shifted by i and the dz tile in LDS, and its 4 waves
// own the filter columns j = w, w+4, ... (<= 4 taps, one 32x32 accumulator each), so accumulators never leave
// registers until the end.  Partials go to a workspace [S][kh][kw][Cin][Cout]; a second kernel sums over S in a
// fixed order (bit-reproducible run to run - no float atomics).
#include "pcnn_internal.h"

namespace {

constexpr int WTH = 8, WTW = 32, WAVES = 4;

struct WgradParams {
  const float* x; const float* dz; float* ws;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, kh, kw, pt, pl, pad_mode; float pad_value;
  int tiles_x, tiles_y, ntiles, S, KWG;
  int vecx, vecdz;
};

struct WgradPlan { int S, KWG, TAPS, MTC, NTC, gz; size_t lds; int ntiles, tiles_x, tiles_y; };

static WgradPlan make_plan(const pcnn_conv_desc* d) {
  WgradPlan pl;
  pl.MTC = pcnn_cdiv(d->Cin, 32); pl.NTC = pcnn_cdiv(d->Cout, 32);
  int maxTaps = 4 / (pl.MTC * pl.NTC);   // <= 4 accumulator tiles (+4 running totals) per wave: 128 VGPRs
  if (maxTaps > 4) maxTaps = 4;
  if (maxTaps < 1) maxTaps = 1;
  pl.KWG = d->kw < 4 * maxTaps ? d->kw : 4 * maxTaps;
  pl.TAPS = pcnn_cdiv(pl.KWG, WAVES);
  pl.gz = pcnn_cdiv(d->kw, pl.KWG);
  pl.tiles_x = pcnn_cdiv(d->Wo, WTW); pl.tiles_y = pcnn_cdiv(d->Ho, WTH);
  pl.ntiles = d->N * pl.tiles_x * pl.tiles_y;
  int S = pcnn_cdiv(2048, d->kh * pl.gz);
  if (S > pl.ntiles) S = pl.ntiles;
  if (S < 1) S = 1;
  pl.S = S;
  pl.lds = ((size_t)WTH * (WTW + pl.KWG - 1) * d->Cin + (size_t)WTH * WTW * d->Cout + 128) * 4;
  return pl;
}

template <int MTC, int NTC, int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int split = blockIdx.x, ki = blockIdx.y, kj0 = blockIdx.z * p.KWG;
  const int kwg = min(p.KWG, p.kw - kj0);
  const int TCx = WTW + p.KWG - 1;
  float* xs = lds;
  float* dzs = lds + ((WTH * TCx * p.Cin + 64 + 3) & ~3);

  // blocked summation: `acc` holds one tile (256 pixels), `tot` the running sum over the workgroup's tiles
  f32x16 acc[TAPS][MTC][NTC], tot[TAPS][MTC][NTC];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int m = 0; m < MTC; ++m)
#pragma unroll
      for (int q = 0; q < NTC; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[t][m][q][i] = 0.f; tot[t][m][q][i] = 0.f; }

  for (int tile = split; tile < p.ntiles; tile += p.S) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * WTH, x0 = tx * WTW;
    const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
    const float* dzin = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
    __syncthreads();
    // ---- stage x rows (shifted by the filter row) with the BC padding applied
    {
      const int V = p.vecx ? 4 : 1;
      const int units = TCx * p.Cin / V;
      for (int r = wave; r < WTH; r += WAVES) {
        const int sy = pcnn_pad_index(y0 + r + ki - p.pt, p.H, p.pad_mode);
        for (int u = lane; u < units; u += 64) {
          const int e = u * V, c = e / p.Cin, ch = e - c * p.Cin;
          const int sx = pcnn_pad_index(x0 + c + kj0 - p.pl, p.W, p.pad_mode);
          float* dst = &xs[(r * TCx + c) * p.Cin + ch];
          if (sy < 0 || sx < 0) {
            for (int q = 0; q < V; ++q) dst[q] = p.pad_value;
          } else {
            const float* src = xin + ((int64_t)sy * p.W + sx) * p.ldx + ch;
            if (V == 4) *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
            else dst[0] = src[0];
          }
        }
      }
    }
    // ---- stage the dz tile (zero outside the image)
    {
      const int V = p.vecdz ? 4 : 1;
      const int units = WTW * p.Cout / V;
      for (int r = wave; r < WTH; r += WAVES) {
        const int oy = y0 + r;
        for (int u = lane; u < units; u += 64) {
          const int e = u * V, c = e / p.Cout, ch = e - c * p.Cout;
          const int ox = x0 + c;
          float* dst = &dzs[(r * WTW + c) * p.Cout + ch];
          if (oy >= p.Ho || ox >= p.Wo) {
            for (int q = 0; q < V; ++q) dst[q] = 0.f;
          } else {
            const float* src = dzin + ((int64_t)oy * p.Wo + ox) * p.lddz + ch;
            if (V == 4) *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
            else dst[0] = src[0];
          }
        }
      }
    }
    __syncthreads();
    // ---- accumulate: K = pixels of the tile, two per MFMA.  Blocks of 8 pixel pairs: the dz fragments (B) are read once
    // and reused by every filter column this wave owns; the x fragments (A) are read in batches of 8 so the MFMAs of a
    // column issue back to back.  `ntw` is wave-uniform (scalar branch, no exec masking).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ntw = kwg > wv ? (kwg - wv + WAVES - 1) / WAVES : 0;
    for (int r = 0; r < WTH; ++r) {
      for (int xp0 = 0; xp0 < WTW; xp0 += 16) {
        float b[8][NTC];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < NTC; ++q) b[u][q] = dzs[(r * WTW + xp0 + 2 * u + half) * p.Cout + q * 32 + col];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          if (t < ntw) {
            const int kj = wv + WAVES * t;
            float a[8][MTC];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
              for (int m = 0; m < MTC; ++m) a[u][m] = xs[(r * TCx + xp0 + 2 * u + half + kj) * p.Cin + m * 32 + col];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
              for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int q = 0; q < NTC; ++q) acc[t][m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u][q], acc[t][m][q], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int m = 0; m < MTC; ++m)
#pragma unroll
        for (int q = 0; q < NTC; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i) { tot[t][m][q][i] += acc[t][m][q][i]; acc[t][m][q][i] = 0.f; }
  }
  // ---- write partials ws[split][ki][kj][ci][co]
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int kj = __builtin_amdgcn_readfirstlane(wave) + WAVES * t;
    if (kj >= kwg) continue;
    float* dst = p.ws + (((int64_t)split * p.kh + ki) * p.kw + (kj0 + kj)) * p.Cin * p.Cout;
#pragma unroll
    for (int m = 0; m < MTC; ++m)
#pragma unroll
      for (int q = 0; q < NTC; ++q) {
        const int co = q * 32 + col;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ci = m * 32 + 8 * (i >> 2) + 4 * half + (i & 3);
          if (ci < p.Cin && co < p.Cout) dst[ci * p.Cout + co] = tot[t][m][q][i];
        }
      }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t nel, int S) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nel; e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += ws[(int64_t)k * nel + e];
    dw[e] = s;
  }
}

template <int MTC, int NTC, int TAPS>
void launch_wgrad(pcnn_handle h, const WgradParams& p, const WgradPlan& pl) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<MTC, NTC, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((wgrad_kernel<MTC, NTC, TAPS>), dim3(pl.S, p.kh, pl.gz), dim3(256), pl.lds, h->stream, p);
}

}  // namespace

extern "C" size_t pcnn_conv2d_wgrad_workspace(const pcnn_conv_desc* d) {
  if (!d) return 0;
  WgradPlan pl = make_plan(d);
  return (size_t)pl.S * d->kh * d->kw * d->Cin * d->Cout * sizeof(float);
}

extern "C" int pcnn_conv2d_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw,
                                 void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, h && d && x && dz && dw && workspace, "pcnn_conv2d_wgrad: null argument");
  PCNN_REQUIRE(h, d->Cin >= 1 && d->Cin <= 64 && d->Cout >= 1 && d->Cout <= 64, "pcnn_conv2d_wgrad: channels %d->%d unsupported (<=64)", d->Cin, d->Cout);
  PCNN_REQUIRE(h, d->ldx >= d->Cin && d->ldy >= d->Cout, "pcnn_conv2d_wgrad: channel stride smaller than channel count");
  WgradPlan pl = make_plan(d);
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_conv2d_wgrad_workspace(d), "pcnn_conv2d_wgrad: workspace too small");
  PCNN_REQUIRE(h, pl.lds <= 160 * 1024, "pcnn_conv2d_wgrad: tile needs %zu B of LDS", pl.lds);
  WgradParams p;
  p.x = x; p.dz = dz; p.ws = static_cast<float*>(workspace);
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.lddz = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.S = pl.S; p.KWG = pl.KWG;
  p.vecx = (d->Cin % 4 == 0) && (d->ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.vecdz = (d->Cout % 4 == 0) && (d->ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15) == 0);
#define PCNN_WG(M, Q, T) if (pl.MTC == M && pl.NTC == Q && pl.TAPS == T) { launch_wgrad<M, Q, T>(h, p, pl); } else
  PCNN_WG(1, 1, 1) PCNN_WG(1, 1, 2) PCNN_WG(1, 1, 3) PCNN_WG(1, 1, 4)
  PCNN_WG(2, 1, 1) PCNN_WG(2, 1, 2) PCNN_WG(1, 2, 1) PCNN_WG(1, 2, 2) PCNN_WG(2, 2, 1)
  { PCNN_FAIL(h, "pcnn_conv2d_wgrad: no kernel for MTC=%d NTC=%d TAPS=%d", pl.MTC, pl.NTC, pl.TAPS); }
#undef PCNN_WG
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad");
  const int64_t nel = (int64_t)d->kh * d->kw * d->Cin * d->Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(nel, 256), 2048)), dim3(256), 0, h->stream,
                     p.ws, dw, nel, pl.S);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(reduce)");
  return 0;
}
