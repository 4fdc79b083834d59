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
  int tiles_x, tiles_y, ntiles, S, KWG, lg;
  int vecx, vecdz;
};

struct WgradPlan { int S, KWG, TAPS, MTC, NTC, gz, lg; size_t lds; int ntiles, tiles_x, tiles_y; };

static WgradPlan make_plan(const pcnn_conv_desc* d) {
  WgradPlan pl;
  // The (filter column, input channel) index space of one filter row is CONTIGUOUS in the dense NHWC x tile
  // ((x + kj)*Cin + ci = x*Cin + (kj*Cin + ci)), so it is cut into 32-row MFMA tiles irrespective of Cin: no channel
  // padding of the M dimension, and tiles are dealt round-robin to the 4 waves (<= 4 tiles + 4 running totals each).
  pl.MTC = 1; pl.lg = 0;
  pl.NTC = pcnn_cdiv(d->Cout, 32);
  int maxTiles = 16 / pl.NTC;
  if (maxTiles < 4) maxTiles = 4;
  int kwg = (maxTiles * 32) / d->Cin;               // filter columns per workgroup
  if (kwg < 1) kwg = 1;
  pl.KWG = d->kw < kwg ? d->kw : kwg;
  pl.TAPS = pcnn_cdiv(pcnn_cdiv(pl.KWG * d->Cin, 32), WAVES);
  pl.gz = pcnn_cdiv(d->kw, pl.KWG);
  pl.tiles_x = pcnn_cdiv(d->Wo, WTW); pl.tiles_y = pcnn_cdiv(d->Ho, WTH);
  pl.ntiles = d->N * pl.tiles_x * pl.tiles_y;
  pl.lds = ((size_t)WTH * (WTW + pl.KWG - 1) * d->Cin + (size_t)WTH * WTW * d->Cout + 128) * 4;
  // All workgroups of this grid run for about the same time, so the grid is sized to fill the chip's resident slots
  // (256 CUs x workgroups per CU by LDS) an integral number of times: a 2055-workgroup grid on 512 slots costs 5 rounds.
  const int per_cu = pl.lds * 2 <= 160 * 1024 ? 2 : 1;
  const int slots = 256 * per_cu, rounds = 4;
  int S = (slots * rounds) / (d->kh * pl.gz);
  if (S > pl.ntiles) S = pl.ntiles;
  if (S < 1) S = 1;
  pl.S = S;
  return pl;
}

template <int NTC, int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int split = blockIdx.x, ki = blockIdx.y, kj0 = blockIdx.z * p.KWG;
  const int kwg = min(p.KWG, p.kw - kj0);
  const int TCx = WTW + p.KWG - 1;
  float* xs = lds;
  float* dzs = lds + ((WTH * TCx * p.Cin + 64 + 3) & ~3);

  // blocked summation: `acc` holds one tile (256 pixels), `tot` the running sum over the workgroup's tiles
  f32x16 acc[TAPS][NTC], tot[TAPS][NTC];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int q = 0; q < NTC; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[t][q][i] = 0.f; tot[t][q][i] = 0.f; }

  for (int tile = split; tile < p.ntiles; tile += p.S) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * WTH, x0 = tx * WTW;
    const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
    const float* dzin = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
    __syncthreads();
    // ---- stage x rows (shifted by the filter row, BC padding applied) and the dz tile (zero outside the image).
    // Fast path (16-byte aligned channel runs): 8 loads per lane are issued back to back from always-valid (select-ed)
    // addresses and only afterwards padded and written to LDS, so 8 global loads are in flight per lane instead of one
    // load -> wait -> store round trip per element.  Slow path (Cin or Cout not a multiple of 4): scalar elements.
    // Interior fast path: the whole x window of this tile lies inside the image and pixels are dense (ldx == Cin), so
    // every tile row is one contiguous run of TCx*Cin floats in HBM and in LDS: no per-element index arithmetic at all.
    const int wy0 = y0 + ki - p.pt, wx0 = x0 + kj0 - p.pl;
    const bool fast_x = p.vecx && p.ldx == p.Cin && wy0 >= 0 && wy0 + WTH <= p.H && wx0 >= 0 && wx0 + TCx <= p.W;
    if (fast_x) {
      const int upr = (TCx * p.Cin) >> 2;
      for (int r = wave; r < WTH; r += WAVES) {
        const f32x4* src = reinterpret_cast<const f32x4*>(xin + ((int64_t)(wy0 + r) * p.W + wx0) * p.Cin);
        f32x4* dst = reinterpret_cast<f32x4*>(xs + r * TCx * p.Cin);
        for (int u0 = 0; u0 < upr; u0 += 64 * 8) {
          f32x4 v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) { const int u = u0 + 64 * q + lane; v[q] = src[u < upr ? u : 0]; }
#pragma unroll
          for (int q = 0; q < 8; ++q) { const int u = u0 + 64 * q + lane; if (u < upr) dst[u] = v[q]; }
        }
      }
    } else if (p.vecx) {
      const int G = p.Cin >> 2;                        // float4 units per pixel
      const int upr = TCx * G;                         // units per tile row
      const int total = WTH * upr;
      const float inv_upr = 1.0f / (float)upr, inv_G = 1.0f / (float)G;   // exact small-integer division via (u + 0.5) * 1/d
      for (int base = tid; base < total; base += 256 * 8) {
        f32x4 v[8];
        int off[8];                                     // LDS float offset, or -1: skip, or (-2 - offset): padding value
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int u = base + 256 * q;
          const int uu = u < total ? u : 0;
          const int r = (int)(((float)uu + 0.5f) * inv_upr);
          const int w = uu - r * upr;
          const int c = (int)(((float)w + 0.5f) * inv_G);
          const int ch = (w - c * G) << 2;
          const int sy = pcnn_pad_index(y0 + r + ki - p.pt, p.H, p.pad_mode);
          const int sx = pcnn_pad_index(x0 + c + kj0 - p.pl, p.W, p.pad_mode);
          const bool inimg = sy >= 0 && sx >= 0;
          v[q] = *reinterpret_cast<const f32x4*>(xin + ((int64_t)(inimg ? sy : 0) * p.W + (inimg ? sx : 0)) * p.ldx + ch);
          const int o = (r * TCx + c) * p.Cin + ch;
          off[q] = u < total ? (inimg ? o : -2 - o) : -1;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (off[q] != -1) {
            f32x4 t = v[q];
            int o = off[q];
            if (o < 0) { o = -2 - o; const float pv = p.pad_value; t[0] = pv; t[1] = pv; t[2] = pv; t[3] = pv; }
            *reinterpret_cast<f32x4*>(&xs[o]) = t;
          }
        }
      }
    } else {
      const int units = TCx * p.Cin;
      for (int r = wave; r < WTH; r += WAVES) {
        const int sy = pcnn_pad_index(y0 + r + ki - p.pt, p.H, p.pad_mode);
        for (int e = lane; e < units; e += 64) {
          const int c = e / p.Cin, ch = e - c * p.Cin;
          const int sx = pcnn_pad_index(x0 + c + kj0 - p.pl, p.W, p.pad_mode);
          xs[(r * TCx + c) * p.Cin + ch] = (sy < 0 || sx < 0) ? p.pad_value : xin[((int64_t)sy * p.W + sx) * p.ldx + ch];
        }
      }
    }
    const bool fast_dz = p.vecdz && p.lddz == p.Cout && y0 + WTH <= p.Ho && x0 + WTW <= p.Wo;
    if (fast_dz) {
      const int upr = (WTW * p.Cout) >> 2;
      for (int r = wave; r < WTH; r += WAVES) {
        const f32x4* src = reinterpret_cast<const f32x4*>(dzin + ((int64_t)(y0 + r) * p.Wo + x0) * p.Cout);
        f32x4* dst = reinterpret_cast<f32x4*>(dzs + r * WTW * p.Cout);
        for (int u0 = 0; u0 < upr; u0 += 64 * 4) {
          f32x4 v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int u = u0 + 64 * q + lane; v[q] = src[u < upr ? u : 0]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int u = u0 + 64 * q + lane; if (u < upr) dst[u] = v[q]; }
        }
      }
    } else if (p.vecdz) {
      const int G = p.Cout >> 2;
      const int upr = WTW * G;
      const int total = WTH * upr;
      const float inv_upr = 1.0f / (float)upr, inv_G = 1.0f / (float)G;
      for (int base = tid; base < total; base += 256 * 8) {
        f32x4 v[8];
        int off[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int u = base + 256 * q;
          const int uu = u < total ? u : 0;
          const int r = (int)(((float)uu + 0.5f) * inv_upr);
          const int w = uu - r * upr;
          const int c = (int)(((float)w + 0.5f) * inv_G);
          const int ch = (w - c * G) << 2;
          const int oy = y0 + r, ox = x0 + c;
          const bool inimg = oy < p.Ho && ox < p.Wo;
          v[q] = *reinterpret_cast<const f32x4*>(dzin + ((int64_t)(inimg ? oy : 0) * p.Wo + (inimg ? ox : 0)) * p.lddz + ch);
          const int o = (r * WTW + c) * p.Cout + ch;
          off[q] = u < total ? (inimg ? o : -2 - o) : -1;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (off[q] != -1) {
            f32x4 t = v[q];
            int o = off[q];
            if (o < 0) { o = -2 - o; t[0] = 0.f; t[1] = 0.f; t[2] = 0.f; t[3] = 0.f; }
            *reinterpret_cast<f32x4*>(&dzs[o]) = t;
          }
        }
      }
    } else {
      const int units = WTW * p.Cout;
      for (int r = wave; r < WTH; r += WAVES) {
        const int oy = y0 + r;
        for (int e = lane; e < units; e += 64) {
          const int c = e / p.Cout, ch = e - c * p.Cout;
          const int ox = x0 + c;
          dzs[(r * WTW + c) * p.Cout + ch] = (oy >= p.Ho || ox >= p.Wo) ? 0.f : dzin[((int64_t)oy * p.Wo + ox) * p.lddz + ch];
        }
      }
    }
    __syncthreads();
    // ---- accumulate: K = pixels of the tile, two per MFMA.  Blocks of 8 pixel pairs: the dz fragments (B) are read once
    // and reused by every filter column this wave owns; the x fragments (A) are read in batches of 8 so the MFMAs of a
    // column issue back to back.  `ntw` is wave-uniform (scalar branch, no exec masking).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ntile = (kwg * p.Cin + 31) >> 5;                 // 32-row tiles over the flattened (column, channel) space
    const int ntw = ntile > wv ? (ntile - wv + WAVES - 1) / WAVES : 0;
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
            const float* ap = &xs[(r * TCx + xp0 + half) * p.Cin + (wv + WAVES * t) * 32 + col];
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = ap[2 * u * p.Cin];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
              for (int q = 0; q < NTC; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][q], acc[t][q], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int q = 0; q < NTC; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) { tot[t][q][i] += acc[t][q][i]; acc[t][q][i] = 0.f; }
  }
  // ---- write partials ws[split][ki][kj][ci][co]; C row R of tile jt -> flattened index jt*32 + R = kj*Cin + ci
  {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nflat = kwg * p.Cin;
    float* dst = p.ws + (((int64_t)split * p.kh + ki) * p.kw + kj0) * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int jt = wv + WAVES * t;
      if (jt * 32 >= nflat) continue;
#pragma unroll
      for (int q = 0; q < NTC; ++q) {
        const int co = q * 32 + col;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int R = jt * 32 + 8 * (i >> 2) + 4 * half + (i & 3);
          if (R < nflat && co < p.Cout) dst[(int64_t)R * p.Cout + co] = tot[t][q][i];   // [kj][ci][co] is contiguous in R
        }
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

template <int NTC, int TAPS>
void launch_wgrad(pcnn_handle h, const WgradParams& p, const WgradPlan& pl) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<NTC, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((wgrad_kernel<NTC, TAPS>), dim3(pl.S, p.kh, pl.gz), dim3(256), pl.lds, h->stream, p);
}


// ============================================================================================================================
// 3 x fp16 split variant (PCNN_MATH_SPLIT_F16; numerics: conv_fwd_split.hip).  The reduction runs over pixels, so both MFMA
// operands need 8 consecutive K values = 8 pixels per lane.  The tile (8 rows x 32 columns) is therefore staged TRANSPOSED:
// one 16-byte LDS unit = the 8 rows of one (column, channel) as fp16, units ordered [column][channel] - which keeps the
// flattened (filter column, channel) trick: unit(x, R) = x*Cin + R is contiguous in R, so an A fragment is 32 consecutive
// units (conflict-free ds_read_b128) and no channel padding is needed.  One K step (16 pixels) = 2 columns x 8 rows: lanes
// 0-31 take column 2t, lanes 32-63 column 2t+1.  Each loader task gathers the 8 rows of 4 channels of one column (8 dwordx4
// loads), converts to hi/lo and writes 4 + 4 units.  x and dz are scaled per tile by powers of two (block max), the tile's
// accumulator is rescaled when folded into the running total.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ void pow2_scale(float mx, float& s, float& inv_s) {
  int e = 0;
  if (mx > 0.f) (void)frexpf(mx, &e);
  s = mx > 0.f ? ldexpf(1.0f, 13 - e) : 1.0f;
  inv_s = mx > 0.f ? ldexpf(1.0f, e - 13) : 1.0f;
}

template <int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int split = blockIdx.x, ki = blockIdx.y, kj0 = blockIdx.z * p.KWG;
  const int kwg = min(p.KWG, p.kw - kj0);
  const int TCx = WTW + p.KWG - 1;
  // units per plane; fragment over-reads (< 32 units past a plane, garbage rows/columns that are discarded) land in the next
  // plane, only the last plane needs 512 B of slack - every byte counts: k = 15, 32 channels must fit twice into 160 KiB
  const int nux = TCx * p.Cin, nudz = WTW * p.Cout;
  f16x8* xh = reinterpret_cast<f16x8*>(lds);
  f16x8* xl = xh + nux;
  f16x8* zh = xl + nux;
  f16x8* zl = zh + nudz;
  float* red = reinterpret_cast<float*>(zl + nudz + 32);             // 8 floats, after the over-read slack

  f32x16 acc[TAPS], tot[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[t][i] = 0.f; tot[t][i] = 0.f; }

  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int ntile = (kwg * p.Cin + 31) >> 5;
  const int ntw = ntile > wv ? (ntile - wv + WAVES - 1) / WAVES : 0;
  const int GX = p.Cin >> 2, GZ = p.Cout >> 2;                        // channel quads
  const int ntask_x = TCx * GX, ntask_z = WTW * GZ;

  for (int tile = split; tile < p.ntiles; tile += p.S) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * WTH, x0 = tx * WTW;
    const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
    const float* dzin = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
    // source row offsets (wave-uniform): x rows are shifted by the filter row, BC padding applied
    int64_t xrow[WTH]; bool xrow_in[WTH];
#pragma unroll
    for (int r = 0; r < WTH; ++r) {
      const int sy = pcnn_pad_index(y0 + r + ki - p.pt, p.H, p.pad_mode);
      xrow_in[r] = sy >= 0;
      xrow[r] = (int64_t)(sy >= 0 ? sy : 0) * p.W;
    }
    // ---- pass 1: block maxima of the x window and the dz tile
    float mxx = 0.f, mxz = 0.f;
    for (int task = tid; task < ntask_x; task += 256) {
      const int c = task / GX, ch = (task - c * GX) << 2;
      const int sx = pcnn_pad_index(x0 + c + kj0 - p.pl, p.W, p.pad_mode);
#pragma unroll
      for (int r = 0; r < WTH; ++r) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xin + (xrow[r] + (sx >= 0 ? sx : 0)) * p.ldx + ch);
        if (!(xrow_in[r] && sx >= 0)) { const float pv = p.pad_value; v = (f32x4){pv, pv, pv, pv}; }
        mxx = fmaxf(mxx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      }
    }
    for (int task = tid; task < ntask_z; task += 256) {
      const int c = task / GZ, ch = (task - c * GZ) << 2;
      const int ox = x0 + c;
#pragma unroll
      for (int r = 0; r < WTH; ++r) {
        const bool in = y0 + r < p.Ho && ox < p.Wo;
        f32x4 v = *reinterpret_cast<const f32x4*>(dzin + ((int64_t)(in ? y0 + r : 0) * p.Wo + (in ? ox : 0)) * p.lddz + ch);
        if (in) mxz = fmaxf(mxz, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      }
    }
    mxx = wg_wave_max(mxx); mxz = wg_wave_max(mxz);
    __syncthreads();                                   // previous tile's fragment reads are done
    if (lane == 0) { red[wave] = mxx; red[4 + wave] = mxz; }
    __syncthreads();
    mxx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    mxz = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    float sx_, isx, sz_, isz;
    pow2_scale(mxx, sx_, isx); pow2_scale(mxz, sz_, isz);
    // ---- pass 2: reload (L1/L2 hits), split into fp16 hi/lo, write transposed units
    for (int task = tid; task < ntask_x; task += 256) {
      const int c = task / GX, ch = (task - c * GX) << 2;
      const int sx = pcnn_pad_index(x0 + c + kj0 - p.pl, p.W, p.pad_mode);
      f32x4 v[WTH];
#pragma unroll
      for (int r = 0; r < WTH; ++r) {
        v[r] = *reinterpret_cast<const f32x4*>(xin + (xrow[r] + (sx >= 0 ? sx : 0)) * p.ldx + ch);
        if (!(xrow_in[r] && sx >= 0)) { const float pv = p.pad_value; v[r] = (f32x4){pv, pv, pv, pv}; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f16x8 h8, l8;
#pragma unroll
        for (int r = 0; r < WTH; ++r) { const float a = v[r][j] * sx_; const _Float16 ah = (_Float16)a; h8[r] = ah; l8[r] = (_Float16)(a - (float)ah); }
        xh[c * p.Cin + ch + j] = h8; xl[c * p.Cin + ch + j] = l8;
      }
    }
    for (int task = tid; task < ntask_z; task += 256) {
      const int c = task / GZ, ch = (task - c * GZ) << 2;
      const int ox = x0 + c;
      f32x4 v[WTH];
#pragma unroll
      for (int r = 0; r < WTH; ++r) {
        const bool in = y0 + r < p.Ho && ox < p.Wo;
        v[r] = *reinterpret_cast<const f32x4*>(dzin + ((int64_t)(in ? y0 + r : 0) * p.Wo + (in ? ox : 0)) * p.lddz + ch);
        if (!in) v[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f16x8 h8, l8;
#pragma unroll
        for (int r = 0; r < WTH; ++r) { const float a = v[r][j] * sz_; const _Float16 ah = (_Float16)a; h8[r] = ah; l8[r] = (_Float16)(a - (float)ah); }
        zh[c * p.Cout + ch + j] = h8; zl[c * p.Cout + ch + j] = l8;
      }
    }
    __syncthreads();
    // ---- accumulate: 16 K steps of 2 columns x 8 rows
#pragma unroll 2
    for (int xp = 0; xp < WTW / 2; ++xp) {
      const int cx = 2 * xp + half;
      const f16x8 bh = zh[cx * p.Cout + col], bl = zl[cx * p.Cout + col];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t < ntw) {
          const int u = cx * p.Cin + (wv + WAVES * t) * 32 + col;
          const f16x8 ah = xh[u], al = xl[u];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
        }
      }
    }
    const float rs = isx * isz;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) { tot[t][i] = fmaf(acc[t][i], rs, tot[t][i]); acc[t][i] = 0.f; }
  }
  // ---- write partials (same layout as the fp32 kernel)
  {
    const int nflat = kwg * p.Cin;
    float* dst = p.ws + (((int64_t)split * p.kh + ki) * p.kw + kj0) * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int jt = wv + WAVES * t;
      if (jt * 32 >= nflat) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int R = jt * 32 + 8 * (i >> 2) + 4 * half + (i & 3);
        if (R < nflat && col < p.Cout) dst[(int64_t)R * p.Cout + col] = tot[t][i];
      }
    }
  }
}

template <int TAPS>
void launch_wgrad_split(pcnn_handle h, const WgradParams& p, const WgradPlan& pl, size_t lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_kernel<TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((wgrad_split_kernel<TAPS>), dim3(pl.S, p.kh, pl.gz), dim3(256), lds, h->stream, p);
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
  PCNN_REQUIRE(h, d->Cin >= 1 && d->Cin <= 128 && d->Cout >= 1 && d->Cout <= 64, "pcnn_conv2d_wgrad: channels %d->%d unsupported (<=64)", d->Cin, d->Cout);
  PCNN_REQUIRE(h, d->ldx >= d->Cin && d->ldy >= d->Cout, "pcnn_conv2d_wgrad: channel stride smaller than channel count");
  WgradPlan pl = make_plan(d);
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_conv2d_wgrad_workspace(d), "pcnn_conv2d_wgrad: workspace too small");
  PCNN_REQUIRE(h, pl.lds <= 160 * 1024, "pcnn_conv2d_wgrad: tile needs %zu B of LDS", pl.lds);
  WgradParams p;
  p.x = x; p.dz = dz; p.ws = static_cast<float*>(workspace);
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.lddz = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.S = pl.S; p.KWG = pl.KWG; p.lg = pl.lg;
  p.vecx = (d->Cin % 4 == 0) && (d->ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.vecdz = (d->Cout % 4 == 0) && (d->ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15) == 0);
  const bool split_ok = h->math_mode == PCNN_MATH_SPLIT_F16 && p.vecx && p.vecdz && pl.NTC == 1;
  if (split_ok) {
    const size_t lds = ((size_t)(WTW + pl.KWG - 1) * d->Cin + (size_t)WTW * d->Cout) * 32 + 512 + 64;
    switch (pl.TAPS) {
      case 1: launch_wgrad_split<1>(h, p, pl, lds); break;
      case 2: launch_wgrad_split<2>(h, p, pl, lds); break;
      case 3: launch_wgrad_split<3>(h, p, pl, lds); break;
      default: launch_wgrad_split<4>(h, p, pl, lds); break;
    }
  } else
#define PCNN_WG(Q, T) if (pl.NTC == Q && pl.TAPS == T) { launch_wgrad<Q, T>(h, p, pl); } else
  PCNN_WG(1, 1) PCNN_WG(1, 2) PCNN_WG(1, 3) PCNN_WG(1, 4) PCNN_WG(2, 1) PCNN_WG(2, 2)
  { PCNN_FAIL(h, "pcnn_conv2d_wgrad: no kernel for NTC=%d TAPS=%d", pl.NTC, pl.TAPS); }
#undef PCNN_WG
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad");
  const int64_t nel = (int64_t)d->kh * d->kw * d->Cin * d->Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(nel, 256), 2048)), dim3(256), 0, h->stream,
                     p.ws, dw, nel, pl.S);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(reduce)");
  return 0;
}
