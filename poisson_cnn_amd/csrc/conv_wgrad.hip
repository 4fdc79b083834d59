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

int pcnn_spectral_conv_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw);   // spectral_conv.hip
bool pcnn_spectral_eligible(pcnn_handle h, const pcnn_conv_desc* d, bool wgrad);
int pcnn_conv_small_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes);   // conv_small.hip
bool pcnn_conv_small_wgrad_eligible(const pcnn_conv_desc* d);
size_t pcnn_conv_small_wgrad_workspace(const pcnn_conv_desc* d);

namespace {

constexpr int WTH = 8, WTW = 32, WAVES = 4;

struct WgradParams {
  const float* x; const float* dz; float* ws;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, kh, kw, pt, pl, pad_mode; float pad_value;
  int tiles_x, tiles_y, ntiles, S, KWG, lg, gz;
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
  // splits are dealt round-robin to the 8 XCDs (see the kernels' index decode): a whole number of splits per XCD
  int S = 8 * ((slots / 8 * rounds) / (d->kh * pl.gz));
  if (S < 8) S = (slots * rounds) / (d->kh * pl.gz);
  if (S > pl.ntiles) S = pl.ntiles;
  if (S < 1) S = 1;
  pl.S = S;
  return pl;
}

template <int NTC, int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
  // XCD-aware decode: the kh*gz workgroups that sweep the SAME tiles (one per filter row / column group) sit on one XCD next to
  // each other in dispatch order, so the kh-fold re-read of every x tile is served by that XCD's L2 instead of the fabric
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, nper = p.kh * p.gz;
  const int lsplit = q / nper, rr = q - lsplit * nper;
  const int split = lsplit * 8 + xcd;
  if (split >= p.S) return;
  const int ki = rr % p.kh, kj0 = (rr / p.kh) * p.KWG;
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

// maxbits != nullptr: the partials come from the split kernel - in its units (maxbits = {max|x|, max|dz|} as float bits), with
// channels padded to multiples of 4 (Cip, Cop) and in its channel order (channel 4q + j of a pixel at position j*Cp/4 + q).
// nel counts the partials' elements (kh*kw*Cip*Cop), dw is dense [kh*kw][Cin][Cout].
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t nel, int S, const unsigned* __restrict__ maxbits,
                                    int Cin, int Cout, int Cip, int Cop, int SL) {
  float rs = 1.f;
  if (maxbits) {
    int ex = 0, ez = 0;
    const float mx = __uint_as_float(maxbits[0]), mz = __uint_as_float(maxbits[1]);
    if (mx > 0.f) (void)frexpf(mx, &ex);
    if (mz > 0.f) (void)frexpf(mz, &ez);
    const bool okx = mx > 0.f && ex > -100 && ex < 100, okz = mz > 0.f && ez > -100 && ez < 100;   // as pow2_scale()
    rs = (okx ? ldexpf(1.0f, ex - 13) : 1.0f) * (okz ? ldexpf(1.0f, ez - 13) : 1.0f);
  }
  const int GX = Cip >> 2, GZ = Cop >> 2;
  // SL slices of the split index per element (fixed summation tree): small filters would otherwise leave the chip to a handful
  // of threads walking hundreds of partials each
  __shared__ float red[256];
  const int EL = 256 / SL;                       // elements per workgroup
  const int le = threadIdx.x % EL, sl = threadIdx.x / EL;
  for (int64_t e0 = (int64_t)blockIdx.x * EL; e0 < nel; e0 += (int64_t)gridDim.x * EL) {
    const int64_t e = e0 + le;
    float s = 0.f;
    if (e < nel)
      for (int k = sl; k < S; k += SL) s += ws[(int64_t)k * nel + e];
    if (SL > 1) {
      red[threadIdx.x] = s;
      __syncthreads();
      if (sl == 0)
        for (int q = 1; q < SL; ++q) s += red[q * EL + le];
      __syncthreads();
    }
    if (sl != 0 || e >= nel) continue;
    int64_t o = e;
    if (maxbits) {
      const int pz = (int)(e % Cop); const int64_t t = e / Cop;
      const int px = (int)(t % Cip); const int64_t tap = t / Cip;
      const int ci = 4 * (px % GX) + px / GX, co = 4 * (pz % GZ) + pz / GZ;
      if (ci >= Cin || co >= Cout) continue;
      o = (tap * Cin + ci) * Cout + co;
    }
    dw[o] = s * rs;
  }
}

template <int NTC, int TAPS>
void launch_wgrad(pcnn_handle h, const WgradParams& p, const WgradPlan& pl) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<NTC, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((wgrad_kernel<NTC, TAPS>), dim3(8 * pcnn_cdiv(pl.S, 8) * p.kh * pl.gz), dim3(256), pl.lds, h->stream, p);
}


// ============================================================================================================================
// 3 x fp16 split variant (PCNN_MATH_SPLIT_F16; numerics: conv_fwd_split.hip).  The reduction runs over pixels, so both MFMA
// operands need 8 consecutive K values = 8 pixels per lane.  The tile (8 rows x 32 columns) is therefore staged TRANSPOSED:
// one 16-byte LDS unit = the 8 rows of one (column, channel) as fp16, units ordered [column][channel] - which keeps the
// flattened (filter column, channel) trick: unit(x, R) = x*Cin + R is contiguous in R, so an A fragment is 32 consecutive
// units (conflict-free ds_read_b128) and no channel padding is needed.  One K step (16 pixels) = 2 columns x 8 rows: lanes
// 0-31 take column 2t, lanes 32-63 column 2t+1.
//
// Every x element is staged kh times (once per filter row), so the fp32 -> (hi, lo) conversion is hoisted out of the tile
// loader: a pre-pass finds max|x| and max|dz| (one power-of-two scale per TENSOR - the weight gradient is a sum over the whole
// batch, so errors relative to the tensor maximum are what fp32 itself delivers) and writes dense fp16 hi/lo planes into the
// workspace; the tile loader is then 8-byte loads + an 8x4 register transpose (v_perm) + 16-byte LDS writes, no arithmetic.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ void pow2_scale(float mx, float& s, float& inv_s) {
  int e = 0;
  if (mx > 0.f) (void)frexpf(mx, &e);
  const bool ok = mx > 0.f && e > -100 && e < 100;
  s = ok ? ldexpf(1.0f, 13 - e) : 1.0f;
  inv_s = ok ? ldexpf(1.0f, e - 13) : 1.0f;
}

struct SplitPlanes {
  const float* src[2]; _Float16* hi[2]; _Float16* lo[2];
  int64_t nquad[2]; int cq[2]; int C[2]; int ld[2]; int vec[2];     // quads in total, quads per pixel, channels, source channel stride, 1 = float4-loadable, 2 = and dense
  unsigned* maxbits;
  const float* hint[2];                                              // device max|.| supplied by the caller (NULL: computed by split_absmax_kernel)
  int zN, zHo, zWo, zTY, zWp, zC, zCp;                                // dz unit planes: [N][TY = ceil(Ho/8)][Wp = 32*ceil(Wo/32)][Cp] units of 8 rows
};

// channels 4q .. 4q+3 of one pixel; channels past C read as zero (the planes are padded to a multiple of 4 channels)
__device__ __forceinline__ f32x4 load_quad(const float* __restrict__ src, int64_t qi, int cq, int C, int ld, int vec) {
  if (vec == 2) return reinterpret_cast<const f32x4*>(src)[qi];      // dense tensor: quad qi is the qi-th float4, no index arithmetic
  const int64_t pix = qi / cq; const int ch = (int)(qi - pix * cq) << 2;
  const float* q = src + pix * ld + ch;
  if (vec) return *reinterpret_cast<const f32x4*>(q);
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = ch + j < C ? q[j] : 0.f;
  return v;
}

// blockIdx.y: 0 = x, 1 = dz.  max|v| as float bits (non-negative floats order like unsigned integers).  Both pre-pass kernels
// are pure streaming: 4 independent 16-byte loads in flight per thread.
__global__ __launch_bounds__(256) void split_absmax_kernel(SplitPlanes sp) {
  const int which = blockIdx.y;
  if (sp.hint[which]) {                // the producer already knows the maximum
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const float v = fabsf(sp.hint[which][0]);
      sp.maxbits[which] = __float_as_uint(v <= 3.0e38f ? v : 3.0e38f);
    }
    return;
  }
  const float* src = sp.src[which];
  const int cq = sp.cq[which], ld = sp.ld[which], C = sp.C[which], vec = sp.vec[which];
  const int64_t nq = sp.nquad[which], stride = (int64_t)gridDim.x * 256;
  float mx = 0.f;
  for (int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t q = q0 + u * stride < nq ? q0 + u * stride : q0;
      v[u] = load_quad(src, q, cq, C, ld, vec);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
  }
  mx = wg_wave_max(mx);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (!(mx <= 3.0e38f)) mx = 3.0e38f;               // inf / nan: keep the scale finite, the result carries the nan anyway
    atomicMax(sp.maxbits + which, __float_as_uint(mx));
  }
}

__global__ __launch_bounds__(256) void split_convert_kernel(SplitPlanes sp) {
  const int which = blockIdx.y;
  const float* src = sp.src[which];
  const int cq = sp.cq[which], ld = sp.ld[which], C = sp.C[which], vec = sp.vec[which];
  const int64_t nq = sp.nquad[which], stride = (int64_t)gridDim.x * 256;
  float s, inv_s;
  pow2_scale(__uint_as_float(sp.maxbits[which]), s, inv_s);
  f16x4* hi = reinterpret_cast<f16x4*>(sp.hi[which]);
  f16x4* lo = reinterpret_cast<f16x4*>(sp.lo[which]);
  for (int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t q = q0 + u * stride < nq ? q0 + u * stride : q0;
      v[u] = load_quad(src, q, cq, C, ld, vec);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t q = q0 + u * stride;
      if (q >= nq) break;
      f16x4 h4, l4;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float a = v[u][j] * s; const _Float16 ah = (_Float16)a; h4[j] = ah; l4[j] = (_Float16)(a - (float)ah); }
      hi[q] = h4; lo[q] = l4;
    }
  }
}

// dz is consumed as the B operand straight from global memory, so its planes are written in the MFMA unit layout: one 16-byte
// unit = the 8 rows of a tile row-block for one (column, channel), units ordered [n][row block][column][channel position], channel
// 4q + j at position j*Cp/4 + q (the same order the x tile uses in LDS).  Rows past Ho, columns past Wo and channels past C are zero.
__global__ __launch_bounds__(256) void split_convert_dz_units_kernel(SplitPlanes sp) {
  const float* src = sp.src[1];
  const int GZ = sp.zCp >> 2, ld = sp.ld[1], C = sp.zC, vec = sp.vec[1];
  float s, inv_s;
  pow2_scale(__uint_as_float(sp.maxbits[1]), s, inv_s);
  f16x8* hi = reinterpret_cast<f16x8*>(sp.hi[1]);
  f16x8* lo = reinterpret_cast<f16x8*>(sp.lo[1]);
  const int64_t total = (int64_t)sp.zN * sp.zTY * sp.zWp * GZ;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % GZ); int64_t r = i / GZ; const int x = (int)(r % sp.zWp); r /= sp.zWp; const int ty = (int)(r % sp.zTY); const int n = (int)(r / sp.zTY);
    f32x4 v[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int y = ty * 8 + rr;
      if (y < sp.zHo && x < sp.zWo) v[rr] = load_quad(src, ((int64_t)n * sp.zHo + y) * sp.zWo * (int64_t)GZ + (int64_t)x * GZ + q, GZ, C, ld, vec);
      else v[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int64_t ubase = (((int64_t)n * sp.zTY + ty) * sp.zWp + x) * sp.zCp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x8 h8, l8;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) { const float a = v[rr][j] * s; const _Float16 ah = (_Float16)a; h8[rr] = ah; l8[rr] = (_Float16)(a - (float)ah); }
      hi[ubase + j * GZ + q] = h8; lo[ubase + j * GZ + q] = l8;
    }
  }
}

struct WgradSplitParams {
  WgradParams b;
  const _Float16* xh; const _Float16* xl; const _Float16* zh; const _Float16* zl;   // x: dense NHWC planes; dz: unit planes (see the dz convert kernel)
  const unsigned* maxbits;
  int zTY, zWp;
};

// Software pipeline: the two co-resident workgroups of a CU run in lock-step (same work, same start), so their load and MFMA
// phases do NOT overlap by themselves.  Each workgroup therefore prefetches its NEXT tile from the planes into registers
// (XR task rounds of 8-byte loads, in flight during the MFMA loop) and only the register transpose + LDS writes sit between two
// MFMA loops.  The dz (B) fragments are not staged at all: the pre-pass wrote them in fragment order, each wave loads its pair
// (hi, lo) of the next K step from global/L2 while the MFMAs of the current step run - like the filter in the forward kernel.  Accumulators: one fp32 set per wave, folded into the workgroup's partial-sum slot in
// global memory every FOLD tiles (blocked summation without a second register set - the registers hold the prefetch).
template <int TAPS, int XR>
__global__ __launch_bounds__(256, (TAPS >= 3 && XR == 2) ? 3 : 2) void wgrad_split_kernel(WgradSplitParams sp) {
  const WgradParams& p = sp.b;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, nper = p.kh * p.gz;
  const int lsplit = q / nper, rr = q - lsplit * nper;
  const int split = lsplit * 8 + xcd;
  if (split >= p.S) return;
  const int ki = rr % p.kh, kj0 = (rr / p.kh) * p.KWG;
  const int kwg = min(p.KWG, p.kw - kj0);
  const int TCx = WTW + p.KWG - 1;
  // units per plane; fragment over-reads (< 32 units past a plane, garbage rows/columns that are discarded) land in the next
  // plane, only the last plane needs 512 B of slack - every byte counts: k = 15, 32 channels must fit twice into 160 KiB
  const int nux = TCx * p.Cin;
  f16x8* xh = reinterpret_cast<f16x8*>(lds);
  f16x8* xl = xh + nux;

  float sx_, isx, sz_, isz;
  pow2_scale(__uint_as_float(sp.maxbits[0]), sx_, isx);
  pow2_scale(__uint_as_float(sp.maxbits[1]), sz_, isz);
  f16x4 padh, padl;                                                  // the constant-padding value in the planes' units
  {
    const float a = p.pad_value * sx_; const _Float16 ah = (_Float16)a, al = (_Float16)(a - (float)ah);
    padh = (f16x4){ah, ah, ah, ah}; padl = (f16x4){al, al, al, al};
  }

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int ntile = (kwg * p.Cin + 31) >> 5;
  const int ntw = ntile > wv ? (ntile - wv + WAVES - 1) / WAVES : 0;
  const int GX = p.Cin >> 2;                                          // channel quads
  const int ntask_x = TCx * GX;

  // this thread's tasks (fixed for all tiles): x task k -> column xc[k], first channel xch[k]
  int xc[XR], xch[XR];
#pragma unroll
  for (int k = 0; k < XR; ++k) { const int task = min(tid + 256 * k, ntask_x - 1); xc[k] = task / GX; xch[k] = (task - xc[k] * GX) << 2; }

  f16x4 pxh[XR][WTH], pxl[XR][WTH];                                   // the prefetched x tile
  unsigned xin_mask[XR];                                              // bit r: row r of the task is inside the image

  // all plane addresses are (uniform base pointer) + (32-bit byte offset): hi and lo share the offset register
  const char* const bxh = reinterpret_cast<const char*>(sp.xh); const char* const bxl = reinterpret_cast<const char*>(sp.xl);
  auto issue = [&](int tile) {                                        // global loads only: nothing here waits for them
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * WTH, x0 = tx * WTW;
    const unsigned xbase = (unsigned)n * p.H * p.W;                  // pixels; planes are < 4 GiB (host check)
    // interior tiles (the x window lies inside the image: all but the border ring of tiles) need no boundary-condition index maps:
    // the generic path below costs ~2 000 scalar instructions per tile, which is what bounds the small layers
    const int sy0 = y0 + ki - p.pt, sx0 = x0 + kj0 - p.pl;
    if (sy0 >= 0 && sy0 + WTH <= p.H && sx0 >= 0 && sx0 + TCx <= p.W) {
      const unsigned row0 = xbase + (unsigned)sy0 * p.W + (unsigned)sx0, rstride = (unsigned)p.W * p.Cin * 2u;
#pragma unroll
      for (int k = 0; k < XR; ++k) {
        xin_mask[k] = 0xffu;
        unsigned e = ((row0 + (unsigned)xc[k]) * p.Cin + xch[k]) * 2u;
#pragma unroll
        for (int r = 0; r < WTH; ++r) {
          pxh[k][r] = *reinterpret_cast<const f16x4*>(bxh + e);
          pxl[k][r] = *reinterpret_cast<const f16x4*>(bxl + e);
          e += rstride;
        }
      }
      return;
    }
    unsigned xrow[WTH]; unsigned rowmask = 0;                         // wave-uniform: x rows shifted by the filter row, BC applied
#pragma unroll
    for (int r = 0; r < WTH; ++r) {
      const int sy = pcnn_pad_index(y0 + r + ki - p.pt, p.H, p.pad_mode);
      if (sy >= 0) rowmask |= 1u << r;
      xrow[r] = xbase + (unsigned)(sy >= 0 ? sy : 0) * p.W;
    }
#pragma unroll
    for (int k = 0; k < XR; ++k) {
      const int sx = pcnn_pad_index(x0 + xc[k] + kj0 - p.pl, p.W, p.pad_mode);
      xin_mask[k] = sx >= 0 ? rowmask : 0u;
#pragma unroll
      for (int r = 0; r < WTH; ++r) {
        const unsigned e = ((xrow[r] + (unsigned)(sx >= 0 ? sx : 0)) * p.Cin + xch[k]) * 2u;
        pxh[k][r] = *reinterpret_cast<const f16x4*>(bxh + e);
        pxl[k][r] = *reinterpret_cast<const f16x4*>(bxl + e);
      }
    }
  };

  auto stage = [&]() {                                                // register transpose (8 rows x 4 channels -> 4 units) + LDS writes
#pragma unroll
    for (int k = 0; k < XR; ++k) {
      if (tid + 256 * k < ntask_x) {
#pragma unroll
        for (int r = 0; r < WTH; ++r)
          if (!((xin_mask[k] >> r) & 1u)) { pxh[k][r] = padh; pxl[k][r] = padl; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f16x8 h8, l8;
#pragma unroll
          for (int r = 0; r < WTH; ++r) { h8[r] = pxh[k][r][j]; l8[r] = pxl[k][r][j]; }
          const int u = xc[k] * p.Cin + (xch[k] >> 2) + j * GX;      // channel 4q + j sits at position j*GX + q of its column
          xh[u] = h8; xl[u] = l8;
        }
      }
    }
  };

  // partial sums of this (split, filter row, column group): same layout as the fp32 kernel's, still in the planes' units (the
  // reduction kernel applies 1/(sx*sz))
  const int nflat = kwg * p.Cin;
  float* dst = p.ws + (((int64_t)split * p.kh + ki) * p.kw + kj0) * p.Cin * p.Cout;
  auto fold = [&](bool first) {
    // element offsets are rebuilt here from an opaque base: left to itself the compiler hoists all 16*TAPS 64-bit addresses out
    // of the tile loop and keeps them in registers next to the prefetched tile
    unsigned lane_off = (unsigned)((wv * 32 + 4 * half) * p.Cout + col);
    asm volatile("" : "+v"(lane_off));
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int jt = wv + WAVES * t;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int R = jt * 32 + 8 * (i >> 2) + 4 * half + (i & 3);
        if (R < nflat && col < p.Cout) {
          // later folds: fire-and-forget adds (no return value, so no registers wait on memory); every address has exactly
          // one writer thread, adding in program order - deterministic
          float* o = dst + (lane_off + (unsigned)((WAVES * t * 32 + 8 * (i >> 2) + (i & 3)) * p.Cout));
          if (first) *o = acc[t][i]; else (void)unsafeAtomicAdd(o, acc[t][i]);
        }
        acc[t][i] = 0.f;
      }
    }
  };

  // tiles are processed in groups of FOLD; the prefetch runs one tile ahead inside a group and restarts after the fold
  constexpr int FOLD = 8;
  int tile = split;
  bool first = true;
  issue(tile);
  while (tile < p.ntiles) {
    for (int g = 0; g < FOLD && tile < p.ntiles; ++g, tile += p.S) {
      __syncthreads();                                 // previous tile's fragment reads are done
      stage();
      __syncthreads();
      if (g + 1 < FOLD && tile + p.S < p.ntiles) issue(tile + p.S);   // in flight during the MFMA loop
      // ---- accumulate: 16 K steps of 2 columns x 8 rows; B (dz) fragments from the unit planes, one step ahead
      {
        int tt = tile;
        const int tx = tt % p.tiles_x; tt /= p.tiles_x;
        const int ty = tt % p.tiles_y;
        const int n = tt / p.tiles_y;
        const int64_t zu = (((int64_t)n * sp.zTY + ty) * sp.zWp + tx * WTW + half) * p.Cout + col;     // unit of step 0
        const f16x8* zh_t = reinterpret_cast<const f16x8*>(sp.zh) + zu;
        const f16x8* zl_t = reinterpret_cast<const f16x8*>(sp.zl) + zu;
        const int zstep = 2 * p.Cout;                                 // units per K step (2 columns)
        f16x8 bh = zh_t[0], bl = zl_t[0];
#pragma unroll 2
        for (int xp = 0; xp < WTW / 2; ++xp) {
          const int cx = 2 * xp + half;
          const f16x8 nbh = zh_t[(xp + 1) * zstep], nbl = zl_t[(xp + 1) * zstep];   // past the last step: the next tile's units (unused)
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
          bh = nbh; bl = nbl;
        }
      }
    }
    if (first) fold(true); else fold(false);
    first = false;
    if (tile < p.ntiles) issue(tile);
  }
}

template <int TAPS, int XR>
void launch_wgrad_split(pcnn_handle h, const WgradSplitParams& p, const WgradPlan& pl, size_t lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_kernel<TAPS, XR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((wgrad_split_kernel<TAPS, XR>), dim3(8 * pcnn_cdiv(pl.S, 8) * p.b.kh * pl.gz), dim3(256), lds, h->stream, p);
}

// workspace = [split partials | 256 B: the two tensor maxima | x hi | x lo | dz hi | dz lo]; the planes exist for shapes the
// split kernel takes: it sees the layer with both channel counts padded to multiples of 4 (the planes are private copies)
size_t partials_bytes(const pcnn_conv_desc* d, const WgradPlan& pl) {
  return (((size_t)pl.S * d->kh * d->kw * d->Cin * d->Cout * sizeof(float)) + 255) & ~(size_t)255;
}
pcnn_conv_desc padded_desc(const pcnn_conv_desc* d) {
  pcnn_conv_desc pd = *d;
  pd.Cin = (d->Cin + 3) & ~3; pd.Cout = (d->Cout + 3) & ~3;
  return pd;
}
size_t plane_elems_x(const pcnn_conv_desc* pd) { return (((size_t)pd->N * pd->H * pd->W * pd->Cin) + 127) & ~(size_t)127; }
int z_ty(const pcnn_conv_desc* pd) { return pcnn_cdiv(pd->Ho, WTH); }
int z_wp(const pcnn_conv_desc* pd) { return pcnn_cdiv(pd->Wo, WTW) * WTW; }
// dz unit planes: 8 * [N][TY][Wp][Cout] halfs + one K step of slack for the fragment prefetch past the last tile
size_t plane_elems_z(const pcnn_conv_desc* pd) { return (size_t)pd->N * z_ty(pd) * z_wp(pd) * pd->Cout * 8 + (size_t)2 * 64 * pd->Cout * 8; }
size_t plane_bytes(const pcnn_conv_desc* pd) { return 2 * sizeof(_Float16) * (plane_elems_x(pd) + plane_elems_z(pd)); }
int split_xr(const pcnn_conv_desc* pd, const WgradPlan& pls) { return pcnn_cdiv((WTW + pls.KWG - 1) * (pd->Cin >> 2), 256); }
size_t split_lds(const pcnn_conv_desc* pd, const WgradPlan& pls) { return (size_t)(WTW + pls.KWG - 1) * pd->Cin * 32 + 512; }
// Grid of the split kernel: sized like the fp32 kernel's (2 workgroups per CU).  Sizing it to the split kernel's own, higher
// occupancy limit (up to 5 per CU for the small variants) was measured: no gain - the small layers are bound by the kh-fold
// re-read of the planes, not by latency.
WgradPlan make_split_plan(const pcnn_conv_desc* pd) {
  WgradPlan pl = make_plan(pd);
  static const int occ3 = getenv("PCNN_WG_OCC3") ? atoi(getenv("PCNN_WG_OCC3")) : 1;
  if (occ3 && pl.TAPS >= 3 && split_xr(pd, pl) == 2 && split_lds(pd, pl) * 3 <= 160 * 1024) {
    // these variants are compiled for 3 workgroups per CU (168 VGPRs): size the grid for 768 slots
    int S = 8 * ((768 / 8 * 4) / (pd->kh * pl.gz));
    if (S < 8) S = (768 * 4) / (pd->kh * pl.gz);
    if (S > pl.ntiles) S = pl.ntiles;
    if (S < 1) S = 1;
    pl.S = S;
  }
  return pl;
}
bool split_eligible(const pcnn_conv_desc* pd, const WgradPlan& pls) {      // pd, pls: the padded layer and its plan
  const int xr = split_xr(pd, pls);
  return pls.NTC == 1 && xr >= 1 && xr <= 3 && pls.TAPS <= 4 && split_lds(pd, pls) <= 160 * 1024 &&
         plane_elems_x(pd) * 2 < ((size_t)1 << 32) && plane_elems_z(pd) * 2 < ((size_t)1 << 32);
}

}  // namespace

extern "C" size_t pcnn_conv2d_wgrad_workspace(const pcnn_conv_desc* d) {
  if (!d) return 0;
  const WgradPlan pl = make_plan(d);
  const pcnn_conv_desc pd = padded_desc(d);
  const WgradPlan pls = make_split_plan(&pd);
  const size_t plain = partials_bytes(d, pl);
  const size_t split = split_eligible(&pd, pls) ? partials_bytes(&pd, pls) + 256 + plane_bytes(&pd) : 0;
  return std::max(std::max(plain, split), pcnn_conv_small_wgrad_workspace(d));
}

extern "C" int pcnn_conv2d_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw,
                                 void* workspace, size_t workspace_bytes) {
  return pcnn_conv2d_wgrad_hint(h, d, x, dz, dw, workspace, workspace_bytes, nullptr, nullptr);
}

extern "C" int pcnn_conv2d_wgrad_hint(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw,
                                      void* workspace, size_t workspace_bytes, const float* x_absmax, const float* dz_absmax) {
  PCNN_REQUIRE(h, h && d && x && dz && dw && workspace, "pcnn_conv2d_wgrad: null argument");
  PCNN_REQUIRE(h, d->Cin >= 1 && d->Cin <= 128 && d->Cout >= 1 && d->Cout <= 64, "pcnn_conv2d_wgrad: channels %d->%d unsupported (<=64)", d->Cin, d->Cout);
  PCNN_REQUIRE(h, d->ldx >= d->Cin && d->ldy >= d->Cout, "pcnn_conv2d_wgrad: channel stride smaller than channel count");
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_conv2d_wgrad_workspace(d), "pcnn_conv2d_wgrad: workspace too small");
  if (pcnn_conv_small_wgrad_eligible(d)) return pcnn_conv_small_wgrad(h, d, x, dz, dw, workspace, workspace_bytes);
  if (pcnn_spectral_eligible(h, d, true)) return pcnn_spectral_conv_wgrad(h, d, x, dz, dw);
  const pcnn_conv_desc pd = padded_desc(d);
  const WgradPlan pls = make_split_plan(&pd);
  const bool split_ok = h->math_mode == PCNN_MATH_SPLIT_F16 && split_eligible(&pd, pls);
  const pcnn_conv_desc* ud = split_ok ? &pd : d;                       // the layer as the chosen kernel sees it
  const WgradPlan pl = split_ok ? pls : make_plan(d);
  PCNN_REQUIRE(h, split_ok || pl.lds <= 160 * 1024, "pcnn_conv2d_wgrad: tile needs %zu B of LDS", pl.lds);
  WgradParams p;
  p.x = x; p.dz = dz; p.ws = static_cast<float*>(workspace);
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = ud->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = ud->Cout; p.lddz = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.S = pl.S; p.KWG = pl.KWG; p.lg = pl.lg; p.gz = pl.gz;
  p.vecx = (d->Cin % 4 == 0) && (d->ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.vecdz = (d->Cout % 4 == 0) && (d->ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15) == 0);
  const int xr = split_xr(&pd, pls);
  if (split_ok) {
    char* base = static_cast<char*>(workspace) + partials_bytes(&pd, pl);
    SplitPlanes pp;
    pp.maxbits = reinterpret_cast<unsigned*>(base);
    _Float16* planes = reinterpret_cast<_Float16*>(base + 256);
    pp.src[0] = x; pp.src[1] = dz; pp.hint[0] = x_absmax; pp.hint[1] = dz_absmax;
    pp.hi[0] = planes; pp.lo[0] = planes + plane_elems_x(&pd);
    pp.hi[1] = pp.lo[0] + plane_elems_x(&pd); pp.lo[1] = pp.hi[1] + plane_elems_z(&pd);
    pp.cq[0] = pd.Cin >> 2; pp.cq[1] = pd.Cout >> 2; pp.C[0] = d->Cin; pp.C[1] = d->Cout; pp.ld[0] = d->ldx; pp.ld[1] = d->ldy;
    pp.vec[0] = p.vecx ? (d->ldx == d->Cin ? 2 : 1) : 0; pp.vec[1] = p.vecdz ? (d->ldy == d->Cout ? 2 : 1) : 0;   // 2 = dense
    pp.zN = d->N; pp.zHo = d->Ho; pp.zWo = d->Wo; pp.zTY = z_ty(&pd); pp.zWp = z_wp(&pd); pp.zC = d->Cout; pp.zCp = pd.Cout;
    pp.nquad[0] = (int64_t)d->N * d->H * d->W * pp.cq[0]; pp.nquad[1] = (int64_t)d->N * d->Ho * d->Wo * pp.cq[1];
    (void)hipMemsetAsync(pp.maxbits, 0, 8, h->stream);
    const int64_t nq = std::max(pp.nquad[0], pp.nquad[1]);
    const unsigned gx = (unsigned)std::min<int64_t>(pcnn_cdiv64(nq, 256 * 4), 4096);
    hipLaunchKernelGGL(split_absmax_kernel, dim3(gx, 2), dim3(256), 0, h->stream, pp);
    hipLaunchKernelGGL(split_convert_kernel, dim3(gx, 1), dim3(256), 0, h->stream, pp);          // x -> dense NHWC planes
    {
      const int64_t zt = (int64_t)pp.zN * pp.zTY * pp.zWp * (pp.zCp >> 2);
      hipLaunchKernelGGL(split_convert_dz_units_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(zt, 256), 8192)), dim3(256), 0, h->stream, pp);
    }
    PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(split planes)");
    WgradSplitParams sp;
    sp.b = p; sp.xh = pp.hi[0]; sp.xl = pp.lo[0]; sp.zh = pp.hi[1]; sp.zl = pp.lo[1]; sp.maxbits = pp.maxbits; sp.zTY = pp.zTY; sp.zWp = pp.zWp;
    const size_t lds = split_lds(&pd, pl);
#define PCNN_WGS(T, R) if (pl.TAPS == T && xr == R) launch_wgrad_split<T, R>(h, sp, pl, lds); else
    PCNN_WGS(1, 1) PCNN_WGS(1, 2) PCNN_WGS(1, 3) PCNN_WGS(2, 1) PCNN_WGS(2, 2) PCNN_WGS(2, 3)
    PCNN_WGS(3, 1) PCNN_WGS(3, 2) PCNN_WGS(3, 3) PCNN_WGS(4, 1) PCNN_WGS(4, 2) PCNN_WGS(4, 3)
    { PCNN_FAIL(h, "pcnn_conv2d_wgrad: no split kernel for TAPS=%d XR=%d", pl.TAPS, xr); }
#undef PCNN_WGS
  } else
#define PCNN_WG(Q, T) if (pl.NTC == Q && pl.TAPS == T) { launch_wgrad<Q, T>(h, p, pl); } else
  PCNN_WG(1, 1) PCNN_WG(1, 2) PCNN_WG(1, 3) PCNN_WG(1, 4) PCNN_WG(2, 1) PCNN_WG(2, 2)
  { PCNN_FAIL(h, "pcnn_conv2d_wgrad: no kernel for NTC=%d TAPS=%d", pl.NTC, pl.TAPS); }
#undef PCNN_WG
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad");
  const int64_t nel = (int64_t)d->kh * d->kw * ud->Cin * ud->Cout;
  const int SL = nel >= 65536 ? 1 : (nel >= 8192 ? 4 : 16);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(nel, 256 / SL), 2048)), dim3(256), 0, h->stream,
                     p.ws, dw, nel, pl.S, split_ok ? reinterpret_cast<const unsigned*>(static_cast<char*>(workspace) + partials_bytes(&pd, pl)) : nullptr,
                     d->Cin, d->Cout, ud->Cin, ud->Cout, SL);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(reduce)");
  return 0;
}
