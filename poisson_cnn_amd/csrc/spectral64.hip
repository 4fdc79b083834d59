// 64-point tiles of the tiled spectral convolution (spectral_conv.hip holds the 32-point tiles, the per-frequency mixing kernels and the host
// side).  For a filter of k taps an overlap-save tile of T points yields T - k + 1 valid outputs per axis, so the spectrum volume per output
// pixel is (T / (T - k + 1))^2: 3.2 at k = 15, T = 32 but 1.6 at T = 64 - and the spectrum is what the four passes of a convolution
// (transform, mixing read + write, inverse) stream through HBM.  A 64 x 64 x 32-channel intermediate (512 KB) does not fit the 160 KB of
// LDS, so these kernels never hold it: they STREAM one axis and keep the other axis' result in the matrix-core accumulators.
//
//   spec64_fwd_kernel   item = (tile, 32 channels, x parity px).  One radix-2 step per axis on the vector ALU (u = w[x] +- w[x + 32]; the even /
//                       odd x frequencies are 32-point transforms of u), so an item produces the 16 column pairs (Re, Im) of its parity.
//                       The window is walked in 8 stages of 8 rows (rows y0..y0+3 and their partners y0+32..): each wave x-transforms one row
//                       (16 MFMAs 32x32x2, lane = channel) into a 32 KB LDS stage buffer, then every wave accumulates those rows into the
//                       y-axis transform of ITS columns: 4 column pairs x {Re, Im} x 32 frequencies of its parity = 128 accumulator registers
//                       that stay resident over the 8 stages and are stored once as 128-byte spectrum rows.  One LDS barrier per stage.
//   spec64_inv_kernel   item = (tile, 16 channels), v_mfma_f32_16x16x4_f32 (lane = channel, 64-byte half rows): the mirror image.  Stage =
//                       8 column pairs (4 of even, 4 of odd x frequency): each wave inverse-transforms one pair along y (decimation in time)
//                       straight from global into a stage buffer D[y][column], then every wave accumulates the columns into the x-axis
//                       inverse of ITS output rows (E / O halves of the radix-2 step, 112 accumulator registers), followed by the fused conv
//                       epilogue from the accumulators (bias, activation, BN affine, residual, act_out).
// Index conventions and every table below: tools/spectral_model64.py (numpy model, checked against numpy's FFT).
#include "spectral_common.h"
#include <math.h>
#include <algorithm>

namespace pcnn_spec {

namespace {

constexpr int T = 64, ROWS = 4096;
// table block (floats): GX [px][ks 16][lane]; Y [h][ks 16][C S R][lane]; IY [hh][C S][yb 2][ks 8][lane]; IR [hh][yb][ks 8][lane];
// IX [ex][tau 4][kk 2][xb 2][lane]
constexpr int TB_GX = 0, TB_Y = 2048, TB_IY = 8192, TB_IR = 12288, TB_IX = 14336, TB_END = 16384;
static_assert(TB_END <= TAB64_FLOATS, "table block");
constexpr int SB_FLOATS = 8 * 32 * 32;             // one stage buffer of the forward transform: 8 rows x 32 columns x 32 channels

// (frequency, is-imaginary) of row rho of the real-column transform of parity h, and its spectrum row inside the column's 64 rows
__host__ __device__ __forceinline__ int ry_row(int h, int rho) {
  return h == 0 ? (rho <= 16 ? 2 * rho : 32 + 2 * (rho - 16)) : (rho < 16 ? 2 * rho + 1 : 32 + 2 * (rho - 16) + 1);
}

// ------------------------------------------------------------------------------------------------------------------ forward transform
struct Ctx64 {
  const float* img;      // uniform: image n
  int wy0, wx0;          // uniform: window origin
  int ylim, xlim;        // uniform: rows / columns of the window that can be non-zero
  int tg;                // uniform: spectrum item
  bool fast;             // uniform: the window lies inside the image in x (no index maps)
  unsigned off0;         // lane: float offset of window column `half` + this lane's channel (fast path), or the channel alone
  bool cok;              // lane: this lane's channel exists
};

__device__ __forceinline__ void make_ctx(const FwdParams& p, int tg, int half, int c, Ctx64& cx) {
  const int g = tg % p.groups;
  int t = p.tile0 + tg / p.groups;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int chan = g * p.cstride + c;
  cx.cok = c < p.cvalid && chan < p.C;
  cx.img = p.x + (int64_t)n * p.H * p.W * p.ld;
  cx.wy0 = ty * p.Vy - p.oy;
  cx.wx0 = tx * p.Vx - p.ox;
  cx.ylim = min(p.ylim, p.ext_y - ty * p.Vy);
  cx.xlim = min(p.xlim, p.ext_x - tx * p.Vx);
  cx.tg = tg;
  cx.fast = cx.wx0 >= 0 && cx.wx0 + T <= p.W;
  cx.off0 = (unsigned)((cx.fast ? (cx.wx0 + half) * p.ld : 0) + (cx.cok ? chan : 0));
}

// issues the 32 loads of window row y: lo[ks] = column 2 ks + half, hi[ks] = column 32 + 2 ks + half (always-valid addresses; padding and
// masks are applied when the values are consumed).  y is wave-uniform.
template <bool MASKED>
__device__ __forceinline__ void load_row64(const FwdParams& p, const Ctx64& cx, int y, int half, float (&lo)[16], float (&hi)[16]) {
  if (MASKED && y >= cx.ylim) return;
  const int sy = pcnn_pad_index(cx.wy0 + y, p.H, p.pad_mode);
  const float* row = cx.img + (int64_t)(sy < 0 ? 0 : sy) * p.W * p.ld;
  if (cx.fast) {
    const float* rl = row, *rh = row + 32 * p.ld;                   // uniform pointers step by two pixels; ONE lane offset for all 32 loads
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      lo[ks] = rl[cx.off0];
      hi[ks] = rh[cx.off0];
      rl += 2 * p.ld; rh += 2 * p.ld;
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int sa = pcnn_pad_index(cx.wx0 + 2 * ks + half, p.W, p.pad_mode), sb = pcnn_pad_index(cx.wx0 + 32 + 2 * ks + half, p.W, p.pad_mode);
      lo[ks] = row[cx.off0 + (unsigned)((sa < 0 ? 0 : sa) * p.ld)];
      hi[ks] = row[cx.off0 + (unsigned)((sb < 0 ? 0 : sb) * p.ld)];
    }
  }
}

template <bool MASKED>
__global__ __launch_bounds__(512, 2) void spec64_fwd_kernel(FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const SB = lds;                                   // [2][8 rows][32 columns][32 channels]
  float* const TY = lds + 2 * SB_FLOATS;                   // [h][ks][C S R][lane]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave & 1, pg = wave >> 1;
  for (int i = tid; i < 6144; i += 512) TY[i] = p.tab[TB_Y + i];
  const int ntg = p.ntile * p.groups;
  const int nitem = 2 * ((ntg + 7) & ~7);                  // item -> (tile group, px): px = (item >> 3) & 1, group = ((item >> 4) << 3) | (item & 7):
                                                           // with a grid that is a multiple of 16 a workgroup keeps ONE parity (its table stays in
                                                           // registers) and the two parities of a tile run on workgroups b and b + 8 (one XCD's L2)
  auto group_of = [&](int it) { return ((it >> 4) << 3) | (it & 7); };
  int item = blockIdx.x;
  while (item < nitem && group_of(item) >= ntg) item += gridDim.x;
  if (item >= nitem) return;
  const int px = (item >> 3) & 1;
  const bool special = px == 0 && pg == 0;                 // this wave's first pair is the two real columns (fx = 0 | 32)
  const float sgx = px ? -1.f : 1.f, sgy = h ? -1.f : 1.f;
  float gx[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) gx[ks] = p.tab[TB_GX + (px * 16 + ks) * 64 + lane];
  const float* const ty = TY + h * (16 * 3 * 64) + lane;
  const int srow = (wave & 3) + 32 * (wave >> 2);          // this wave's window row in stage 0 (stage s: + 4 s)
  // LDS addresses of this lane: its x-axis results (row = wave) and the columns of its four pairs in stage rows `half` (+ 2 kap, + 4 for the partner)
  const int wr_off = (wave * 32 + 4 * half) * 32 + c;
  const int rd_off = (half * 32 + 4 * pg) * 32 + c;
  Ctx64 cur;
  make_ctx(p, group_of(item), half, c, cur);
  float lo[16], hi[16];
  load_row64<MASKED>(p, cur, srow, half, lo, hi);
  f32x16 Z[4][2];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) { Z[jj][0] = zero16(); Z[jj][1] = zero16(); }
  int st = 0;
  int next = item + gridDim.x;
  while (next < nitem && group_of(next) >= ntg) next += gridDim.x;
  // ONE loop over (item, stage): stage 7 of an item prefetches the first row of the next one and is followed by the item's stores
#pragma unroll 1
  for (;;) {
    float* const sb = SB + (st & 1) * SB_FLOATS;
    // ---- x axis of this wave's row: D[rho][c] = sum_x GX[rho][x] (w[x] +- w[x + 32]); A = GX (lane = rho), B = the pixel's channel row
    {
      const int y = srow + 4 * st;
      f32x16 acc = zero16();
      if (!(MASKED && y >= cur.ylim)) {
        const int sy = pcnn_pad_index(cur.wy0 + y, p.H, p.pad_mode);
        const bool edge = sy < 0 || (!cur.fast && p.pad_mode == PCNN_PAD_CONSTANT);       // uniform: some pixels of this row are constant padding
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          float a = lo[ks], b = hi[ks];
          const int xa = 2 * ks + half, xb = xa + 32;
          if (edge) {
            if (sy < 0 || (unsigned)(cur.wx0 + xa) >= (unsigned)p.W) a = p.pad_value;
            if (sy < 0 || (unsigned)(cur.wx0 + xb) >= (unsigned)p.W) b = p.pad_value;
          }
          if (MASKED) { if (xa >= cur.xlim) a = 0.f; if (xb >= cur.xlim) b = 0.f; }
          float u = a + sgx * b;
          if (!cur.cok) u = 0.f;
          acc = mfma(gx[ks], u, acc);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sb[wr_off + (8 * (r >> 2) + (r & 3)) * 32] = acc[r];
    }
    lds_barrier();
    // ---- the row of the next stage (or of the next item's first stage): its latency sits under this stage's y phase
    if (st < 7) load_row64<MASKED>(p, cur, srow + 4 * (st + 1), half, lo, hi);
    else if (next < nitem) {
      Ctx64 nx;
      make_ctx(p, group_of(next), half, c, nx);
      load_row64<MASKED>(p, nx, srow, half, lo, hi);
    }
    // ---- y axis, rows 4 st .. 4 st + 3 and + 32: v = D[y] +- D[y + 32]; K step = two consecutive y (lane half).  The real pair of the
    // special wave takes the same four MFMAs with (R, 0) in place of (C, S)
#pragma unroll
    for (int kap = 0; kap < 2; ++kap) {
      const float* t = ty + (2 * st + kap) * (3 * 64);
      const float cy = t[0], sny = t[64], ry = t[128];
      const float c0 = special ? ry : cy, s0 = special ? 0.f : sny;
      const float* slo = sb + rd_off + (2 * kap) * 1024;
      const float* shi = slo + 4 * 1024;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float vr = slo[jj * 32] + sgy * shi[jj * 32];
        const float vi = slo[(16 + jj) * 32] + sgy * shi[(16 + jj) * 32];
        const float ac = jj == 0 ? c0 : cy, as = jj == 0 ? s0 : sny;
        Z[jj][0] = mfma(ac, vr, Z[jj][0]);
        Z[jj][1] = mfma(ac, vi, Z[jj][1]);
        Z[jj][0] = mfma(as, vi, Z[jj][0]);
        Z[jj][1] = mfma(as, -vr, Z[jj][1]);
      }
    }
    if (++st < 8) continue;
    // ---- the item's spectrum rows, straight from the accumulators (accumulator row m <-> fy = 2 m + h)
    float* const out = p.sp + sp_item(cur.tg, ROWS);
    // the 128 store offsets are invariant across items: left alone, the compiler hoists all of them out of the persistent loop (256 registers
    // of 64-bit offsets, spilled); opaque lane coordinates make it form them here, one v_add each
    int hv = half, cv = c;
    asm volatile("" : "+v"(hv), "+v"(cv));
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int fx = 2 * (4 * pg + jj) + px;
      const int base = 128 + 128 * (fx - 1) + h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 8 * (r >> 2) + 4 * hv + (r & 3);
        int r0 = base + 2 * m, r1 = r0 + 64;
        if (jj == 0 && special) { r0 = ry_row(h, m); r1 = 64 + r0; }      // the two real columns: half-complex rows of fx = 0 and fx = 32
        out[(unsigned)(r0 * RS + cv)] = Z[jj][0][r];
        out[(unsigned)(r1 * RS + cv)] = Z[jj][1][r];
      }
      Z[jj][0] = zero16(); Z[jj][1] = zero16();
    }
    if (next >= nitem) break;
    item = next;
    make_ctx(p, group_of(item), half, c, cur);
    next = item + gridDim.x;
    while (next < nitem && group_of(next) >= ntg) next += gridDim.x;
    st = 0;
  }
}

// ------------------------------------------------------------------------------------------------------------------ inverse transform + epilogue
constexpr int NR = 7;                  // output rows per wave: rows y = wave + 8 i < 56 (tiles of >= 9 taps)

struct InvItem { int tg, q, vy; };

template <bool TANH, bool RES>
__global__ __launch_bounds__(512, 2) void spec64_inv_kernel(InvParams p, int vycap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const TIY = lds;                                  // [hh][C S][yb][ks][lane]
  float* const TIR = lds + 4096;                           // [hh][yb][ks][lane]
  float* const TIX = lds + 6144;                           // [ex][tau][kk][xb][lane]
  float* const DB = lds + 8192;                            // [2][vycap rows][16 columns][16 channels]
  const int tid = threadIdx.x, lane = tid & 63, g4 = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ex = wave >> 2, s = wave & 3;
  for (int i = tid; i < 8192; i += 512) lds[i] = p.tab[TB_IY + i];
  const int ntg = p.ntile * p.groups;
  const int nitem = 2 * ((ntg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);     // this workgroup's items: (tile group, channel half)
  if (nitem <= 0) return;
  auto item_of = [&](int n, InvItem& it) {
    it.tg = blockIdx.x + (n >> 1) * gridDim.x; it.q = n & 1;
    int t = p.tile0 + it.tg / p.groups;
    t /= p.tiles_x;
    it.vy = min(p.Vy, p.Ho - (t % p.tiles_y) * p.Vy);
  };
  // spectrum rows of pair j = 4 tau + s of this wave's x parity, 16 channels: z[16 hh + 8 part + ks] = row (fy = 2 (4 ks + g4) + hh)
  auto load_pair = [&](const InvItem& it, int tau, float (&z)[32]) {
    const float* in = p.sp + sp_item(it.tg, ROWS) + 16 * it.q + c;
    const int j = 4 * tau + s;
    int g4 = lane >> 4;
    asm volatile("" : "+v"(g4));                               // opaque: the 32 row offsets are formed per call, not hoisted out of the item loop
    if (ex == 0 && j == 0) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const int row = ry_row(hh, 4 * ks + g4);
          z[16 * hh + ks] = in[(unsigned)(row * RS)];
          z[16 * hh + 8 + ks] = in[(unsigned)((64 + row) * RS)];
        }
    } else {
      const float* src = in + (128 + 128 * (2 * j + ex - 1)) * RS;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const int row = 2 * (4 * ks + g4) + hh;
          z[16 * hh + ks] = src[(unsigned)(row * RS)];
          z[16 * hh + 8 + ks] = src[(unsigned)((64 + row) * RS)];
        }
    }
  };
  f32x4 XE[NR][2], XO[NR][2];
#pragma unroll
  for (int i = 0; i < NR; ++i) { XE[i][0] = zero4(); XE[i][1] = zero4(); XO[i][0] = zero4(); XO[i][1] = zero4(); }
  float ymax = 0.f;
  float z[32];
  InvItem ia, ib;                                             // ia: the item whose columns are being transformed, ib: the item being accumulated
  item_of(0, ia);
  ib = ia;
  load_pair(ia, 0, z);
  const int nphase = 4 * nitem;
  // phase k: A(k) = y-axis inverse of this wave's column pair of (item k / 4, stage k % 4) into DB[k & 1]; B(k - 1) = x-axis accumulation of the
  // previous stage's 16 columns from DB[(k - 1) & 1]; one barrier per phase
#pragma unroll 1
  for (int k = 0; k <= nphase; ++k) {
    if (k < nphase) {
      const int tau = k & 3;
      const bool special = ex == 0 && tau == 0 && s == 0;
      f32x4 Er[2] = {zero4(), zero4()}, Ei[2] = {zero4(), zero4()}, Or[2] = {zero4(), zero4()}, Oi[2] = {zero4(), zero4()};
      if (special) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const float a00 = TIR[((0 * 2 + 0) * 8 + ks) * 64 + lane], a01 = TIR[((0 * 2 + 1) * 8 + ks) * 64 + lane];
          const float a10 = TIR[((1 * 2 + 0) * 8 + ks) * 64 + lane], a11 = TIR[((1 * 2 + 1) * 8 + ks) * 64 + lane];
          Er[0] = mfma16(a00, z[ks], Er[0]); Ei[0] = mfma16(a00, z[8 + ks], Ei[0]);
          Er[1] = mfma16(a01, z[ks], Er[1]); Ei[1] = mfma16(a01, z[8 + ks], Ei[1]);
          Or[0] = mfma16(a10, z[16 + ks], Or[0]); Oi[0] = mfma16(a10, z[24 + ks], Oi[0]);
          Or[1] = mfma16(a11, z[16 + ks], Or[1]); Oi[1] = mfma16(a11, z[24 + ks], Oi[1]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          // [hh][C S][yb][ks]
          const float c00 = TIY[(((0 * 2 + 0) * 2 + 0) * 8 + ks) * 64 + lane], c01 = TIY[(((0 * 2 + 0) * 2 + 1) * 8 + ks) * 64 + lane];
          const float s00 = TIY[(((0 * 2 + 1) * 2 + 0) * 8 + ks) * 64 + lane], s01 = TIY[(((0 * 2 + 1) * 2 + 1) * 8 + ks) * 64 + lane];
          const float zr = z[ks], zi = z[8 + ks], nzi = -zi;
          Er[0] = mfma16(c00, zr, Er[0]); Ei[0] = mfma16(s00, zr, Ei[0]);
          Er[1] = mfma16(c01, zr, Er[1]); Ei[1] = mfma16(s01, zr, Ei[1]);
          Er[0] = mfma16(s00, nzi, Er[0]); Ei[0] = mfma16(c00, zi, Ei[0]);
          Er[1] = mfma16(s01, nzi, Er[1]); Ei[1] = mfma16(c01, zi, Ei[1]);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const float c10 = TIY[(((1 * 2 + 0) * 2 + 0) * 8 + ks) * 64 + lane], c11 = TIY[(((1 * 2 + 0) * 2 + 1) * 8 + ks) * 64 + lane];
          const float s10 = TIY[(((1 * 2 + 1) * 2 + 0) * 8 + ks) * 64 + lane], s11 = TIY[(((1 * 2 + 1) * 2 + 1) * 8 + ks) * 64 + lane];
          const float zr = z[16 + ks], zi = z[24 + ks], nzi = -zi;
          Or[0] = mfma16(c10, zr, Or[0]); Oi[0] = mfma16(s10, zr, Oi[0]);
          Or[1] = mfma16(c11, zr, Or[1]); Oi[1] = mfma16(s11, zr, Oi[1]);
          Or[0] = mfma16(s10, nzi, Or[0]); Oi[0] = mfma16(c10, zi, Oi[0]);
          Or[1] = mfma16(s11, nzi, Or[1]); Oi[1] = mfma16(c11, zi, Oi[1]);
        }
      }
      const int vy = ia.vy;
      // the next pair's rows: their latency sits under the rest of this phase and the next one's x-axis part
      if (k + 1 < nphase) {
        if (tau == 3) item_of((k + 1) >> 2, ia);
        load_pair(ia, (k + 1) & 3, z);
      }
      // D[y] = E + O (y < 32), E - O (y >= 32) -> stage buffer [y][column 2 wave + part][c]
      float* db = DB + (k & 1) * (vycap * 256) + (2 * wave) * 16 + c;
#pragma unroll
      for (int yb = 0; yb < 2; ++yb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int y = 16 * yb + 4 * g4 + r;
          if (y < vy) { db[y * 256] = Er[yb][r] + Or[yb][r]; db[y * 256 + 16] = Ei[yb][r] + Oi[yb][r]; }
          if (y + 32 < vy) { db[(y + 32) * 256] = Er[yb][r] - Or[yb][r]; db[(y + 32) * 256 + 16] = Ei[yb][r] - Oi[yb][r]; }
        }
    }
    if (k > 0) {
      // ---- B(k - 1): this wave's output rows y = wave + 8 i; columns 0..7 feed the even-frequency half E_x, 8..15 the odd half O_x
      const int tau = (k - 1) & 3;
      const float* db = DB + ((k - 1) & 1) * (vycap * 256) + g4 * 16 + c;
      float ax[2][2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int xb = 0; xb < 2; ++xb) ax[e][kk][xb] = TIX[(((e * 4 + tau) * 2 + kk) * 2 + xb) * 64 + lane];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int y = wave + 8 * i;
        if (y < ib.vy) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const float be = db[y * 256 + (4 * kk) * 16], bo = db[y * 256 + (8 + 4 * kk) * 16];
            XE[i][0] = mfma16(ax[0][kk][0], be, XE[i][0]);
            XE[i][1] = mfma16(ax[0][kk][1], be, XE[i][1]);
            XO[i][0] = mfma16(ax[1][kk][0], bo, XO[i][0]);
            XO[i][1] = mfma16(ax[1][kk][1], bo, XO[i][1]);
          }
        }
      }
      if (tau == 3) {
        // ---- epilogue of item ib from the accumulators: out[x] = E + O (x < 32), E - O (x >= 32); lane = (pixel group g4, channel)
        const int g = ib.tg % p.groups;
        int t = p.tile0 + ib.tg / p.groups;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int tyy = t % p.tiles_y;
        const int n = t / p.tiles_y;
        const int y0 = tyy * p.Vy, x0 = tx * p.Vx;
        const int vx = min(p.Vx, p.Wo - x0);
        const int cc = 16 * ib.q + c, chan = g * p.cstride + cc;
        const bool cok = cc < p.cvalid && chan < p.C;
        const float bias = (p.bias && cok) ? p.bias[chan] : 0.f;
        const float sc = (p.bn_scale && cok) ? p.bn_scale[chan] : 1.f, sh = (p.bn_scale && cok) ? p.bn_shift[chan] : 0.f;
        const int xsgn = p.flip ? -1 : 1;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int y = wave + 8 * i;
          if (y < ib.vy) {
            const int64_t rowpix = p.flip ? ((int64_t)n * p.Ho + (p.Ho - 1 - y0 - y)) * p.Wo + (p.Wo - 1 - x0) : ((int64_t)n * p.Ho + y0 + y) * p.Wo + x0;
            float* yrow = p.y + rowpix * p.ldy;
            float* arow = p.act_out ? p.act_out + rowpix * p.ld_act : nullptr;
            const float* rrow = RES ? p.res + rowpix * p.ld_res : nullptr;
            unsigned chv = (unsigned)chan;
            asm volatile("" : "+v"(chv));
            if (cok) {
#pragma unroll
              for (int hx = 0; hx < 2; ++hx) {                       // x < 32 | x >= 32
                float rv[8];
                if (RES) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) {
                    const int xx = 32 * hx + 16 * (e >> 2) + 4 * g4 + (e & 3);
                    rv[e] = rrow[xx < vx ? (int)(xsgn * xx * p.ld_res) + (int)chv : (int)chv];
                  }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const int xb = e >> 2, r = e & 3;
                  const int xx = 32 * hx + 16 * xb + 4 * g4 + r;
                  if (xx < vx) {
                    float v = (hx ? XE[i][xb][r] - XO[i][xb][r] : XE[i][xb][r] + XO[i][xb][r]) + bias;
                    v = TANH ? tanhf(v) : (v > 0.f ? v : v * p.alpha);
                    const int xo = xsgn * xx;
                    if (arow) arow[(int)(xo * p.ld_act) + (int)chv] = v;
                    v = v * sc + sh;
                    if (RES) v += rv[e];
                    yrow[(int)(xo * p.ldy) + (int)chv] = v;
                    ymax = fmaxf(ymax, fabsf(v));
                  }
                }
              }
            }
          }
          XE[i][0] = zero4(); XE[i][1] = zero4(); XO[i][0] = zero4(); XO[i][1] = zero4();
        }
        if (k < nphase) item_of(k >> 2, ib);
      }
    }
    lds_barrier();
  }
  if (p.absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if (lane == 0) {
      const unsigned bits = __float_as_uint(ymax <= 3.0e38f ? ymax : 3.0e38f);
      if (bits > __atomic_load_n(p.absmax, __ATOMIC_RELAXED)) atomicMax(p.absmax, bits);
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------ host side
void build_tables64(float* tab, int* slots) {
  const double tp = 2.0 * M_PI / T;
  for (int i = 0; i < TAB64_FLOATS; ++i) tab[i] = 0.f;
  auto ry_val = [&](int h, int rho, int y, double w) {
    const int row = ry_row(h, rho);
    const bool im = row > 32;
    const int fy = im ? row - 32 : row;
    const double ww = (fy == 0 || fy == 32) ? 1.0 : w;
    return im ? -ww * sin(tp * fy * y) : ww * cos(tp * fy * y);
  };
  for (int lane = 0; lane < 64; ++lane) {
    const int half = lane >> 5, m = lane & 31;
    for (int ks = 0; ks < 16; ++ks) {
      const int x = 2 * ks + half;                             // K index of the 32x32x2 forms: x (x axis) or y (y axis)
      for (int px = 0; px < 2; ++px) {
        const int part = m >> 4, j = m & 15, fx = 2 * j + px;
        double v;
        if (px == 0 && j == 0) v = part == 0 ? 1.0 : ((x & 1) ? -1.0 : 1.0);
        else v = part == 0 ? cos(tp * fx * x) : -sin(tp * fx * x);
        tab[TB_GX + (px * 16 + ks) * 64 + lane] = (float)v;
      }
      for (int h = 0; h < 2; ++h) {
        float* t = tab + TB_Y + ((h * 16 + ks) * 3) * 64 + lane;
        t[0] = (float)cos(tp * (((2 * m + h) * x) & 63));
        t[64] = (float)sin(tp * (((2 * m + h) * x) & 63));
        t[128] = (float)ry_val(h, m, x, 1.0);
      }
    }
    // 16x16x4 forms of the inverse: A[m = lane & 15][k = lane >> 4]
    const int mr = lane & 15, kq = lane >> 4;
    for (int hh = 0; hh < 2; ++hh)
      for (int yb = 0; yb < 2; ++yb)
        for (int ks = 0; ks < 8; ++ks) {
          const int y = 16 * yb + mr, mm = 4 * ks + kq;
          const double th = tp * (((2 * mm + hh) * y) & 63);
          tab[TB_IY + (((hh * 2 + 0) * 2 + yb) * 8 + ks) * 64 + lane] = (float)(cos(th) / T);
          tab[TB_IY + (((hh * 2 + 1) * 2 + yb) * 8 + ks) * 64 + lane] = (float)(sin(th) / T);
          tab[TB_IR + ((hh * 2 + yb) * 8 + ks) * 64 + lane] = (float)(ry_val(hh, mm, y, 2.0) / T);
        }
    for (int e = 0; e < 2; ++e)
      for (int tau = 0; tau < 4; ++tau)
        for (int kk = 0; kk < 2; ++kk)
          for (int xb = 0; xb < 2; ++xb) {
            const int x = 16 * xb + mr;
            const int part = kq & 1, j = 4 * tau + 2 * kk + (kq >> 1), fx = 2 * j + e;
            double v;
            if (e == 0 && j == 0) v = part == 0 ? 1.0 : ((x & 1) ? -1.0 : 1.0);
            else v = part == 0 ? 2.0 * cos(tp * fx * x) : -2.0 * sin(tp * fx * x);
            tab[TB_IX + (((e * 4 + tau) * 2 + kk) * 2 + xb) * 64 + lane] = (float)(v / T);
          }
  }
  int n = 0;
  auto put = [&](int rr, int ri, int kind) { slots[4 * n] = rr; slots[4 * n + 1] = ri; slots[4 * n + 2] = kind; slots[4 * n + 3] = 0; ++n; };
  for (int b = 0; b < 2; ++b) {
    const int base = b ? T : 0;
    put(base, base + T / 2, 1);
    for (int fy = 1; fy < T / 2; ++fy) put(base + fy, base + T / 2 + fy, 0);
  }
  for (int fx = 1; fx < T / 2; ++fx)
    for (int fy = 0; fy < T; ++fy) put(2 * T + 2 * T * (fx - 1) + fy, 2 * T + 2 * T * (fx - 1) + T + fy, 0);
}

void launch_fwd64(pcnn_handle h, FwdParams p, int ntile) {
  p.ntile = ntile;
  const int ntg = ntile * p.groups;
  const int nitem = 2 * ((ntg + 7) & ~7);
  const unsigned grid = (unsigned)std::min((nitem + 15) & ~15, 256);
  const size_t lds = (2 * SB_FLOATS + 6144) * sizeof(float);
  const bool masked = p.ylim < T || p.xlim < T || p.ext_y < (1 << 29) || p.ext_x < (1 << 29);
  if (masked) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spec64_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(spec64_fwd_kernel<true>, dim3(grid), dim3(512), lds, h->stream, p);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spec64_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(spec64_fwd_kernel<false>, dim3(grid), dim3(512), lds, h->stream, p);
  }
}

template <bool TANH, bool RES>
static void launch_inv64_t(pcnn_handle h, const InvParams& p, unsigned grid, int vycap, size_t lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spec64_inv_kernel<TANH, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((spec64_inv_kernel<TANH, RES>), dim3(grid), dim3(512), lds, h->stream, p, vycap);
}

void launch_inv64(pcnn_handle h, InvParams p, int ntile) {
  p.ntile = ntile;
  const int vycap = std::min(p.Vy, p.Ho);                        // <= 56: checked by the caller (pick of the tile size)
  const unsigned grid = (unsigned)std::min(ntile * p.groups, 256);
  const size_t lds = (8192 + 2 * (size_t)vycap * 256) * sizeof(float);
  if (p.act == PCNN_ACT_TANH) {
    if (p.res) launch_inv64_t<true, true>(h, p, grid, vycap, lds); else launch_inv64_t<true, false>(h, p, grid, vycap, lds);
  } else {
    p.alpha = p.act == PCNN_ACT_LINEAR ? 1.f : (p.act == PCNN_ACT_RELU ? 0.f : p.alpha);
    if (p.res) launch_inv64_t<false, true>(h, p, grid, vycap, lds); else launch_inv64_t<false, false>(h, p, grid, vycap, lds);
  }
}

}  // namespace pcnn_spec
