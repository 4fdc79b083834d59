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
#include <stdlib.h>

// Removal studies (what a kernel costs without its loads / stores) exist only in a diagnostic build: -DPCNN_REMOVAL_STUDY, bits from the
// environment variable PCNN_DBG64 (forward: 1 no stores, 2 no loads; inverse: 4 no epilogue, 8 no loads).  The shipped library compiles the
// forward kernel's tests out and never sets the bits; the INVERSE kernel keeps its two tests as run-time tests of a launch parameter that is
// always 0: with them folded away the compiler moves 28 floats of the epilogue into scratch memory (private segment 8 -> 112 bytes) and the
// kernel runs 1.00 instead of 0.66 ms per launch (profiles/r03, found by the per-kernel table).
#ifdef PCNN_REMOVAL_STUDY
#define DBG64(flags, bit) ((flags) & (bit))
#else
#define DBG64(flags, bit) 0
#endif

namespace pcnn_spec {

namespace {

constexpr int T = 64, ROWS = 4096;
// table block (floats): GX [px][ks 16][lane]; Y [h][ks 16][C S R][lane]; IY [hh][C S][yb 2][ks 8][lane]; IR [hh][yb][ks 8][lane];
// IX [ex][tau 4][kk 2][xb 2][lane]
constexpr int TB_GX = 0, TB_Y = 2048, TB_IY = 8192, TB_IR = 12288, TB_IX = 14336, TB_Y4 = 16384, TB_END = 16384 + 3 * 4096;
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

// tf.pad index map without control flow (selects only): the 32 loads of a boundary row are issued back to back
__device__ __forceinline__ int pad_sel(int i, int n, int mode) {
  const int refl = mode == PCNN_PAD_SYMMETRIC ? (i < 0 ? -i - 1 : 2 * n - 1 - i) : (i < 0 ? -i : 2 * n - 2 - i);
  const int rc = min(max(refl, 0), n - 1);
  return (unsigned)i < (unsigned)n ? i : (mode == PCNN_PAD_CONSTANT ? 0 : rc);       // constant padding: any valid pixel (replaced when consumed)
}

// issues the 32 loads of window row y: lo[ks] = column 2 ks + half, hi[ks] = column 32 + 2 ks + half (always-valid addresses; padding and
// masks are applied when the values are consumed).  y is wave-uniform.
template <bool MASKED>
__device__ __forceinline__ void load_row64(const FwdParams& p, const Ctx64& cx, int y, int half, float (&lo)[16], float (&hi)[16]) {
  if (MASKED && y >= cx.ylim) return;
  if (DBG64(p.cpt, 2)) return;                              // removal study: no window loads
  const int sy = pad_sel(cx.wy0 + y, p.H, p.pad_mode);
  const float* row = cx.img + (int64_t)sy * p.W * p.ld;
  if (cx.fast) {
    const float* rl = row, *rh = row + 32 * p.ld;                   // uniform pointers step by two pixels; ONE lane offset for all 32 loads
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      lo[ks] = rl[cx.off0];
      hi[ks] = rh[cx.off0];
      rl += 2 * p.ld; rh += 2 * p.ld;
    }
  } else {
    const int xa = cx.wx0 + half;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      lo[ks] = row[cx.off0 + (unsigned)(pad_sel(xa + 2 * ks, p.W, p.pad_mode) * p.ld)];
      hi[ks] = row[cx.off0 + (unsigned)(pad_sel(xa + 32 + 2 * ks, p.W, p.pad_mode) * p.ld)];
    }
  }
}

// x axis of one window row: D[rho][c] = sum_x GX[rho][x] (w[x] +- w[x + 32]); A = GX (lane = rho), B = the pixel's channel row
template <bool MASKED>
__device__ __forceinline__ f32x16 x_row64(const FwdParams& p, const Ctx64& cx, int y, int half, float sgx, const float (&gx)[16], const float (&lo)[16],
                                          const float (&hi)[16]) {
  f32x16 acc = zero16();
  if (MASKED && y >= cx.ylim) return acc;
  const int gy = cx.wy0 + y;
  const bool rowpad = p.pad_mode == PCNN_PAD_CONSTANT && (unsigned)gy >= (unsigned)p.H;      // uniform: the whole row is constant padding
  const bool edge = rowpad || (!cx.fast && p.pad_mode == PCNN_PAD_CONSTANT);                   // uniform: some pixels of this row are
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    float a = lo[ks], b = hi[ks];
    const int xa = 2 * ks + half, xb = xa + 32;
    if (edge) {
      if (rowpad || (unsigned)(cx.wx0 + xa) >= (unsigned)p.W) a = p.pad_value;
      if (rowpad || (unsigned)(cx.wx0 + xb) >= (unsigned)p.W) b = p.pad_value;
    }
    if (MASKED) { if (xa >= cx.xlim) a = 0.f; if (xb >= cx.xlim) b = 0.f; }
    float u = a + sgx * b;
    if (!cx.cok) u = 0.f;
    acc = mfma(gx[ks], u, acc);
  }
  return acc;
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
  auto next_item = [&](int it) { it += gridDim.x; while (it < nitem && group_of(it) >= ntg) it += gridDim.x; return it; };
  int item = blockIdx.x;
  if (group_of(item) >= ntg) item = next_item(item);
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
  // Software pipeline over the stages q = 0, 1, ... of this workgroup's items (8 per item): while the y axis consumes stage q from one LDS
  // buffer, the x axis of stage q + 1 runs on the matrix pipe from registers and lands in the other buffer, and the window row of stage q + 2
  // is in flight from global.  cy / cx / cl: the items those three stages belong to.  One LDS barrier per stage.
  Ctx64 cy, cx, cl;
  make_ctx(p, group_of(item), half, c, cy);
  cx = cy; cl = cy;
  float lo[16], hi[16];
  f32x16 Z[4][2];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) { Z[jj][0] = zero16(); Z[jj][1] = zero16(); }
  int item_l = item;                                        // item of the load stage
  load_row64<MASKED>(p, cl, srow, half, lo, hi);
  {
    const f32x16 acc = x_row64<MASKED>(p, cx, srow, half, sgx, gx, lo, hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) SB[wr_off + (8 * (r >> 2) + (r & 3)) * 32] = acc[r];
  }
  load_row64<MASKED>(p, cl, srow + 4, half, lo, hi);      // stage 1
  lds_barrier();
  int st = 0;                                               // stage of the y axis
  bool more_x = true;                                       // a stage q + 1 exists
#pragma unroll 1
  for (;;) {
    const float* const sb = SB + (st & 1) * SB_FLOATS;
    float* const sbn = SB + ((st + 1) & 1) * SB_FLOATS;
    const int stx = (st + 1) & 7;                           // stage of the x axis (stage q + 1), of item cx
    // ---- x axis of stage q + 1: its sixteen MFMAs run while the LDS reads of the y axis below are in flight
    f32x16 accx = zero16();
    if (more_x) accx = x_row64<MASKED>(p, cx, srow + 4 * stx, half, sgx, gx, lo, hi);
    // ---- the window row of stage q + 2 (item cl): its latency sits under this stage's y axis and the next stage's
    {
      const int stl = (st + 2) & 7;
      if (stl == 0 && more_x) { item_l = next_item(item_l); if (item_l < nitem) make_ctx(p, group_of(item_l), half, c, cl); }
      if (item_l < nitem) load_row64<MASKED>(p, cl, srow + 4 * stl, half, lo, hi);
    }
    // ---- y axis of stage q, rows 4 st .. 4 st + 3 and + 32: v = D[y] +- D[y + 32]; K step = two consecutive y (lane half).  The real pair
    // of the special wave takes the same four MFMAs with (R, 0) in place of (C, S)
#pragma unroll
    for (int kap = 0; kap < 2; ++kap) {
      const float* t = ty + (2 * st + kap) * (3 * 64);
      const float cyv = t[0], sny = t[64], ry = t[128];
      const float c0 = special ? ry : cyv, s0 = special ? 0.f : sny;
      const float* slo = sb + rd_off + (2 * kap) * 1024;
      const float* shi = slo + 4 * 1024;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float vr = slo[jj * 32] + sgy * shi[jj * 32];
        const float vi = slo[(16 + jj) * 32] + sgy * shi[(16 + jj) * 32];
        const float ac = jj == 0 ? c0 : cyv, as = jj == 0 ? s0 : sny;
        Z[jj][0] = mfma(ac, vr, Z[jj][0]);
        Z[jj][1] = mfma(ac, vi, Z[jj][1]);
        Z[jj][0] = mfma(as, vi, Z[jj][0]);
        Z[jj][1] = mfma(as, -vr, Z[jj][1]);
      }
    }
    if (more_x) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sbn[wr_off + (8 * (r >> 2) + (r & 3)) * 32] = accx[r];
    }
    lds_barrier();
    if (++st < 8) {
      if (st == 7) {                                        // the x stage moves on to the next item (its first row is what lo / hi receive now)
        cx = cl;
        more_x = item_l < nitem && (item_l != item);
      }
      continue;
    }
    // ---- the item's spectrum rows, straight from the accumulators (accumulator row m <-> fy = 2 m + h).  (Measured and rejected: the
    // transposed product - data as the A operand - whose accumulators hold four consecutive channels per lane: 32 16-byte stores instead of
    // 128 dword stores, but each store instruction then writes 32-byte pieces of 32 different rows: 1.24 -> 1.39 ms per 8 x 1024^2 layer.)
    float* const out = p.sp + sp_item(cy.tg, ROWS);
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
        if (DBG64(p.cpt, 1) && r > 0) continue;                  // removal study: one store instead of sixteen
        const int m = 8 * (r >> 2) + 4 * hv + (r & 3);
        int r0 = base + 2 * m, r1 = r0 + 64;
        if (jj == 0 && special) { r0 = ry_row(h, m); r1 = 64 + r0; }      // the two real columns: half-complex rows of fx = 0 and fx = 32
        __builtin_nontemporal_store(Z[jj][0][r], &out[(unsigned)(r0 * RS + cv)]);      // streamed once, read by the next kernel: see spectral_conv.hip
        __builtin_nontemporal_store(Z[jj][1][r], &out[(unsigned)(r1 * RS + cv)]);
      }
      Z[jj][0] = zero16(); Z[jj][1] = zero16();
    }
    if (!more_x) break;
    item = item_l;                                          // == the item of cx
    cy = cx;
    st = 0;
  }
}

// The same transform with a SECOND radix-2 step on the y axis (round 4): class h2 = fy mod 4 per wave, sixteen-point transforms of
// W[y] = sum_q D[y + 16 q] (-i)^(q h2) formed on the vector ALU from the four stage rows y + 16 q; real and imaginary outputs stacked in the 32 rows of ONE
// accumulator tile ([Zr; Zi] = [C S; -S C] [Wr; Wi], K step = (Wr[y], Wi[y])), so the y axis costs 16 instead of 32 MFMAs per wave and stage.
template <bool MASKED>
__global__ __launch_bounds__(512, 2) void spec64_fwd4_kernel(FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const SB = lds;                                   // [2][8 rows][32 columns][32 channels]
  float* const TY = lds + 2 * SB_FLOATS;                   // Y4 | Y4A | Y4B, each [h2][y 16][lane]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = wave & 3, pg = wave >> 2;                 // this wave's y-frequency class fy = 4 m + h2 and its group of eight column pairs
  for (int i = tid; i < 3 * 4096; i += 512) TY[i] = p.tab[TB_Y4 + i];
  const int ntg = p.ntile * p.groups;
  const int nitem = 2 * ((ntg + 7) & ~7);                  // item -> (tile group, px): px = (item >> 3) & 1, group = ((item >> 4) << 3) | (item & 7):
                                                           // with a grid that is a multiple of 16 a workgroup keeps ONE parity (its table stays in
                                                           // registers) and the two parities of a tile run on workgroups b and b + 8 (one XCD's L2)
  auto group_of = [&](int it) { return ((it >> 4) << 3) | (it & 7); };
  auto next_item = [&](int it) { it += gridDim.x; while (it < nitem && group_of(it) >= ntg) it += gridDim.x; return it; };
  int item = blockIdx.x;
  if (group_of(item) >= ntg) item = next_item(item);
  if (item >= nitem) return;
  const int px = (item >> 3) & 1;
  const bool special = px == 0 && pg == 0;                 // this wave's first pair is the two real columns (fx = 0 | 32)
  const float sgx = px ? -1.f : 1.f;
  // second radix-2 step: W[y] = sum_q D[y + 16 q] (-i)^(q h2), y < 16.  This lane's B row is the component `half` (0: Re, 1: Im) of W: from row
  // q it takes component coff[q] (float offset 0 | 16 columns) with sign sg[q].  The real pair: B = W_re (pass a) / W_im (pass b) of the REAL
  // column `half` (fx = 0 | 32), coefficients ca / cb.
  int coff[4]; float sg[4], ca[4], cb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = (q * h2) & 3;
    const bool src_im = (e & 1) ? !half : (bool)half;
    coff[q] = src_im ? 16 * 32 : 0;
    sg[q] = half == 0 ? (e < 2 ? 1.f : -1.f) : ((e == 0 || e == 3) ? 1.f : -1.f);
    ca[q] = (h2 & 1) ? ((q & 1) ? 0.f : (q == 0 ? 1.f : -1.f)) : ((h2 == 2 && (q & 1)) ? -1.f : 1.f);
    cb[q] = (h2 & 1) ? ((q & 1) ? ((q == 1) == (h2 == 1) ? -1.f : 1.f) : 0.f) : 0.f;
  }
  const bool passb = (h2 & 1) != 0;                        // the real pair's W has an imaginary part only in the odd classes
  float gx[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) gx[ks] = p.tab[TB_GX + (px * 16 + ks) * 64 + lane];
  const float* const ty = TY + h2 * (16 * 64) + lane;
  const int srow = (wave & 1) + 16 * (wave >> 1);          // this wave's window row in stage 0 (stage s: + 2 s): stage rows y = 2 s + ky, + 16 q -> buffer row ky + 2 q
  // LDS addresses of this lane: its x-axis results (row = wave) and the first column of its eight pairs
  const int wr_off = (wave * 32 + 4 * half) * 32 + c;
  const int rd_off = (8 * pg) * 32 + c;
  // Software pipeline over the stages q = 0, 1, ... of this workgroup's items (8 per item): while the y axis consumes stage q from one LDS
  // buffer, the x axis of stage q + 1 runs on the matrix pipe from registers and lands in the other buffer, and the window row of stage q + 2
  // is in flight from global.  cy / cx / cl: the items those three stages belong to.  One LDS barrier per stage.
  Ctx64 cy, cx, cl;
  make_ctx(p, group_of(item), half, c, cy);
  cx = cy; cl = cy;
  float lo[16], hi[16];
  f32x16 Z[8];                                             // pair jj: rows 0..15 Re, 16..31 Im of the sixteen frequencies of this class
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) Z[jj] = zero16();
  int item_l = item;                                        // item of the load stage
  load_row64<MASKED>(p, cl, srow, half, lo, hi);
  {
    const f32x16 acc = x_row64<MASKED>(p, cx, srow, half, sgx, gx, lo, hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) SB[wr_off + (8 * (r >> 2) + (r & 3)) * 32] = acc[r];
  }
  load_row64<MASKED>(p, cl, srow + 2, half, lo, hi);      // stage 1
  lds_barrier();
  int st = 0;                                               // stage of the y axis
  bool more_x = true;                                       // a stage q + 1 exists
#pragma unroll 1
  for (;;) {
    const float* const sb = SB + (st & 1) * SB_FLOATS;
    float* const sbn = SB + ((st + 1) & 1) * SB_FLOATS;
    const int stx = (st + 1) & 7;                           // stage of the x axis (stage q + 1), of item cx
    // ---- x axis of stage q + 1: its sixteen MFMAs run while the LDS reads of the y axis below are in flight
    f32x16 accx = zero16();
    if (more_x) accx = x_row64<MASKED>(p, cx, srow + 2 * stx, half, sgx, gx, lo, hi);
    // ---- the window row of stage q + 2 (item cl): its latency sits under this stage's y axis and the next stage's
    {
      const int stl = (st + 2) & 7;
      if (stl == 0 && more_x) { item_l = next_item(item_l); if (item_l < nitem) make_ctx(p, group_of(item_l), half, c, cl); }
      if (item_l < nitem) load_row64<MASKED>(p, cl, srow + 2 * stl, half, lo, hi);
    }
    // ---- y axis of stage q: the two rows y = 2 st + ky of the sixteen-point transforms.  K step = (Re W[y], Im W[y]) in the lane halves;
    // accumulator rows 0..15 = Re, 16..31 = Im of this class' frequencies: ONE MFMA per pair and row (the parity form took four)
#pragma unroll
    for (int ky = 0; ky < 2; ++ky) {
      const float* t = ty + (2 * st + ky) * 64;
      const float ay = t[0];
      const float* s0 = sb + rd_off + ky * 1024;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (jj == 0 && special) {
          const float* sq = s0 + half * (16 * 32);
          const float d0 = sq[0], d1 = sq[2048], d2 = sq[4096], d3 = sq[6144];
          Z[0] = mfma(t[4096], ca[0] * d0 + ca[1] * d1 + ca[2] * d2 + ca[3] * d3, Z[0]);
          if (passb) Z[0] = mfma(t[8192], cb[1] * d1 + cb[3] * d3, Z[0]);
        } else {
          const float* sq = s0 + jj * 32;
          const float w = sq[coff[0]] + sg[1] * sq[2048 + coff[1]] + sg[2] * sq[4096 + coff[2]] + sg[3] * sq[6144 + coff[3]];
          Z[jj] = mfma(ay, w, Z[jj]);
        }
      }
    }
    if (more_x) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sbn[wr_off + (8 * (r >> 2) + (r & 3)) * 32] = accx[r];
    }
    lds_barrier();
    if (++st < 8) {
      if (st == 7) {                                        // the x stage moves on to the next item (its first row is what lo / hi receive now)
        cx = cl;
        more_x = item_l < nitem && (item_l != item);
      }
      continue;
    }
    // ---- the item's spectrum rows, straight from the accumulators (accumulator row m <-> fy = 2 m + h).  (Measured and rejected: the
    // transposed product - data as the A operand - whose accumulators hold four consecutive channels per lane: 32 16-byte stores instead of
    // 128 dword stores, but each store instruction then writes 32-byte pieces of 32 different rows: 1.24 -> 1.39 ms per 8 x 1024^2 layer.)
    float* const out = p.sp + sp_item(cy.tg, ROWS);
    // the 128 store offsets are invariant across items: left alone, the compiler hoists all of them out of the persistent loop (256 registers
    // of 64-bit offsets, spilled); opaque lane coordinates make it form them here, one v_add each
    int hv = half, cv = c;
    asm volatile("" : "+v"(hv), "+v"(cv));
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int fx = 2 * (8 * pg + jj) + px;
      const int base = 128 + 128 * (fx - 1) + h2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (DBG64(p.cpt, 1) && r > 0) continue;                  // removal study: one store instead of sixteen
        const int m = 8 * (r >> 2) + 4 * hv + (r & 3);           // accumulator row: m < 16 Re, m >= 16 Im of fy = 4 (m & 15) + h2
        int row = base + 4 * (m & 15) + 64 * (m >> 4);
        if (jj == 0 && special) {                                // rows m < 16: the real column fx = 0, m >= 16: fx = 32; half-complex row order
          const int mm = m & 15, im = h2 == 0 ? (mm > 8) : (mm > 7);
          const int fy = 4 * (im ? mm - 8 : mm) + h2;
          row = 64 * (m >> 4) + (im ? 32 + fy : fy);
        }
        __builtin_nontemporal_store(Z[jj][r], &out[(unsigned)(row * RS + cv)]);      // streamed once, read by the next kernel: see spectral_conv.hip
      }
      Z[jj] = zero16();
    }
    if (!more_x) break;
    item = item_l;                                          // == the item of cx
    cy = cx;
    st = 0;
  }
}

// ------------------------------------------------------------------------------------------------------------------ inverse transform + epilogue
constexpr int NR = 7;                  // output rows per wave: rows y = wave + 8 i < 56 (tiles of >= 9 taps)

struct InvItem { int tg, q, vy; };

// POST (data-gradient launches, InvParams): the activation backward of the layer that produced this convolution's input - y2 (if given) receives
// v = conv + residual, y receives v act'(gact), and every lane adds what it stored for its four channels into bsum[((block 8 + wave) 64 + lane) 4 + j]
template <bool TANH, bool RES, bool POST = false>
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
  // items (tile group, channel half q): global item i -> q = (i >> 3) & 1, group = ((i >> 4) << 3) | (i & 7); with a grid that is a multiple of 16
  // a workgroup keeps one q and the two halves of a tile's 128-byte spectrum rows are read at the same time by workgroups b and b + 8 - one
  // XCD, one L2: each line comes from HBM once (the same workgroup taking both halves in turn fetched every line twice: 3.7 GB for 1.85 GB)
  const int nall = 2 * ((ntg + 7) & ~7);
  auto group_of = [&](int i) { return ((i >> 4) << 3) | (i & 7); };
  int nitem = 0;
  for (int i = blockIdx.x; i < nall; i += gridDim.x) nitem += group_of(i) < ntg ? 1 : 0;
  if (nitem <= 0) return;
  auto item_of = [&](int n, InvItem& it) {
    int i = blockIdx.x, seen = -1;                            // the n-th valid item of this workgroup (items are few: a short scalar walk)
    for (;; i += gridDim.x) { if (group_of(i) < ntg && ++seen == n) break; }
    it.tg = group_of(i); it.q = (i >> 3) & 1;
    int t = p.tile0 + it.tg / p.groups;
    t /= p.tiles_x;
    it.vy = min(p.Vy, p.Ho - (t % p.tiles_y) * p.Vy);
  };
  // spectrum rows of pair j = 4 tau + s of this wave's x parity, 16 channels: z[16 hh + 8 part + ks] = row (fy = 2 (4 ks + g4) + hh) of this
  // lane's channel.  Loaded 16 bytes per lane - lane (g4, quad q, i) fetches channels 4 q .. 4 q + 3 of the rows of K steps 4 b + i - and brought
  // into the per-channel operand layout by quad transposes when consumed (fix_pair): 8 loads instead of 32 (1.26 -> 1.14 ms per 8 x 1024^2 layer)
  auto load_pair = [&](const InvItem& it, int tau, float (&z)[32]) {
    int g4v = lane >> 4, cv = c;
    asm volatile("" : "+v"(g4v), "+v"(cv));                   // opaque: the row offsets are formed per call, not hoisted out of the item loop
    if (p.cpt & 2) return;                                     // removal study (diagnostic builds set the bit; see DBG64 below for why this stays a run-time test)
    const int iq = cv & 3;
    const float* in = p.sp + sp_item(it.tg, ROWS) + 16 * it.q + (cv & ~3);
    const int j = 4 * tau + s;
    const bool real = ex == 0 && j == 0;
    const float* src = real ? in : in + (128 + 128 * (2 * j + ex - 1)) * RS;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int kidx = 4 * (4 * b + iq) + g4v;              // K index of this lane's load: K step 4 b + iq, row g4 of the step
        const int row = real ? ry_row(hh, kidx) : 2 * kidx + hh;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (unsigned)(row * RS));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (unsigned)((64 + row) * RS));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { z[16 * hh + 4 * b + jj] = v0[jj]; z[16 * hh + 8 + 4 * b + jj] = v1[jj]; }
      }
  };
  auto fix_pair = [&](float (&z)[32]) {
    const bool odd = c & 1, upper = c & 2;
#pragma unroll
    for (int b = 0; b < 8; ++b) quad_transpose(z[4 * b], z[4 * b + 1], z[4 * b + 2], z[4 * b + 3], odd, upper);
  };
  f32x4 XE[NR][2], XO[NR][2];
#pragma unroll
  for (int i = 0; i < NR; ++i) { XE[i][0] = zero4(); XE[i][1] = zero4(); XO[i][0] = zero4(); XO[i][1] = zero4(); }
  float ymax = 0.f;
  f32x4 bsum = zero4();                                       // POST: this lane's share of the bias gradient (its four channels; a workgroup keeps one q)
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
      fix_pair(z);
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
      // ---- B(k - 1): this wave's output rows y = wave + 8 i; columns 0..7 feed the even-frequency half E_x, 8..15 the odd half O_x.
      // The product is taken TRANSPOSED - the data is the A operand (rows = channels), the table the B operand (columns = pixels) - so that an
      // accumulator's four registers are four consecutive CHANNELS of one pixel: the epilogue stores 16 bytes per lane (a quarter of the store
      // instructions of the channel-per-lane form, which bounded this kernel: 112 dword stores per wave and item)
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
            XE[i][0] = mfma16(be, ax[0][kk][0], XE[i][0]);
            XE[i][1] = mfma16(be, ax[0][kk][1], XE[i][1]);
            XO[i][0] = mfma16(bo, ax[1][kk][0], XO[i][0]);
            XO[i][1] = mfma16(bo, ax[1][kk][1], XO[i][1]);
          }
        }
      }
      if (tau == 3) {
        // ---- epilogue of item ib from the accumulators: out[x] = E + O (x < 32), E - O (x >= 32); lane = (channel quad g4, pixel lane & 15)
        const int g = ib.tg % p.groups;
        int t = p.tile0 + ib.tg / p.groups;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int tyy = t % p.tiles_y;
        const int n = t / p.tiles_y;
        const int y0 = tyy * p.Vy, x0 = tx * p.Vx;
        const int vx = min(p.Vx, p.Wo - x0);
        const int cc = 16 * ib.q + 4 * g4, chan = g * p.cstride + cc;       // this lane's four channels: chan .. chan + 3
        const int nv = max(0, min(4, min(p.cvalid - cc, p.C - chan)));       // how many of them exist
        const bool vec = nv == 4 && (p.cpt & 4) && (chan & 3) == 0;           // 16-byte accesses (host: pointers and strides are multiples of 16 B)
        float bias[4], sc[4], sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bias[j] = (p.bias && j < nv) ? p.bias[chan + j] : 0.f;
          sc[j] = (p.bn_scale && j < nv) ? p.bn_scale[chan + j] : 1.f;
          sh[j] = (p.bn_scale && j < nv) ? p.bn_shift[chan + j] : 0.f;
        }
        const int xsgn = p.flip ? -1 : 1;
        const int xl = c;                                        // pixel inside a 16-pixel block
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int y = wave + 8 * i;
          if (y < ib.vy && nv > 0 && !(p.cpt & 1)) {             // (p.cpt & 1: removal study - no epilogue)
            const int64_t rowpix = p.flip ? ((int64_t)n * p.Ho + (p.Ho - 1 - y0 - y)) * p.Wo + (p.Wo - 1 - x0) : ((int64_t)n * p.Ho + y0 + y) * p.Wo + x0;
            float* yrow = p.y + rowpix * p.ldy;
            float* arow = (!POST && p.act_out) ? p.act_out + rowpix * p.ld_act : nullptr;
            const float* rrow = RES ? p.res + rowpix * p.ld_res : nullptr;
            const float* grow = POST ? p.gact + rowpix * p.ld_gact : nullptr;
            float* y2row = (POST && p.y2) ? p.y2 + rowpix * p.ld_y2 : nullptr;
            int chv = chan;
            asm volatile("" : "+v"(chv));                         // opaque: the pixel offsets below are formed here, not hoisted out of the item loop
            f32x4 rv[4], gv[4];
            if (POST) {                                            // the producing layer's activation output: same burst form as the residual
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int xx = 32 * (e >> 1) + 16 * (e & 1) + xl;
                const float* gp = grow + (xx < vx ? (int)(xsgn * xx * p.ld_gact) : 0) + chv;
                if (vec) gv[e] = *reinterpret_cast<const f32x4*>(gp);
                else {
#pragma unroll
                  for (int j = 0; j < 4; ++j) gv[e][j] = j < nv ? gp[j] : 0.f;
                }
              }
            }
            if (RES) {                                             // the row's residual values first, as one burst of loads (no load between stores)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int xx = 32 * (e >> 1) + 16 * (e & 1) + xl;
                const float* rp = rrow + (xx < vx ? (int)(xsgn * xx * p.ld_res) : 0) + chv;
                if (vec) rv[e] = *reinterpret_cast<const f32x4*>(rp);
                else {
#pragma unroll
                  for (int j = 0; j < 4; ++j) rv[e][j] = j < nv ? rp[j] : 0.f;
                }
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int hx = e >> 1, xb = e & 1;
              const int xx = 32 * hx + 16 * xb + xl;
              if (xx < vx) {
                f32x4 a4, o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  float v = hx ? XE[i][xb][j] - XO[i][xb][j] : XE[i][xb][j] + XO[i][xb][j];
                  if (!POST) {                                     // (a data-gradient launch has no bias, activation, BN or act_out)
                    v += bias[j];
                    v = TANH ? tanhf(v) : (v > 0.f ? v : v * p.alpha);
                    a4[j] = v;
                    v = v * sc[j] + sh[j];
                  }
                  if (RES) v += rv[e][j];
                  if (POST) {
                    a4[j] = v;                                     // the un-multiplied gradient (skip connection): y2
                    const float g = gv[e][j];
                    v *= p.gmode == PCNN_ACT_TANH ? 1.f - g * g : (g > 0.f ? 1.f : p.galpha);
                    if (j < nv) bsum[j] += v;
                  }
                  o4[j] = v;
                  if (j < nv) ymax = fmaxf(ymax, fabsf(v));
                }
                const int xo = xsgn * xx;
                float* second = POST ? y2row : arow;               // the second output: activation (forward) or raw gradient (POST)
                const int ld2 = POST ? p.ld_y2 : p.ld_act;
                if (vec) {
                  if (second) *reinterpret_cast<f32x4*>(second + (int)(xo * ld2) + chv) = a4;
                  *reinterpret_cast<f32x4*>(yrow + (int)(xo * p.ldy) + chv) = o4;
                } else {
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    if (j < nv) {
                      if (second) second[(int)(xo * ld2) + chv + j] = a4[j];
                      yrow[(int)(xo * p.ldy) + chv + j] = o4[j];
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
  if (POST && p.bsum) {                                        // own slot: launches of one call follow each other on the stream
    float* bs = p.bsum + ((size_t)(blockIdx.x * 8 + wave) * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] += bsum[j];
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

// POST: dbias[ch] from the lanes' partial sums of spec64_inv_kernel<.., POST>, added in a fixed order.  Channel ch = 16 q + 4 g4 + j lives in the workgroups
// with (block >> 3) & 1 == q (the grid is a multiple of 16: a workgroup keeps one q), lanes with lane >> 4 == g4, component j.  One workgroup per channel.
__global__ __launch_bounds__(256) void spec_post_bias64_kernel(const float* __restrict__ bsum, int nblocks, float* __restrict__ dbias) {
  __shared__ float red[256];
  const int ch = blockIdx.x, t = threadIdx.x, q = ch >> 4, g4 = (ch & 15) >> 2, j = ch & 3;
  float a = 0.f;
  for (int sl = t; sl < nblocks * 8 * 16; sl += 256) {          // (block, wave, pixel lane)
    const int px = sl & 15, bw = sl >> 4, block = bw >> 3;
    if (((block >> 3) & 1) == q) a += bsum[((size_t)bw * 64 + g4 * 16 + px) * 4 + j];
  }
  red[t] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (t < st) red[t] += red[t + st]; __syncthreads(); }
  if (t == 0) dbias[ch] = red[0];
}

}  // namespace

void launch_post_bias64(pcnn_handle h, const float* bsum, int nblocks, int C, float* dbias) {
  hipLaunchKernelGGL(spec_post_bias64_kernel, dim3((unsigned)C), dim3(256), 0, h->stream, bsum, nblocks, dbias);
}

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
    // radix-4 y axis of spec64_fwd4_kernel: A[m][k = half] for K step y < 16 of class h2.  Complex pairs (Y4): rows m < 16 Re, m >= 16 Im of
    // fy = 4 (m & 15) + h2; k = 0 multiplies Re W[y], k = 1 Im W[y]: [C S; -S C].  The real pair: rows m < 16 belong to the real column fx = 0
    // and take k = 0 only, rows m >= 16 to fx = 32 and take k = 1 only; pass a (Y4A) multiplies Re W, pass b (Y4B) Im W; row order = the
    // half-complex order of the class (class 0: Re fy = 0, 4 .. 32 then Im fy = 4 .. 28; else Re of its 8 frequencies <= 31, then their Im).
    for (int h2 = 0; h2 < 4; ++h2)
      for (int y = 0; y < 16; ++y) {
        const int mm = m & 15;
        const double th = tp * (((4 * mm + h2) * y) & 63);
        tab[TB_Y4 + (h2 * 16 + y) * 64 + lane] = (float)(m < 16 ? (half ? sin(th) : cos(th)) : (half ? cos(th) : -sin(th)));
        const bool im = h2 == 0 ? mm > 8 : mm > 7;
        const int fy = 4 * (im ? mm - 8 : mm) + h2;
        const double ts = tp * ((fy * y) & 63);
        const bool mine = (m >= 16) == (half == 1);
        tab[TB_Y4 + 4096 + (h2 * 16 + y) * 64 + lane] = mine ? (float)(im ? -sin(ts) : cos(ts)) : 0.f;
        tab[TB_Y4 + 8192 + (h2 * 16 + y) * 64 + lane] = (mine && (h2 & 1)) ? (float)(im ? cos(ts) : sin(ts)) : 0.f;
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

#ifdef PCNN_REMOVAL_STUDY
static int dbg64() { static const int v = getenv("PCNN_DBG64") ? atoi(getenv("PCNN_DBG64")) : 0; return v; }
#else
static int dbg64() { return 0; }
#endif

void launch_fwd64(pcnn_handle h, FwdParams p, int ntile) {
  p.ntile = ntile;
  p.cpt = dbg64() & 3;
  const int ntg = ntile * p.groups;
  const int nitem = 2 * ((ntg + 7) & ~7);
  const unsigned grid = (unsigned)std::min((nitem + 15) & ~15, 256);
  const bool masked = p.ylim < T || p.xlim < T || p.ext_y < (1 << 29) || p.ext_x < (1 << 29);
  // Unmasked windows take the form with the second radix-2 step on the y axis (spec64_fwd4_kernel: a third less matrix-core work, measured 1.154 ->
  // 1.080 ms per 8 x 1024^2 x 32-channel launch); the masked form stays on the parity kernel (its radix-4 build needs 20 bytes of scratch per lane and
  // runs 0.393 -> 0.415 ms).  PCNN_FWD64_RADIX = 2 | 4 forces one form for both (tests, A/B timing).
  const int forced = getenv("PCNN_FWD64_RADIX") ? atoi(getenv("PCNN_FWD64_RADIX")) : 0;
  const int radix = forced ? forced : (masked ? 2 : 4);
  if (radix == 4) {
    const size_t lds4 = (2 * SB_FLOATS + 3 * 4096) * sizeof(float);
    if (masked) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spec64_fwd4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      hipLaunchKernelGGL(spec64_fwd4_kernel<true>, dim3(grid), dim3(512), lds4, h->stream, p);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spec64_fwd4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      hipLaunchKernelGGL(spec64_fwd4_kernel<false>, dim3(grid), dim3(512), lds4, h->stream, p);
    }
    return;
  }
  const size_t lds = (2 * SB_FLOATS + 6144) * sizeof(float);
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
  p.cpt = (dbg64() >> 2) & 3;
  // bit 2: every tensor the epilogue touches allows 16-byte accesses at channel offsets that are multiples of four
  auto al = [](const void* q, int ld) { return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0); };
  if (al(p.y, p.ldy) && al(p.act_out, p.ld_act) && al(p.res, p.ld_res) && al(p.gact, p.ld_gact) && al(p.y2, p.ld_y2) && (p.cstride & 3) == 0) p.cpt |= 4;
  const int vycap = std::min(p.Vy, p.Ho);                        // <= 56: checked by the caller (pick of the tile size)
  const int ntg = ntile * p.groups;
  const unsigned grid = (unsigned)std::min((2 * ((ntg + 7) & ~7) + 15) & ~15, 256);
  const size_t lds = (8192 + 2 * (size_t)vycap * 256) * sizeof(float);
  if (p.gact) {                                                  // data gradient + the producer's activation backward (linear conv epilogue)
    p.alpha = 1.f;
    p.galpha = p.gmode == PCNN_ACT_LINEAR ? 1.f : (p.gmode == PCNN_ACT_RELU ? 0.f : p.galpha);
    if (p.res) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spec64_inv_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((spec64_inv_kernel<false, true, true>), dim3(grid), dim3(512), lds, h->stream, p, vycap);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spec64_inv_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((spec64_inv_kernel<false, false, true>), dim3(grid), dim3(512), lds, h->stream, p, vycap);
    }
    return;
  }
  if (p.act == PCNN_ACT_TANH) {
    if (p.res) launch_inv64_t<true, true>(h, p, grid, vycap, lds); else launch_inv64_t<true, false>(h, p, grid, vycap, lds);
  } else {
    p.alpha = p.act == PCNN_ACT_LINEAR ? 1.f : (p.act == PCNN_ACT_RELU ? 0.f : p.alpha);
    if (p.res) launch_inv64_t<false, true>(h, p, grid, vycap, lds); else launch_inv64_t<false, false>(h, p, grid, vycap, lds);
  }
}

}  // namespace pcnn_spec
