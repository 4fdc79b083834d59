// Tiled spectral convolution in exact fp32 on the CDNA4 matrix cores: the DFT itself is a GEMM.
//
// For the wide filters of the Poisson CNN (15x15 ... 5x5, 32 channels) the direct implicit GEMM spends k*k*Cin*Cout MACs per pixel -
// at the fp32 MFMA rate that is what bounds the whole training step.  Overlap-save in the frequency domain needs ~40x fewer FLOP, but an
// FFT butterfly network is VALU/LDS work.  On an MFMA machine the 32-point DFT is instead applied as a 32x32 real matrix product
// (v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation - the same instruction, hence the same arithmetic class, as the direct
// kernel), with the CHANNEL index in the N / lane dimension, so that every global access of every kernel below is a pixel's (or a
// spectrum row's) 128-byte channel vector and nothing is ever transposed:
//
//   spec_fwd_kernel   window (32 x 32 pixels, boundary-condition padding applied by the loader) -> 1024 spectrum rows per tile:
//                     x axis real -> half-complex (G), y axis: the two real columns (fx = 0, 16) half-complex again, the 15 complex
//                     columns a full complex DFT (Fc).  Intermediate U[y][s][c] lives in LDS (128 KB), one workgroup of 8 waves per tile.
//   spec_mix_kernel   per frequency slot: [Yr | Yi] = [Xr | Xi] M_f over all tiles (M = tiles, K = 2 Cin, N = 2 Cout); M_f holds
//                     conj(W^)[f] as a real 2x2 block matrix and stays in registers while the tiles stream through.
//   spec_inv_kernel   inverse transforms + the fused conv epilogue (bias, activation, BN affine, residual, act_out, max|y|) on the valid
//                     (33-k)^2 outputs of the tile, stored as NHWC channel rows straight from the accumulators.
//   spec_wmix_kernel  weight gradient: P_f = sum_tiles [Xr | Xi]^T [Dr | Di] (K = tiles), combined to X^ conj(D^) and inverse-transformed
//                     by spec_inv_kernel as a one-tile "image" with Cin*Cout channels; the first k x k taps are dw.
// The filter spectrum is produced by spec_fwd_kernel too (the filter is a kh x kw image with Cin*Cout channels).
// Index conventions: tools/spectral_model.py (numpy model, checked against direct correlation).
#include "spectral_common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

int pcnn_spectral_conv_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                           const float* bn_shift, const float* residual, float* y, float* act_out);
int pcnn_spectral_conv_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw);
bool pcnn_spectral_eligible(pcnn_handle h, const pcnn_conv_desc* d, bool wgrad);
bool pcnn_conv_small_fwd_eligible(const pcnn_conv_desc* d);   // conv_small.hip
bool pcnn_conv_fwd_takes_narrow_route(pcnn_handle h, const pcnn_conv_desc* d);

using namespace pcnn_spec;

// Non-temporal hints on the spectrum streams (a spectrum is written by one kernel and read once by the next, gigabytes apart).  Measured per
// access at 8 x 1024^2, 7 and 9 taps (forward / weight gradient / fused backward of one layer, same box, against no hints):
//   1  forward-transform stores   -1 % / -2.5 % / -2.3 %        2  mixing loads    +6 % / +1 % / +5 %       4  mixing stores   -0.5 % (noise)
//   8  inverse-transform loads     0                            16  weight-gradient GEMM loads   0 / -1 % / -0.5 %
// so the hint is kept on the forward transforms' stores (32- and 64-point) and on the weight-gradient loads: 1 + 16.
#ifndef PCNN_NT
#define PCNN_NT 17
#endif
#define NT_LOAD(bit, p) ((PCNN_NT & (bit)) ? __builtin_nontemporal_load(p) : *(p))
#define NT_STORE(bit, v, p) do { if (PCNN_NT & (bit)) __builtin_nontemporal_store(v, p); else *(p) = (v); } while (0)

namespace {

constexpr int T = 32, ROWS = 1024, NSLOT = 512;
// tables (all stored [k][m], 32 x 32): G / GI the real <-> half-complex 32-point transforms; F2[parity] / FI2[parity] the two 16-point complex
// transforms of one radix-2 step of the 32-point complex DFT (see build_tables); GI2 the real-column inverse split into its two parities
constexpr int TAB_G = 0, TAB_GI = 1024, TAB_F2 = 2048, TAB_FI2 = 4096, TAB_GI2 = 6144, TAB_FLOATS = 7168;
constexpr size_t LDS_U = (size_t)T * T * 32 * sizeof(float);   // 128 KB
constexpr size_t LDS_INV = LDS_U + 1024 * sizeof(float);       // the inverse kernel keeps the real-column table (GI2) behind U: see spec_inv_kernel
// (a storage order with rows blocked 32 tiles wide - whole DRAM pages for the per-frequency kernels - was measured in round 2: no gain)
__host__ __device__ __forceinline__ int64_t sp_item(int64_t item) { return pcnn_spec::sp_item(item, ROWS); }

// ------------------------------------------------------------------------------------------------------------------ forward transform
// one (tile, channel group) work item as the loader sees it: a wave-uniform image base plus 32-bit per-lane offsets (so the loads take
// the scalar-base + vector-offset form and no 64-bit address lives in vector registers)
struct FwdItem {
  const float* img;      // uniform: image n
  int wy0;               // uniform: window origin row
  unsigned off[16];      // lane: float offset of window column 2 xs + half (clamped into the image) + this lane's channel
  unsigned cmask, zmask; // bit xs: column is constant padding / lies beyond xlim (zero)
  int ylim, xlim;        // uniform: rows / columns of this item's window that can be non-zero
  bool cok;              // this lane's channel exists
};

__device__ __forceinline__ void fwd_item(const FwdParams& p, int item, int half, int c, FwdItem& it) {
  const int g = item % p.groups;
  int t = p.tile0 + item / p.groups;
  const int txg = t % p.tgx; t /= p.tgx;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int sub = p.pack > 1 ? c / p.cpt : 0, cc = c - sub * p.cpt;            // this lane's tile of the group, and its channel
  const int tx = txg * p.pack + sub;
  const int chan = g * p.cstride + cc;
  it.cok = cc < p.cvalid && chan < p.C && tx < p.tiles_x;
  it.img = p.x + (int64_t)n * p.H * p.W * p.ld;
  it.wy0 = ty * p.Vy - p.oy;
  const int wx0 = tx * p.Vx - p.ox;
  it.ylim = min(p.ylim, p.ext_y - ty * p.Vy);
  it.xlim = min(p.xlim, p.ext_x - txg * p.pack * p.Vx);                       // uniform bound (the group's first tile); per lane: zmask
  const int xlim_lane = min(p.xlim, p.ext_x - tx * p.Vx);
  it.cmask = 0u; it.zmask = 0u;
  if (p.pack == 1 && wx0 >= 0 && wx0 + T <= p.W) {                          // (uniform) the window lies inside the image in x: no index maps
    const unsigned base = (unsigned)((wx0 + half) * p.ld + (it.cok ? chan : 0));
    const int nz = min(16, max(0, (xlim_lane - half + 1) >> 1));            // columns 2 xs + half < xlim
    it.zmask = 0xffffu & ~((1u << nz) - 1u);
#pragma unroll
    for (int xs = 0; xs < 16; ++xs) it.off[xs] = base + (unsigned)(2 * xs * p.ld);
    return;
  }
#pragma unroll
  for (int xs = 0; xs < 16; ++xs) {
    const int xc = 2 * xs + half;
    const int sx = pcnn_pad_index(wx0 + xc, p.W, p.pad_mode);
    if (sx < 0) it.cmask |= 1u << xs;
    if (xc >= xlim_lane) it.zmask |= 1u << xs;
    it.off[xs] = (unsigned)((sx < 0 ? 0 : sx) * p.ld + (it.cok ? chan : 0));
  }
}
// issues the 16 loads of window row y (always-valid addresses; padding is applied when the values are consumed).  y is wave-uniform.
template <bool MASKED>
__device__ __forceinline__ void fwd_load_row(const FwdParams& p, const FwdItem& it, int y, float (&v)[16]) {
  if (MASKED && y >= it.ylim) return;                                            // zero row (gradient windows hold Vy x Vx values): nothing to fetch
  const int sy = pcnn_pad_index(it.wy0 + y, p.H, p.pad_mode);
  const float* row = it.img + (int64_t)(sy < 0 ? 0 : sy) * p.W * p.ld;
#pragma unroll
  for (int xs = 0; xs < 16; ++xs) v[xs] = row[it.off[xs]];
}
template <bool MASKED>
__device__ __forceinline__ f32x16 fwd_row_mfma(const FwdParams& p, const FwdItem& it, int y, const float (&greg)[16], const float (&v)[16]) {
  const int sy = pcnn_pad_index(it.wy0 + y, p.H, p.pad_mode);
  f32x16 acc = zero16();
  if (MASKED && y >= it.ylim) return acc;
#pragma unroll
  for (int xs = 0; xs < 16; ++xs) {
    if (!MASKED || 2 * xs < it.xlim) {                                             // uniform: columns beyond xlim are zero, their K steps are skipped
      float val = (sy < 0 || ((it.cmask >> xs) & 1u)) ? p.pad_value : v[xs];
      if (!it.cok || (MASKED && ((it.zmask >> xs) & 1u))) val = 0.f;
      acc = mfma(greg[xs], val, acc);
    }
  }
  return acc;
}

// Persistent: one workgroup (8 waves) per CU walks the (tile, group) items.  A wave's four window rows of the NEXT item are requested
// before the y-axis phase of the current one (four register sets), so the global-load latency sits under that phase's MFMAs.
// y axis: one radix-2 decimation-in-frequency step on the vector ALU (u[y] +- u[y + 16], in-lane), then the even / odd output frequencies
// are two 16-point complex DFTs = two real 32 x 32 products (F2[0], F2[1]) - half the matrix-core work of the 64 x 64 real form.
// MASKED: the window holds values only in its first ylim x xlim entries (gradient tiles): their zero rows / columns are skipped.
// FENCE: fetch a unit's LDS operands as one burst before its MFMA chain (see lds_fence).
template <bool MASKED, bool FENCE>
__global__ __launch_bounds__(512, 1) void spec_fwd_kernel(FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float U[];          // U[(y*32 + s)*32 + c]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave & 1, q = wave >> 1;
  const int total = p.ntile * p.groups;
  int item = blockIdx.x;
  if (item >= total) return;
  float greg[16], freg[16];
#pragma unroll
  for (int xs = 0; xs < 16; ++xs) greg[xs] = p.tab[TAB_G + (2 * xs + half) * 32 + c];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) freg[ks] = p.tab[TAB_F2 + h * 1024 + (2 * ks + half) * 32 + c];
  FwdItem cur;
  fwd_item(p, item, half, c, cur);
  float v[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j) fwd_load_row<MASKED>(p, cur, wave + 8 * j, v[j]);
  // ---- x axis: D_y[s][c] = sum_x G[s][x] xw[y][x][c]; A = G (lane = s), B = the pixel's channel row (lane = c), two x per MFMA
  auto x_phase = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = wave + 8 * j;
      const f32x16 acc = fwd_row_mfma<MASKED>(p, cur, y, greg, v[j]);
#pragma unroll
      for (int r = 0; r < 16; ++r) U[(y * 32 + acc_row(r, half)) * 32 + c] = acc[r];
    }
  };
  // The first item's x phase stands OUTSIDE the loop and the y phase is unrolled over its four units: every x phase inside the loop then
  // meets the same, statically known memory-counter state (the next item's loads, then the current item's stores), so the s_waitcnt vmcnt
  // values the compiler derives for it do not have to be the ones of a first pass that has no stores pending.
  x_phase();
  for (;;) {
    const int next = item + gridDim.x;
    const int ylim = cur.ylim;                                       // `cur` describes the NEXT item from here on
    if (next < total) {
      fwd_item(p, next, half, c, cur);
#pragma unroll
      for (int j = 0; j < 4; ++j) fwd_load_row<MASKED>(p, cur, wave + 8 * j, v[j]);
    }
    lds_barrier();
    // ---- y axis.  Wave (q, h), unit u: complex column fx = 1 + q + 4 u, or - q == 3, u == 3 - the real column 0 (h = 0) / 16 (h = 1);
    // h = parity of the output frequencies fy the wave produces.
    float* out = p.sp + sp_item(item);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool realcol = q == 3 && u == 3;                         // uniform
      int fx = 1 + q + 4 * u;
      asm volatile("" : "+s"(fx));                                   // opaque: the unit's LDS / store offsets are formed here, from this scalar, and
                                                                     // not hoisted out of the item loop for all four units at once (registers)
      float bu[16];
      f32x16 acc = zero16();
      float* o;
      if (!realcol) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {                            // K step ks: k = 2 ks + half = 16 part + y, y < 16
          const int y = 2 * (ks & 7) + half, sc = ks < 8 ? fx : 16 + fx;
          const float lo = U[(y * 32 + sc) * 32 + c];
          const float hi = (!MASKED || 16 + 2 * (ks & 7) < ylim) ? U[((y + 16) * 32 + sc) * 32 + c] : 0.f;     // rows beyond ylim are zero (uniform test)
          bu[ks] = h ? lo - hi : lo + hi;
        }
        if (FENCE) lds_fence();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = mfma(freg[ks], bu[ks], acc);
        o = out + (64 + 64 * (fx - 1) + h) * RS;                     // accumulator row 16 part + m  ->  spectrum row 32 part + 2 m + h
      } else {
        const int col = h ? 16 : 0;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          if (!MASKED || 2 * ks < ylim) bu[ks] = U[((2 * ks + half) * 32 + col) * 32 + c];
        if (FENCE) lds_fence();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          if (!MASKED || 2 * ks < ylim) acc = mfma(greg[ks], bu[ks], acc);
        o = out + (h ? 32 : 0) * RS;
      }
      // spectrum row of accumulator row `row`: complex columns interleave the parities, the real column stores its 32 entries in order
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, half);
        const int srow = realcol ? row : 32 * (row >> 4) + 2 * (row & 15);
        NT_STORE(1, acc[r], &o[(unsigned)(srow * RS + c)]);
      }
    }
    if (next >= total) break;
    item = next;
    lds_barrier();                                                       // U is free for the next item
    x_phase();
  }
}

// ------------------------------------------------------------------------------------------------------------------ inverse transform + epilogue

// unit u of wave (q, h): complex column fx = 1 + q + 4u, input frequencies fy of parity h (K = (part, m), fy = 2 m + h), or - q == 3, u == 3 -
// the real column 0 / 16 (all 32 half-complex entries).  `in` is the item's uniform base, `loff` = RS half + c the lane's offset.
__device__ __forceinline__ void inv_load_unit(const float* in, int q, int h, unsigned loff, int u, float (&b)[16]) {
  const bool realcol = q == 3 && u == 3;
  if (realcol) {
    const float* src = in + (h ? 32 : 0) * RS;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) b[ks] = NT_LOAD(8, &src[loff + (unsigned)(2 * RS) * ks]);                    // row 2 ks + half
  } else {
    const float* src = in + (64 + 64 * (q + 4 * u) + h) * RS;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) b[ks] = NT_LOAD(8, &src[2u * loff - (loff & 31u) + (unsigned)RS * (32 * (ks >> 3) + 4 * (ks & 7))]);   // row 32 part + 2 (2 (ks & 7) + half) + h
  }
}

// y axis: one radix-2 decimation-in-time step - wave (q, h) produces E (h = 0) or O (h = 1) for rows y < 16 of its columns, two real
// 32 x 32 products (FI2) instead of the 64 x 64 real form; the x-axis phase forms u[y] = E + O (y < 16) or E - O (y >= 16) while it
// reads its operands.  LDS: E[(y*32 + s)*32 + c] in the first 64 KB, O in the second.
// TANH = false: linear / relu / leaky-relu as one select with the negative-side slope in p.alpha (1 / 0 / alpha) - sixteen inlined
// tanhf bodies would otherwise cost every layer ~90 registers and spills
template <bool TANH, bool RES, bool POST = false>
__global__ __launch_bounds__(512, 1) void spec_inv_kernel(InvParams p) {
  extern __shared__ __attribute__((aligned(16))) float U[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave & 1, q = wave >> 1;
  const unsigned loff = (unsigned)RS * half + c;
  const int total = p.ntile * p.groups;
  int item = blockIdx.x;
  if (item >= total) return;
  // The table of the two real columns (GI2) is used by two of the eight waves, once per item: it lives in LDS behind U and is read where it is used
  // (16 transient registers) instead of occupying 16 registers of EVERY wave for the whole kernel - those registers are what lets the epilogue
  // request a whole output row's residual / activation values in ONE burst (BURST below).
  float* const G2 = U + T * T * 32;
  for (int i = tid; i < 1024; i += 512) G2[i] = p.tab[TAB_GI2 + i];
  float fireg[16], gireg[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    fireg[ks] = p.tab[TAB_FI2 + h * 1024 + (2 * ks + half) * 32 + c];
    gireg[ks] = p.tab[TAB_GI + (2 * ks + half) * 32 + c];
  }
  __syncthreads();
  float* const Uh = U + h * (16 * 32 * 32);
  float b[4][16];
  {
    const float* in = p.sp + sp_item(item);
#pragma unroll
    for (int u = 0; u < 4; ++u) inv_load_unit(in, q, h, loff, u, b[u]);
  }
  float ymax = 0.f;
  // values of an output row whose residual / activation inputs are requested together, before any of the row's stores: the memory counter is
  // in-order, so a burst issued behind the previous burst's stores waits until those stores have COMPLETED - every burst boundary is one exposed
  // write latency.  One burst per row (two where both inputs are read) instead of two (four): an input stream cost 0.32 ms per 8 x 1024^2 x 32
  // launch, now 0.10 (residual), 0.25 -> 0.06 (POST), 0.84 -> 0.34 (both) - tools/probe_epilogue_variants.py.
  constexpr int BURST = POST ? (RES ? 8 : 16) : 16;
  float bsum = 0.f;                                                  // POST: sum of what this lane stored (its channel's bias-gradient share)
  for (;;) {
    const int next = item + gridDim.x;
    const float* nin = p.sp + sp_item(next);
    // ---- y axis inverse: complex columns -> E / O [y][fx] (real part) and [y][16 + fx] (imaginary part); real columns -> [y][0], [y][16]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool realcol = q == 3 && u == 3;
      f32x16 acc = zero16();
      if (!realcol) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = mfma(fireg[ks], b[u][ks], acc);
        if (next < total) inv_load_unit(nin, q, h, loff, u, b[u]);  // the next item's unit u: lands under the rest of this item
        const int fx = 1 + q + 4 * u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = acc_row(r, half);                          // 16 part + y
          Uh[((row & 15) * 32 + 16 * (row >> 4) + fx) * 32 + c] = acc[r];
        }
      } else {                                                       // rows 16 par + y: both parities from this one wave
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = mfma(G2[(2 * ks + half) * 32 + c], b[u][ks], acc);
        if (next < total) inv_load_unit(nin, q, h, loff, u, b[u]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = acc_row(r, half);
          U[(row >> 4) * (16 * 32 * 32) + ((row & 15) * 32 + (h ? 16 : 0)) * 32 + c] = acc[r];
        }
      }
    }
    lds_barrier();  
    // ---- x axis inverse on the valid rows + epilogue (lane = channel: per-channel constants are per-lane scalars)
    {
      const int g = item % p.groups;
      int t = p.tile0 + item / p.groups;
      const int txg = t % p.tgx; t /= p.tgx;
      const int ty = t % p.tiles_y;
      const int n = t / p.tiles_y;
      const int sub = p.pack > 1 ? c / p.cpt : 0, cc = c - sub * p.cpt;      // this lane's tile of the group, and its channel
      const int subx = sub * p.Vx;                                            // ... and that tile's x offset from the group's first tile
      const int y0 = ty * p.Vy, x0 = txg * p.pack * p.Vx;
      const int vy = min(p.Vy, p.Ho - y0), vx = min(p.Vx, p.Wo - x0 - subx);
      const int chan = g * p.cstride + cc;
      const bool cok = cc < p.cvalid && chan < p.C;
      const float bias = (p.bias && cok) ? p.bias[chan] : 0.f;
      const float sc = (p.bn_scale && cok) ? p.bn_scale[chan] : 1.f, sh = (p.bn_scale && cok) ? p.bn_shift[chan] : 0.f;
#pragma unroll 1
      for (int yy = wave; yy < vy; yy += 8) {
        float bu[16];
        const float osign = yy < 16 ? 1.f : -1.f;                    // u[y] = E[y & 15] +- O[y & 15]
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int e = ((yy & 15) * 32 + 2 * ks + half) * 32 + c;
          bu[ks] = U[e] + osign * U[16 * 32 * 32 + e];
        }
        f32x16 acc = zero16();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = mfma(gireg[ks], bu[ks], acc);
        const int64_t rowpix = p.flip ? ((int64_t)n * p.Ho + (p.Ho - 1 - y0 - yy)) * p.Wo + (p.Wo - 1 - x0) : ((int64_t)n * p.Ho + y0 + yy) * p.Wo + x0;
        const int xsgn = p.flip ? -1 : 1;
        float* yrow = p.y + rowpix * p.ldy;
        float* arow = (!POST && p.act_out) ? p.act_out + rowpix * p.ld_act : nullptr;
        const float* rrow = RES ? p.res + rowpix * p.ld_res : nullptr;
        const float* grow = POST ? p.gact + rowpix * p.ld_gact : nullptr;
        float* y2row = (POST && p.y2) ? p.y2 + rowpix * p.ld_y2 : nullptr;
        // the per-pixel offsets below are invariant across rows and items: left alone, the compiler hoists all 48 of them out of the
        // persistent loop and keeps them in registers (256 VGPRs + spills); an opaque channel index makes it recompute them (one v_mad each)
        unsigned chv = (unsigned)chan;
        asm volatile("" : "+v"(chv));
        if (cok) {
          // the row's residual values first, as one burst of unconditional loads (columns beyond vx read the row's first pixel): a load placed
          // between the stores is waited for with vmcnt(0) each - sixteen exposed memory latencies per row (measured: + 1.2 ms per convolution
          // of 8 x 1024^2 x 32 with a residual, for 0.25 ms worth of extra traffic)
#pragma unroll
          for (int r0 = 0; r0 < 16; r0 += BURST) {                   // bursts of eight (sixteen values at once do not fit the register file; POST: of four)
            float rv[BURST], gv[BURST];
            if (RES) {
#pragma unroll
              for (int r = 0; r < BURST; ++r) {
                const int xx = acc_row(r0 + r, half);
                rv[r] = rrow[xx < vx ? (int)((xsgn * xx + subx) * p.ld_res) + (int)chv : (int)chv];
              }
            }
            if (POST) {                                              // the producing layer's activation output, same burst form
#pragma unroll
              for (int r = 0; r < BURST; ++r) {
                const int xx = acc_row(r0 + r, half);
                gv[r] = grow[xx < vx ? (int)((xsgn * xx + subx) * p.ld_gact) + (int)chv : (int)chv];
              }
            }
#pragma unroll
            for (int r = 0; r < BURST; ++r) {
              const int xx = acc_row(r0 + r, half);
              if (xx < vx) {
                float v = acc[r0 + r];
                const int xo = xsgn * xx + subx;
                if (!POST) {                                           // (a data-gradient launch has no bias, activation, BN or act_out)
                  v += bias;
                  v = TANH ? tanhf(v) : (v > 0.f ? v : v * p.alpha);
                  if (arow) arow[(int)(xo * p.ld_act) + (int)chv] = v;
                  v = v * sc + sh;
                }
                if (RES) v += rv[r];
                if (POST) {
                  if (y2row) y2row[(int)(xo * p.ld_y2) + (int)chv] = v;
                  const float g = gv[r];
                  v *= p.gmode == PCNN_ACT_TANH ? 1.f - g * g : (g > 0.f ? 1.f : p.galpha);
                  bsum += v;
                }
                yrow[(int)(xo * p.ldy) + (int)chv] = v;
                ymax = fmaxf(ymax, fabsf(v));
              }
            }
          }
        }
      }
    }
    if (next >= total) break;
    item = next;
    lds_barrier();                                                       // U is free for the next item
  }
  if (POST && p.bsum) p.bsum[(blockIdx.x * 8 + wave) * 64 + lane] += bsum;    // own slot: launches of one call follow each other on the stream
  if (p.absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if (lane == 0) {
      const unsigned bits = __float_as_uint(ymax <= 3.0e38f ? ymax : 3.0e38f);
      if (bits > __atomic_load_n(p.absmax, __ATOMIC_RELAXED)) atomicMax(p.absmax, bits);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ per-frequency channel mixing
struct MixParams { const float* xs; float* ys; const float* wsp; const int4* slots; int ntile, gin, gout, rows, Cin, cpt; };   // wsp: filter spectrum; Cin: its input channels; cpt: lanes per packed tile      // rows: spectrum rows per item (T*T)

// Y^ = X^ H per frequency, H = conj(W^) (correlation), as THREE real products (Gauss) instead of the four of [Yr | Yi] = [Xr | Xi] [[Hr, Hi], [-Hi, Hr]]:
//   k1 = (Xr + Xi) Hr,  k2 = Xr (Hi - Hr),  k3 = Xi (Hr + Hi);   Yr = k1 - k3,  Yi = k1 + k2
// - 48 instead of 64 MFMAs per M-tile and channel group, three accumulators instead of two, 48 instead of 64 filter registers per group.  Rounds 4 / 5 measured
// the kernel at 0.62 of HBM with the matrix pipe 0.52 busy and priced this change with diagnostic builds before building it (profiles/r05_study_mix_keep.txt,
// r05_study_mix_layers_upper_bounds.txt: 746 -> 657 us per launch at 7 taps with a quarter of the MFMAs gone, 603 with the real and imaginary row of a
// frequency adjacent as well - the FFT family's row order since round 6, spectral_common.h).  The sums Xr + Xi and the differences of the filter spectrum
// are formed in fp32: the result differs from the four-product form by rounding only (tests/test_gpu_spectral*.py at unchanged tolerances).
// A slot that packs two REAL frequencies (sl.z == 1; rows rr and ri hold two independent real spectra): Y(rr) = X(rr) Wr, Y(ri) = X(ri) Wi - the same three
// accumulators with A1 = Xr, A2 = Xi, B3 = 0 and Yi = k2 alone (a uniform factor s = 0 in place of 1).
// K order: step j pairs channel j (lanes 0-31) with channel 16 + j (lanes 32-63) - a lane's A operands are 16 CONSECUTIVE channels of its tile's rows.
// cpt < 32 (tile packing): lane = cpt * tile + channel on both sides and the filter block is block diagonal - a tile's channels mix only among themselves.
// (3 workgroups per CU - 168 VGPRs - was measured: 7 % slower; the kernel is bound by the HBM read + write stream, not by latency)
template <int GIN>
__global__ __launch_bounds__(256, 2) void spec_mix_kernel(MixParams p) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int slot = blockIdx.x, go = blockIdx.z;
  const int4 sl = p.slots[slot];
  const int rr = sl.x, ri = sl.y;
  int kind = sl.z;
  asm volatile("" : "+v"(kind));                               // per-lane on purpose: a uniform selector becomes scalar branches around the MFMA loop (two copies)
  const bool real2 = kind == 1;
  const float s_cplx = real2 ? 0.f : 1.f;
  // straight from the filter spectrum Wsp[ci * gout + go][row][co]: this lane's 16 input channels j + 16 half of its output channel c - 32 loads of
  // L2-resident rows
  float b1[GIN][16], b2[GIN][16], b3[GIN][16];
  {
    const int co = c % p.cpt;
#pragma unroll
    for (int gi = 0; gi < GIN; ++gi)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kc = j + 16 * half, ci = gi * 32 + kc % p.cpt;
        float wr = 0.f, wi = 0.f;
        if (ci < p.Cin && kc / p.cpt == c / p.cpt) {
          const float* wg = p.wsp + pcnn_spec::sp_item(ci * p.gout + go, p.rows) + co;
          wr = wg[rr * RS]; wi = wg[ri * RS];
        }
        // Hr = wr, Hi = -wi
        b1[gi][j] = wr;
        b2[gi][j] = real2 ? wi : -wi - wr;                 // Hi - Hr
        b3[gi][j] = real2 ? 0.f : wi - wr;                 // -(Hr + Hi): Yr = k1 + Xi b3
      }
  }
  const int nMt = (p.ntile + 31) >> 5;
  const int mstride = gridDim.y * 4;
  auto load_a = [&](int mt, int gi, f32x4 (&a)[2][4]) {
    const int tile = min(mt * 32 + c, p.ntile - 1);
    const float* base = p.xs + pcnn_spec::sp_item((int64_t)tile * GIN + gi, p.rows) + 16 * half;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      a[0][j4] = NT_LOAD(2, reinterpret_cast<const f32x4*>(base + rr * RS + 4 * j4));
      a[1][j4] = NT_LOAD(2, reinterpret_cast<const f32x4*>(base + ri * RS + 4 * j4));
    }
  };
  f32x4 a[2][4], an[2][4];
  int mt = blockIdx.y * 4 + wave;
  if (mt >= nMt) return;
  load_a(mt, 0, a);
  auto m_tile = [&](int mt) {
    f32x16 acc[3] = {zero16(), zero16(), zero16()};
#pragma unroll
    for (int gi = 0; gi < GIN; ++gi) {
      // the next operand block (next channel group, or the next M-tile's first group) is requested before this one's MFMAs (two blocks
      // ahead and 16-byte stores through an in-quad transpose were measured: no gain - the kernel waits on neither)
      if (gi + 1 < GIN) load_a(mt, gi + 1, an);
      else if (mt + mstride < nMt) load_a(mt + mstride, 0, an);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float xr = a[0][j >> 2][j & 3], xi = a[1][j >> 2][j & 3];
        acc[0] = mfma(fmaf(s_cplx, xi, xr), b1[gi][j], acc[0]);
        acc[1] = mfma(real2 ? xi : xr, b2[gi][j], acc[1]);
        acc[2] = mfma(xi, b3[gi][j], acc[2]);
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) a[pp][j4] = an[pp][j4];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = mt * 32 + acc_row(r, half);
      float* o = &p.ys[pcnn_spec::sp_item((int64_t)trow * p.gout + go, p.rows) + c];              // rows >= ntile: padding of the buffer (pad32)
      NT_STORE(4, acc[0][r] + acc[2][r], o + rr * RS);
      NT_STORE(4, fmaf(s_cplx, acc[0][r], acc[1][r]), o + ri * RS);
    }
  };
  // first M-tile outside the loop: every pass inside it then starts from the same memory-counter state (operand loads, then the previous
  // tile's stores), so the waits for its operands do not have to cover a first pass that has no stores pending
  m_tile(mt);
  for (mt += mstride; mt < nMt; mt += mstride) m_tile(mt);
}

// ------------------------------------------------------------------------------------------------------------------ weight gradient
struct WMixParams { const float* xs; const float* ds; float* part; const int4* slots; int ntile, gin, S, accumulate, rows, nslot; };

// The weight-gradient spectrum C^[f] = sum over tiles of conj(X^) D^ (or its conjugate), per frequency a complex product summed over the tiles -
// as THREE real products (Gauss; round 6, like the mixing kernels) instead of the four quadrants of [Xr | Xi]^T [Dr | Di]:
//   Q1 = (Xr - Xi)^T Dr,  Q2 = Xr^T (Di - Dr),  Q3 = Xi^T (Dr + Di);   Re = Q1 + Q3 = Xr^T Dr + Xi^T Di,  Im = Q1 + Q2 = Xr^T Di - Xi^T Dr
// P[slot][gi][quadrant q = 0..2][ci][co] holds Q1, Q2, Q3 (the stride stays four quadrants).  A slot that packs two REAL frequencies (sl.z == 1) wants
// Xr^T Dr and Xi^T Di: the same three products with a factor s = 0 in place of 1 in front of Xi (Q1) and Dr (Q3).  The sums and differences are formed
// in fp32 on the operands: the result differs from the four-quadrant form by rounding only (wgrad tests at unchanged tolerances).
// One wave owns all three quadrants (each operand row is loaded once per tile pair) for its share of the tiles: K is split over gridDim.y workgroups x 4
// waves; the partial sums are combined in a fixed order by spec_wcombine_kernel.
__global__ __launch_bounds__(256, 2) void spec_wmix_kernel(WMixParams p) {
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = blockIdx.x, gi = blockIdx.z;
  const int part = blockIdx.y * 4 + wave, nparts = p.S * 4;
  const int4 sl = p.slots[slot];
  const int per = (((p.ntile + nparts - 1) / nparts) + 1) & ~1;
  const int t0 = part * per, t1 = min(t0 + per, p.ntile);
  const float* xr = p.xs + sl.x * RS, *xi = p.xs + sl.y * RS;
  const float* dr = p.ds + sl.x * RS, *di = p.ds + sl.y * RS;
  int kind = sl.z;
  asm volatile("" : "+v"(kind));                               // per-lane on purpose (a uniform selector becomes scalar branches around the MFMA loop)
  const float s_cplx = kind == 1 ? 0.f : 1.f;
  f32x16 acc[3] = {zero16(), zero16(), zero16()};
  // operands of 2 WU tiles per step, requested one step ahead (two register sets): the loads of step i + 1 fly under the 4 WU MFMAs of step i
  constexpr int WU = 4;
  float ar[WU], ai[WU], br[WU], bi[WU], nar[WU], nai[WU], nbr[WU], nbi[WU];
  auto load = [&](int tb, float (&a0)[WU], float (&a1)[WU], float (&b0)[WU], float (&b1)[WU]) {
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int tt = tb + 2 * u + half;
      const unsigned tc = (unsigned)(tt < t1 ? tt : t0);
      const unsigned ix = (unsigned)pcnn_spec::sp_item((int64_t)tc * p.gin + gi, p.rows) + c, id = (unsigned)pcnn_spec::sp_item(tc, p.rows) + c;
      a0[u] = NT_LOAD(16, &xr[ix]); a1[u] = NT_LOAD(16, &xi[ix]); b0[u] = NT_LOAD(16, &dr[id]); b1[u] = NT_LOAD(16, &di[id]);
    }
  };
  if (t0 < t1) load(t0, ar, ai, br, bi);
  for (int tb = t0; tb < t1; tb += 2 * WU) {
    if (tb + 2 * WU < t1) load(tb + 2 * WU, nar, nai, nbr, nbi);
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const bool ok = tb + 2 * u + half < t1;
      const float a0 = ok ? ar[u] : 0.f, a1 = ok ? ai[u] : 0.f;
      acc[0] = mfma(fmaf(-s_cplx, a1, a0), br[u], acc[0]);
      acc[1] = mfma(a0, bi[u] - br[u], acc[1]);
      acc[2] = mfma(a1, fmaf(s_cplx, br[u], bi[u]), acc[2]);
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) { ar[u] = nar[u]; ai[u] = nai[u]; br[u] = nbr[u]; bi[u] = nbi[u]; }
  }
  float* o = p.part + (((int64_t)part * p.nslot + slot) * p.gin + gi) * 4 * 1024 + c;
#pragma unroll
  for (int qd = 0; qd < 3; ++qd) {
    float old[16];                                                   // accumulate: the quadrant's old values as one burst (a load between two
    if (p.accumulate) {                                              // stores is waited for with vmcnt(0): 64 exposed latencies per workgroup)
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = o[qd * 1024 + acc_row(r, half) * 32];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[qd * 1024 + acc_row(r, half) * 32] = p.accumulate ? old[r] + acc[qd][r] : acc[qd][r];
  }
}

// ------------------------------------------------------------------------------------------------------------------ fused backward mixing
// Backward pass of one layer: the gradient spectrum D^ (dz's windows) feeds BOTH the data gradient (Y^ = D^ M_f, spec_mix_kernel) and the
// weight gradient (P_f = sum_tiles X^^T D^, spec_wmix_kernel).  As two kernels D^ crosses HBM twice (7 spectrum passes per layer); here one
// wave does both for its M-tiles of 32 tiles - D^ comes from HBM once (the second operand layout, lane = channel instead of lane = tile, is
// re-read from L2 within the same M-tile), 6 passes per layer.  One x / dx channel group per blockIdx.z, dz has one group (Cout <= 32).
// The partial sums P are laid out exactly as spec_wmix_kernel's (spec_wcombine_kernel reads them): part = blockIdx.y * 4 + wave.
struct MixWParams { const float* zs; const float* xs; float* ys; float* part; const float* wsp; const int4* slots; int ntile, gx, rows, nslot, Cz, cpt, accumulate; };

__global__ __launch_bounds__(256, 2) void spec_mixw_kernel(MixWParams p) {
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = blockIdx.x, g = blockIdx.z;
  const int part = blockIdx.y * 4 + wave;
  const int4 sl = p.slots[slot];
  const int rr = sl.x, ri = sl.y;
  // M_f = [[Hr, -Hi], [Hi, Hr]] as in spec_mix_kernel<1> (input group: dz's single one; output group g), but held as its 32 DISTINCT values per
  // lane - the sign and, for the two slots that pack two real frequencies (block diagonal: [[Hr, 0], [0, Hi]]), the zero blocks go onto the A
  // operand (x -1, x 0: exact) - 32 registers instead of 64, which is what lets the weight-gradient accumulators live beside them.
  float hr[16], hi[16];
  const bool real2 = sl.z == 1;
  const float s_neg = real2 ? 0.f : -1.f, s_pos = real2 ? 0.f : 1.f;
  int kind = sl.z;
  asm volatile("" : "+v"(kind));                               // per-lane copy for the weight-gradient products (selects, not branches)
  const float s_cplx = kind == 1 ? 0.f : 1.f;
  {
    const int co = c % p.cpt;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int kc = j + 16 * half, ci = kc % p.cpt;
      float wr = 0.f, wi = 0.f;
      if (ci < p.Cz && kc / p.cpt == c / p.cpt) {
        const float* wg = p.wsp + pcnn_spec::sp_item(ci * p.gx + g, p.rows) + co;
        wr = wg[rr * RS]; wi = wg[ri * RS];
      }
      hr[j] = wr; hi[j] = wi;
    }
  }
  const int nMt = (p.ntile + 31) >> 5;
  const int mstride = gridDim.y * 4;
  // Weight-gradient operands of K steps 4 b .. 4 b + 3 (tile pair 2 j + half per step): ONE 16-byte load per spectrum row and batch - lane
  // (quad cq, i) fetches channels 4 cq .. 4 cq + 3 of tile 8 b + 2 i + half, the in-quad transpose turns that into channel 4 cq + i of the four
  // tiles 8 b + 2 j + half (register j): the MFMA operand layout (lane = channel) from a quarter of the memory instructions.
  const float* xr = p.xs + rr * RS, *xi = p.xs + ri * RS;
  const float* dr = p.zs + rr * RS, *di = p.zs + ri * RS;
  const int qi = lane & 3, c4 = c & ~3;
  const bool odd = lane & 1, upper = lane & 2;
  auto load_w = [&](int mt, int b, f32x4 (&w)[4]) {
    const int tt = mt * 32 + 8 * b + 2 * qi + half;
    const unsigned tc = (unsigned)(tt < p.ntile ? tt : 0);
    const unsigned ix = (unsigned)pcnn_spec::sp_item((int64_t)tc * p.gx + g, p.rows) + c4, id = (unsigned)pcnn_spec::sp_item(tc, p.rows) + c4;
    w[0] = NT_LOAD(16, reinterpret_cast<const f32x4*>(&xr[ix])); w[1] = NT_LOAD(16, reinterpret_cast<const f32x4*>(&xi[ix]));
    w[2] = *reinterpret_cast<const f32x4*>(&dr[id]); w[3] = *reinterpret_cast<const f32x4*>(&di[id]);
  };
  f32x16 accp[3] = {zero16(), zero16(), zero16()};                 // Q1, Q2, Q3 of spec_wmix_kernel (three real products per tile pair)
  auto mfma_w = [&](int mt, int b, f32x4 (&w)[4]) {
    float t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) t[q][j] = w[q][j];
      quad_transpose(t[q][0], t[q][1], t[q][2], t[q][3], odd, upper);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = mt * 32 + 8 * b + 2 * j + half < p.ntile;
      const float a0 = ok ? t[0][j] : 0.f, a1 = ok ? t[1][j] : 0.f;
      accp[0] = mfma(fmaf(-s_cplx, a1, a0), t[2][j], accp[0]);
      accp[1] = mfma(a0, t[3][j] - t[2][j], accp[1]);
      accp[2] = mfma(a1, fmaf(s_cplx, t[2][j], t[3][j]), accp[2]);
    }
  };
  // One M-tile = 8 phases of 16 MFMAs, weight-gradient batches and mixing quarters interleaved (W0 M0 W1 M1 W2 M2 W3 M3) so that every operand
  // is requested >= 3 phases (two register sets ws[0], ws[1]) resp. 4 phases (the mixing operand) before its use: a load issued one phase
  // ahead is still in flight when its MFMAs come up (first version of this kernel: 30 % slower than the two kernels it replaces).
  // The mixing operand (lane = tile: 16 consecutive channels of its tile's real and imaginary row) lives in two slots of 8 channels that
  // alternate between the real and the imaginary row: a quarter's registers are refilled, right after it has consumed them, with the piece that
  // is needed 4 phases later - 16 registers instead of 32.
  f32x4 aq[2][2], ws[2][4];
  auto load_a_part = [&](int mt, int pp, int sl2) {                  // channels 8 sl2 .. 8 sl2 + 7 (of this lane's 16) of row pp (0 real, 1 imaginary)
    const int tile = min(mt * 32 + c, p.ntile - 1);
    const float* base = p.zs + pcnn_spec::sp_item(tile, p.rows) + 16 * half + (pp ? ri : rr) * RS + 8 * sl2;
    aq[sl2][0] = *reinterpret_cast<const f32x4*>(base);
    aq[sl2][1] = *reinterpret_cast<const f32x4*>(base + 4);
  };
  int mt = part;
  if (mt < nMt) {
    load_w(mt, 0, ws[0]); load_w(mt, 1, ws[1]);
    load_a_part(mt, 0, 0); load_a_part(mt, 0, 1);
  }
  auto m_tile = [&](int mt) {
    const int mn = mt + mstride < nMt ? mt + mstride : mt;
    f32x16 acc[2] = {zero16(), zero16()};
    auto mix_quarter = [&](int qu) {                                 // K steps 8 qu .. 8 qu + 7 of the mixing product (qu < 2: real rows, else imaginary)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 8 * (qu & 1) + e;
        const float av = aq[qu & 1][e >> 2][e & 3];
        if (qu < 2) {
          acc[0] = mfma(av, hr[j], acc[0]);
          acc[1] = mfma(av * s_neg, hi[j], acc[1]);
        } else {
          acc[0] = mfma(av * s_pos, hi[j], acc[0]);
          acc[1] = mfma(av, real2 ? hi[j] : hr[j], acc[1]);
        }
      }
    };
    // (the last M-tile re-requests its own operands instead of branching around the loads: with a static number of memory operations per
    // pass the compiler's s_waitcnt values name exactly the load a phase needs; behind a branch they become vmcnt(0), i.e. a wait for the
    // previous M-tile's 32 stores.  The scheduling barriers keep the phases in the order written.)
    mfma_w(mt, 0, ws[0]); load_w(mt, 2, ws[0]);     __builtin_amdgcn_sched_barrier(0);
    mix_quarter(0);       load_a_part(mt, 1, 0);    __builtin_amdgcn_sched_barrier(0);
    mfma_w(mt, 1, ws[1]); load_w(mt, 3, ws[1]);     __builtin_amdgcn_sched_barrier(0);
    mix_quarter(1);       load_a_part(mt, 1, 1);    __builtin_amdgcn_sched_barrier(0);
    mfma_w(mt, 2, ws[0]); load_w(mn, 0, ws[0]);     __builtin_amdgcn_sched_barrier(0);
    mix_quarter(2);       load_a_part(mn, 0, 0);    __builtin_amdgcn_sched_barrier(0);
    mfma_w(mt, 3, ws[1]); load_w(mn, 1, ws[1]);     __builtin_amdgcn_sched_barrier(0);
    mix_quarter(3);       load_a_part(mn, 0, 1);    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int row = nt ? ri : rr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = mt * 32 + acc_row(r, half);
        NT_STORE(4, acc[nt][r], &p.ys[pcnn_spec::sp_item((int64_t)trow * p.gx + g, p.rows) + row * RS + c]);            // rows >= ntile: padding of the buffer (pad32)
      }
    }
  };
  // first M-tile outside the loop, as in spec_mix_kernel: a pass inside the loop then always starts with the previous pass's stores behind its
  // operand loads in the memory counter, and its waits need not also cover an entry with no stores pending (which makes them wait for the stores)
  if (mt < nMt) {
    m_tile(mt);
    for (mt += mstride; mt < nMt; mt += mstride) m_tile(mt);
  }
  float* o = p.part + (((int64_t)part * p.nslot + slot) * p.gx + g) * 4 * 1024 + c;
#pragma unroll
  for (int qd = 0; qd < 3; ++qd) {
    float old[16];
    if (p.accumulate) {
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = o[qd * 1024 + acc_row(r, half) * 32];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[qd * 1024 + acc_row(r, half) * 32] = p.accumulate ? old[r] + accp[qd][r] : accp[qd][r];
  }
}

// C^[f] = X^ conj(D^) from the three Gauss sums of spec_wmix_kernel / spec_mixw_kernel: Cr = Q1 + Q3, Ci = -(Q1 + Q2) (isign = -1: its conjugate);
// packed real slots: C(row rr) = Q1 = Xr^T Dr, C(row ri) = Q3 = Xi^T Di.
// Output: spectrum of a one-tile image with Cin*Cout channels, group = ci, lane = co.
// cpt < 32 (tile packing): the wanted products are the 32 / cpt diagonal blocks (tile with itself); they are summed here.
// One workgroup per (slot, channel group): no run-time divisions, every partial-sum read a whole 128-byte row.
__global__ __launch_bounds__(256) void spec_wcombine_kernel(const float* __restrict__ part, const int4* __restrict__ slots, float* __restrict__ csp, int S, int gin, int Cin, float isign, int cpt, int rows) {
  const int nslot = rows / 2;
  const int slot = blockIdx.x, gi = blockIdx.y;
  const int4 sl = slots[slot];
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int cil = e >> 5, co = e & 31, ci = gi * 32 + cil;
    if (ci >= Cin || cil >= cpt || co >= cpt) continue;
    float P[3] = {0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s)
      for (int sub = 0; sub < 32 / cpt; ++sub) {
        const float* b = part + (((int64_t)s * nslot + slot) * gin + gi) * 4 * 1024 + (sub * cpt + cil) * 32 + sub * cpt + co;
#pragma unroll
        for (int qd = 0; qd < 3; ++qd) P[qd] += b[qd * 1024];
      }
    // Q1 + Q3 = Xr^T Dr + Xi^T Di; Q1 + Q2 = Xr^T Di - Xi^T Dr
    const float cr = sl.z == 1 ? P[0] : P[0] + P[2], cim = sl.z == 1 ? P[2] : -isign * (P[0] + P[1]);     // isign = -1: conj(X^) D^ instead of X^ conj(D^)
    csp[pcnn_spec::sp_item(ci, rows) + sl.x * RS + co] = cr;
    csp[pcnn_spec::sp_item(ci, rows) + sl.y * RS + co] = cim;
  }
}

// ------------------------------------------------------------------------------------------------------------------ host side
void build_tables(std::vector<float>& tab, std::vector<int>& slots) {
  tab.assign(TAB_FLOATS, 0.f);
  const double tp = 2.0 * M_PI / T;
  auto G = [&](int s, int x) { return s <= 16 ? cos(tp * s * x) : -sin(tp * (s - 16) * x); };
  auto Gi = [&](int n, int s) {
    if (s == 0) return 1.0 / T;
    if (s == 16) return ((n & 1) ? -1.0 : 1.0) / T;
    return s < 16 ? 2.0 * cos(tp * s * n) / T : -2.0 * sin(tp * (s - 16) * n) / T;
  };
  // one radix-2 step of the 32-point complex DFT along y.  Forward (decimation in frequency): Z[2m + par] = sum_{y<16} (u[y] +- u[y+16])
  // e^{-i th}, th = 2 pi (2m + par) y / 32, as a real product with rows (part_out, m) and K (part_in, y): [[cos, sin], [-sin, cos]].
  // Inverse (decimation in time): E / O[y] = sum_m Z[2m + par] e^{+i th} / 32, u[y] = E + O, u[y + 16] = E - O.
  auto F2 = [&](int par, int row, int k) {
    const int po = row >> 4, m = row & 15, pi = k >> 4, y = k & 15;
    const double th = tp * (((2 * m + par) * y) & 31);
    return po == pi ? cos(th) : (po == 0 ? sin(th) : -sin(th));
  };
  auto FI2 = [&](int par, int row, int k) {
    const int po = row >> 4, y = row & 15, pi = k >> 4, m = k & 15;
    const double th = tp * (((2 * m + par) * y) & 31);
    return (po == pi ? cos(th) : (po == 0 ? -sin(th) : sin(th))) / T;
  };
  for (int k = 0; k < 32; ++k)
    for (int m = 0; m < 32; ++m) {
      tab[TAB_G + k * 32 + m] = (float)G(m, k); tab[TAB_GI + k * 32 + m] = (float)Gi(m, k);
      for (int par = 0; par < 2; ++par) {
        tab[TAB_F2 + par * 1024 + k * 32 + m] = (float)F2(par, m, k);
        tab[TAB_FI2 + par * 1024 + k * 32 + m] = (float)FI2(par, m, k);
      }
      // real columns of the inverse: row (par, y < 16) gathers the half-complex entries k whose frequency has parity par
      const int fk = k <= 16 ? k : k - 16;
      tab[TAB_GI2 + k * 32 + m] = (fk & 1) == (m >> 4) ? (float)Gi(m & 15, k) : 0.f;
    }
  slots.clear();
  for (int b = 0; b < 2; ++b) {
    const int base = b ? 32 : 0;
    slots.insert(slots.end(), {base, base + 16, 1, 0});
    for (int fy = 1; fy < 16; ++fy) slots.insert(slots.end(), {base + fy, base + 16 + fy, 0, 0});
  }
  for (int fx = 1; fx < 16; ++fx)
    for (int fy = 0; fy < 32; ++fy) slots.insert(slots.end(), {64 + 64 * (fx - 1) + fy, 64 + 64 * (fx - 1) + 32 + fy, 0, 0});
}

// The transform geometry of one call: tile size, spectrum rows per item, mixing slots, and the device tables of that size.
struct Geom { int T, rows, nslot; const float* tab; const int4* slots; };

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// the channel-mixing kernel writes whole 32-tile M-tiles (unconditional stores: their number per pass is then static): its output spectra are
// allocated for a multiple of 32 tiles; the rows of tiles that do not exist are never read
int pad32(int tiles) { return (tiles + 31) & ~31; }

int chunk_tiles(int Tg) {
  // tiles per launch of the transform / mixing kernels: 32 768 covers a whole 8 x 1024^2 layer (26 k tiles at 15 taps) - 4 GB of spectrum per
  // buffer, which 288 GB of HBM can afford; fewer, longer launches = fewer tails (8 192: + 1.8 % on the train step, 4 096: + 5.8 %).
  // A 64-point tile holds four times the rows: a quarter of the tiles is the same number of bytes.
  static const int v = getenv("PCNN_SPEC_CHUNK") ? atoi(getenv("PCNN_SPEC_CHUNK")) : 32768;
  const int t = Tg == 64 ? v / 4 : v;
  return t < 32 ? 32 : t;
}
// partial sums of the weight-gradient GEMM: S / 4 workgroups x 4 waves per (slot, channel group).  64-point tiles have 2 048 slots - one workgroup
// per slot already fills the chip four times over, and half the partial sums (128 instead of 256 MB written and read back per layer) are worth
// 0.06-0.08 ms per 13- / 15-tap layer; at 32 points (512 slots) 8 is the measured optimum (4: +0.05 ms at 9 taps, 12 / 16: +0.04 ... 0.08 ms).
// PCNN_WSPLIT32 / PCNN_WSPLIT64 are developer switches (A/B timing): the kernels index their partial sums as blockIdx.y * 4 + wave, so a value is
// rounded down to a positive multiple of 4 (at most 64); anything else - zero, negative, not a number - keeps the default.
int wsplit_env(const char* name, int dflt) {
  const char* e = getenv(name);
  if (!e) return dflt;
  const int v = atoi(e) & ~3;
  return (v >= 4 && v <= 64) ? v : dflt;
}
int wgrad_splits(int Tg = 32) {
  static const int s32 = wsplit_env("PCNN_WSPLIT32", 8), s64 = wsplit_env("PCNN_WSPLIT64", 4);
  return Tg == 64 ? s64 : s32;
}

// workspace header: the constant tables of both tile sizes; the per-call regions follow
// (two slot tables per tile size: the canonical row order of the matrix-core transform family, and the FFT family's interleaved order - spectral_common.h)
constexpr size_t O_TAB32 = 0, O_SLOTS32 = O_TAB32 + ((TAB_FLOATS * 4 + 255) & ~255), O_TAB64 = O_SLOTS32 + NSLOT * 16,
                 O_SLOTS64 = O_TAB64 + TAB64_FLOATS * 4, O_SLOTS32F = O_SLOTS64 + 2048 * 16, O_SLOTS64F = O_SLOTS32F + NSLOT * 16, O_REST = O_SLOTS64F + 2048 * 16;
static_assert(O_REST % 256 == 0, "workspace header alignment");

// grows the handle's workspace (never beyond the caller's limit, pcnn_set_workspace_limit); the tables are (re)uploaded after every growth
int ensure_workspace(pcnn_handle h, size_t bytes_after_tables, char** rest) {
  const size_t need = O_REST + bytes_after_tables;
  if (h->spec_ws_limit && need > h->spec_ws_limit)
    PCNN_FAIL(h, "spectral convolution: %zu B of workspace needed, the caller allows %zu B (pcnn_set_workspace_limit)", need, h->spec_ws_limit);
  if (h->spec_ws_bytes < need) {
    if (h->spec_ws) { pcnn_release(h, h->spec_ws); h->spec_ws = nullptr; h->spec_ws_bytes = 0; }
    size_t cap = need + need / 8;
    if (h->spec_ws_limit && cap > h->spec_ws_limit) cap = h->spec_ws_limit;
    if (hipMalloc(&h->spec_ws, cap) != hipSuccess) PCNN_FAIL(h, "spectral convolution: cannot allocate %zu B of workspace", cap);
    h->spec_ws_bytes = cap;
    static std::vector<float> tab, tab64; static std::vector<int> slots, slots64, slots32f, slots64f;   // static: the async copies below read them after this call returns
    if (tab.empty()) {
      build_tables(tab, slots);
      tab64.resize(TAB64_FLOATS); slots64.resize(2048 * 4);
      build_tables64(tab64.data(), slots64.data());
      slots32f.resize(NSLOT * 4); slots64f.resize(2048 * 4);
      pcnn_spec::sp_build_slots(32, PCNN_SP_P, slots32f.data());
      pcnn_spec::sp_build_slots(64, PCNN_SP_P, slots64f.data());
    }
    char* b = static_cast<char*>(h->spec_ws);
    if (hipMemcpyAsync(b + O_TAB32, tab.data(), TAB_FLOATS * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(b + O_SLOTS32, slots.data(), NSLOT * 16, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(b + O_TAB64, tab64.data(), TAB64_FLOATS * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(b + O_SLOTS64, slots64.data(), 2048 * 16, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(b + O_SLOTS32F, slots32f.data(), NSLOT * 16, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(b + O_SLOTS64F, slots64f.data(), 2048 * 16, hipMemcpyHostToDevice, h->stream) != hipSuccess)
      PCNN_FAIL(h, "spectral convolution: table upload failed");
  }
  *rest = static_cast<char*>(h->spec_ws) + O_REST;
  return 0;
}

// under a caller's workspace limit the launches cover fewer tiles each (never fewer than 32): halve until the call's regions fit
template <typename F>
int fit_chunk(pcnn_handle h, int chunk, F bytes_for) {
  while (h->spec_ws_limit && chunk > 32 && O_REST + bytes_for(chunk) > h->spec_ws_limit) chunk = std::max(32, chunk / 2);
  return chunk;
}

Geom geom_of(pcnn_handle h, int Tg) {
  char* b = static_cast<char*>(h->spec_ws);
  const bool fft = h->spectral_xform == PCNN_XFORM_FFT;                 // the row order belongs to the transform family (spectral_common.h)
  if (Tg == 64) return Geom{64, 4096, 2048, reinterpret_cast<const float*>(b + O_TAB64), reinterpret_cast<const int4*>(b + (fft ? O_SLOTS64F : O_SLOTS64))};
  return Geom{32, ROWS, NSLOT, reinterpret_cast<const float*>(b + O_TAB32), reinterpret_cast<const int4*>(b + (fft ? O_SLOTS32F : O_SLOTS32))};
}

template <typename K>
void set_lds(K kernel, size_t bytes = LDS_U) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); }

// persistent kernels: one 8-wave workgroup per CU (128 KB of LDS each) walking the (tile, channel group) items
template <bool MASKED, bool FENCE>
void launch_fwd_t(pcnn_handle h, const FwdParams& p, int ntile) {
  set_lds(spec_fwd_kernel<MASKED, FENCE>);
  hipLaunchKernelGGL((spec_fwd_kernel<MASKED, FENCE>), dim3((unsigned)std::min(ntile * p.groups, 256)), dim3(512), LDS_U, h->stream, p);
}
void launch_fwd(pcnn_handle h, const Geom& gm, FwdParams p, int ntile) {
  static const int fence = getenv("PCNN_SPEC_FENCE") ? atoi(getenv("PCNN_SPEC_FENCE")) : 0;
  p.ntile = ntile; p.tab = gm.tab;
  if (gm.T == 64) { if (h->spectral_xform == PCNN_XFORM_FFT) launch_fwd_fft64(h, p, ntile); else launch_fwd64(h, p, ntile); return; }
  if (h->spectral_xform == PCNN_XFORM_FFT) { launch_fwd_fft32(h, p, ntile); return; }
  const bool masked = p.ylim < T || p.xlim < T;
  if (masked) { if (fence) launch_fwd_t<true, true>(h, p, ntile); else launch_fwd_t<true, false>(h, p, ntile); }
  else { if (fence) launch_fwd_t<false, true>(h, p, ntile); else launch_fwd_t<false, false>(h, p, ntile); }
}
template <bool TANH, bool RES>
void launch_inv_t(pcnn_handle h, const InvParams& p, const dim3& grid) {
  set_lds(spec_inv_kernel<TANH, RES>, LDS_INV);
  hipLaunchKernelGGL((spec_inv_kernel<TANH, RES>), grid, dim3(512), LDS_INV, h->stream, p);
}
void launch_inv(pcnn_handle h, const Geom& gm, InvParams p, int ntile) {
  p.ntile = ntile; p.tab = gm.tab;
  if (gm.T == 64) { if (h->spectral_xform == PCNN_XFORM_FFT) launch_inv_fft64(h, p, ntile); else launch_inv64(h, p, ntile); return; }
  if (h->spectral_xform == PCNN_XFORM_FFT) { launch_inv_fft32(h, p, ntile); return; }
  const dim3 grid((unsigned)std::min(ntile * p.groups, 256));
  if (p.gact) {                                                      // data gradient + the producer's activation backward (linear conv epilogue)
    p.alpha = 1.f;
    p.galpha = p.gmode == PCNN_ACT_LINEAR ? 1.f : (p.gmode == PCNN_ACT_RELU ? 0.f : p.galpha);
    set_lds(spec_inv_kernel<false, true, true>, LDS_INV); set_lds(spec_inv_kernel<false, false, true>, LDS_INV);
    if (p.res) hipLaunchKernelGGL((spec_inv_kernel<false, true, true>), grid, dim3(512), LDS_INV, h->stream, p);
    else hipLaunchKernelGGL((spec_inv_kernel<false, false, true>), grid, dim3(512), LDS_INV, h->stream, p);
    return;
  }
  if (p.act == PCNN_ACT_TANH) {
    if (p.res) launch_inv_t<true, true>(h, p, grid); else launch_inv_t<true, false>(h, p, grid);
  } else {
    p.alpha = p.act == PCNN_ACT_LINEAR ? 1.f : (p.act == PCNN_ACT_RELU ? 0.f : p.alpha);     // slope of the negative side
    if (p.res) launch_inv_t<false, true>(h, p, grid); else launch_inv_t<false, false>(h, p, grid);
  }
}

void launch_mix(pcnn_handle h, const Geom& gm, MixParams mx, int gin, int gout, int nt) {
  mx.slots = gm.slots; mx.rows = gm.rows; mx.ntile = nt;
  const int nMt = pcnn_cdiv(nt, 32);
  const int gy = std::max(1, std::min(pcnn_cdiv(nMt, 4), 3));
  if (gin == 1) hipLaunchKernelGGL(spec_mix_kernel<1>, dim3(gm.nslot, gy, gout), dim3(256), 0, h->stream, mx);
  else hipLaunchKernelGGL(spec_mix_kernel<2>, dim3(gm.nslot, gy, gout), dim3(256), 0, h->stream, mx);
}

// tiles per lane group for a layer of Cin -> Cout channels: both sides must fit `cpt` lanes (a power of two >= 4)
int pack_for(int Cin, int Cout) {
  static const int on = getenv("PCNN_SPEC_PACK") ? atoi(getenv("PCNN_SPEC_PACK")) : 1;
  const int c = std::max(Cin, Cout);
  if (!on || c > 16) return 1;
  return c <= 4 ? 8 : (c <= 8 ? 4 : 2);
}

// Tile size of a layer's spectral route.  64-point tiles halve the spectrum volume of the 13- and 15-tap layers ((64/50)^2 = 1.6 values per output
// pixel at 15 taps against (32/18)^2 = 3.2): the two mixing passes take 0.55 of their time, the transforms about the same (measured at
// 8 x 1024^2, tools/probe_tile64.py: 15 taps 4.4 -> 3.5 ms forward, 6.9 -> 5.6 ms fused backward; 13 taps 3.8 -> 3.6 / 6.0 -> 5.5).  At 11 taps
// and below their transforms cost more than the mixing saves (11 taps: 2.98 -> 3.25 ms), layers of <= 16 channels keep the tile-packed
// 32-point form, and the inverse kernel holds the rows of a tile's valid region in 7 accumulator sets per wave (56 rows: >= 9 taps).
// Decided on ONE image (>= 36 tiles of 64 points), like the route, so that a sample's arithmetic never depends on its batch neighbours.
int pick_tile(pcnn_handle h, const pcnn_conv_desc* d) {
  const int forced = h->spectral_tile;                     // pcnn_set_spectral_tile / environment PCNN_SPEC_T
  const bool can64 = d->kh >= 9 && d->kw >= 9 && d->kh <= 15 && d->kw <= 15 && pack_for(d->Cin, d->Cout) == 1;
  if (forced == 32 || !can64) return 32;
  if (forced == 64) return 64;
  const int Vy = 65 - d->kh, Vx = 65 - d->kw;
  const int tiles = pcnn_cdiv(d->Ho, Vy) * pcnn_cdiv(d->Wo, Vx);
  // 11 and 12 taps: ahead only on large images (1024^2, 16->32: 2.77 -> 2.66 ms forward, 4.48 -> 4.30 fused backward; 512^2: 0.80 -> 0.93) - with the
  // matrix-core transforms.  With the FFT transforms (round 5) the 32-point kernels gained more than the 64-point ones: 11 taps 2.16 vs 2.19 ms forward,
  // 3.44 vs 3.47 fused backward at 8 x 1024^2 (profiles/r05_probe_xform32.txt / _xform64.txt) - 32-point tiles; 13 / 15 taps stay at 64 (2.70 -> 2.45, 3.22 -> 2.72).
  if (d->kh >= 11 && d->kw >= 11 && d->kh < 13 && d->kw < 13) return (h->spectral_xform != PCNN_XFORM_FFT && tiles >= 256) ? 64 : 32;
  // 14 / 15 taps: 64-point tiles from 9 tiles per image on (round 6, tools/probe_tile_shipped.py at the shipped training shapes - batch 50, grids of 192..384 points,
  // profiles/r06_probe_tile_shipped.txt: 15 taps 32->32 forward 0.92 / 0.70 ms at 192^2 ... 2.96 / 2.32 at 384^2, fused backward 1.31 / 1.10 ... 4.92 / 3.67 -
  // with 32-point tiles a 15-tap layer moves 3.2 spectrum values per output pixel, with 64-point tiles 1.6, and that outweighs the ragged last tile row of a small
  // image); 13 taps break even below ~300 points per side and keep the 36-tile rule.
  if (d->kh >= 14 && d->kw >= 14 && tiles >= 9) return 64;
  return (d->kh >= 13 && d->kw >= 13 && tiles >= 36) ? 64 : 32;
}

// POST: bias gradient = the lanes' partial sums of spec_inv_kernel<.., POST>, added in a fixed order.  One workgroup per channel; thread t sums the
// slots (block, wave) = t, t + 256, ... of the lanes that carry the channel (lane & 31 = sub cpt + channel for each packed tile, both halves).
__global__ __launch_bounds__(256) void spec_post_bias_kernel(const float* __restrict__ bsum, int nslots, int pack, int cpt, float* __restrict__ dbias) {
  __shared__ float red[256];
  const int ch = blockIdx.x, t = threadIdx.x;
  float a = 0.f;
  for (int sl = t; sl < nslots; sl += 256)
    for (int sub = 0; sub < pack; ++sub) {
      const int c = sub * cpt + ch;
      a += bsum[sl * 64 + c];
      a += bsum[sl * 64 + 32 + c];
    }
  red[t] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (t < st) red[t] += red[t + st]; __syncthreads(); }
  if (t == 0) dbias[ch] = red[0];
}

}  // namespace

extern "C" int pcnn_set_workspace_limit(pcnn_handle h, size_t bytes) {
  if (!h) return 1;
  h->spec_ws_limit = bytes;
  if (bytes && h->spec_ws && h->spec_ws_bytes > bytes) {       // what the handle already holds beyond the new cap goes back to the caller's pool
    pcnn_release(h, h->spec_ws);
    h->spec_ws = nullptr; h->spec_ws_bytes = 0;
  }
  return 0;
}

// Route choice.  Both estimates are calibrated on MI355X measurements at 8 x 1024^2 (tools/probe_spectral.py, profiles/r02_probe_spectral.txt):
// the spectral route costs a fixed time per 32 x 32 tile whatever the filter size (0.21 us per tile with <= 32 channels, 0.33 us with
// 64; its three kernels are HBM-bound on the tile spectra), the direct route the layer's padded MAC count over the rate its kernel sustains.
// Cost-model estimates of one convolution on ONE image (seconds): the spectral route and its alternative.  Returns 0 when the shape cannot
// take the spectral route at all, 1 when the policy is "whenever the shape allows" (no estimates), 2 with both estimates filled in.
static int spectral_costs(pcnn_handle h, const pcnn_conv_desc* d, bool wgrad, double* t_spec, double* t_alt) {
  const int mode = h->spectral_mode;                       // 0: never, 1: whenever the shape allows, -1: cost model
  if (mode == 0) return 0;
  if (d->kh > 15 || d->kw > 15 || d->kh < 2 || d->kw < 2 || d->Cin > 64) return 0;
  if (d->Cout > 32 && (wgrad || d->Cout != 64)) return 0;          // 64 output channels: two exact lane groups (forward / data gradient only)
  if (mode == 1) return 1;
  // decided on ONE image, so that a sample's arithmetic never depends on its batch neighbours (the model's per-sample results are
  // bit-identical for any batch size); images of fewer than 4 tiles stay on the direct route (launch overheads dominate there)
  const int Vy = T - d->kh + 1, Vx = T - d->kw + 1;
  const int pack = pack_for(d->Cin, d->Cout);
  const double tiles = (double)((d->Ho + Vy - 1) / Vy) * (((d->Wo + Vx - 1) / Vx + pack - 1) / pack);        // tile groups
  if (tiles < 4) return 0;
  *t_spec = tiles * ((d->Cin > 32 || d->Cout > 32) ? 0.33e-6 : 0.19e-6);
  if (!wgrad && pcnn_conv_small_fwd_eligible(d)) {         // the alternative is the vector-ALU narrow kernel: ~17 T multiply-adds/s on its padded channels
    const double fma = (double)d->Ho * d->Wo * d->kh * d->kw * ((d->Cin + 3) & ~3) * ((d->Cout + 3) & ~3);
    *t_alt = 0.8 / 0.9 * fma / 17e12;                      // (compared below with the common factor 0.9)
    return 2;
  }
  const int cin8 = (d->Cin + 7) & ~7, co32 = (d->Cout + 31) & ~31;
  const double flop = 2.0 * d->Ho * d->Wo * d->kh * d->kw * cin8 * co32;
  const bool split = h->math_mode == PCNN_MATH_SPLIT_F16;
  const double rate = split ? (wgrad ? 210e12 : 330e12) : (wgrad ? 95e12 : 118e12);
  *t_alt = flop / rate;
  return 2;
}

bool pcnn_spectral_eligible(pcnn_handle h, const pcnn_conv_desc* d, bool wgrad) {
  double t_spec = 0.0, t_alt = 0.0;
  const int r = spectral_costs(h, d, wgrad, &t_spec, &t_alt);
  return r == 1 || (r == 2 && t_spec < 0.9 * t_alt);
}

// forward / data-gradient route of a layer the narrow kernels could also take: they keep every 3x3 layer; a 5x5 layer goes spectral when
// that is the faster of the two
bool pcnn_conv_fwd_takes_narrow_route(pcnn_handle h, const pcnn_conv_desc* d) {
  return pcnn_conv_small_fwd_eligible(d) && !(d->kh == 5 && pcnn_spectral_eligible(h, d, false));
}

namespace {
// the filter as a kh x kw one-tile "image" with Cin*Cout channels (group = ci [x output group], lane = co): its spectrum feeds the mixing matrices
FwdParams filter_params(const Geom& gm, const float* w, float* wsp, int kh, int kw, int Cin, int Cout, int gout) {
  FwdParams fw;
  fw.x = w; fw.sp = wsp; fw.tab = gm.tab; fw.H = kh; fw.W = kw; fw.C = Cin * Cout; fw.ld = Cin * Cout; fw.groups = Cin * gout;
  fw.cstride = gout > 1 ? 32 : Cout; fw.cvalid = gout > 1 ? 32 : Cout;     // group ci * gout + go holds output channels 32 go .. 32 go + 31
  fw.tiles_x = 1; fw.tiles_y = 1; fw.tile0 = 0; fw.ntile = 1; fw.Vy = gm.T; fw.Vx = gm.T; fw.oy = 0; fw.ox = 0;
  // the filter is a kh x kw corner of its tile: the masked form of the transform skips the zero rows / columns (same sums: only zero products are dropped)
  fw.pad_mode = PCNN_PAD_CONSTANT; fw.pad_value = 0.f; fw.ylim = kh; fw.xlim = kw; fw.ext_y = 1 << 30; fw.ext_x = 1 << 30;
  fw.pack = 1; fw.cpt = 32; fw.tgx = 1;
  return fw;
}
// the weight gradient's spectrum C^ (one tile, Cin*Cout channels) back to its kh x kw taps
InvParams taps_params(const Geom& gm, const float* csp, float* dw, int kh, int kw, int Cin, int Cout, int flip) {
  InvParams iv;
  iv.sp = csp; iv.tab = gm.tab; iv.y = dw; iv.bias = nullptr; iv.bn_scale = nullptr; iv.bn_shift = nullptr; iv.res = nullptr; iv.act_out = nullptr;
  iv.absmax = nullptr; iv.Ho = kh; iv.Wo = kw; iv.C = Cin * Cout; iv.ldy = Cin * Cout; iv.ld_res = 0; iv.ld_act = 0;
  iv.groups = Cin; iv.cstride = Cout; iv.cvalid = Cout; iv.act = PCNN_ACT_LINEAR; iv.alpha = 0.f;
  iv.tiles_x = 1; iv.tiles_y = 1; iv.tile0 = 0; iv.ntile = 1; iv.Vy = gm.T; iv.Vx = gm.T; iv.flip = flip; iv.pack = 1; iv.cpt = 32; iv.tgx = 1;
  return iv;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------ filter-spectrum cache
// A filter's spectrum depends on the weights alone, yet every convolution call recomputed it: ~140 launches of ~10 us (+ the gaps around them) per
// training step of hpnn.json, on the stream's critical path, and again on every inference call although the weights never change.
// With pcnn_set_filter_version(h, v != 0) the handle keeps the spectrum of every filter it has seen - key: pointer, shape, tile size, transform family -
// in buffers of its own, stamped with v.  A call that finds its filter stamped with the current version uses it as it is; the first call that finds a
// stale stamp refreshes EVERY filter of the handle in one launch per tile size (fft32 / fft64_fwd_multi_kernel: the table of parameter blocks lives in
// device memory and changes only when a filter is added), so a training step pays one or two launches instead of one per layer and direction.
// Nothing is allocated or uploaded while the stream is being captured: a filter first seen under capture is transformed into the workspace as before.
namespace {
struct FilterEntry { const float* w; int kh, kw, Cin, Cout, gout, T, xform; float* buf; unsigned long long version; FwdParams fp; };
struct FilterCache {
  std::vector<FilterEntry> e;
  std::vector<FwdParams> host[2];                                   // [0]: 32-point entries, [1]: 64-point entries (FFT family), in the order of the device table
  FwdParams* dev[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  size_t bytes = 0;
};
FilterCache* cache_of(pcnn_handle h) {
  if (!h->filter_cache) h->filter_cache = new FilterCache();
  return static_cast<FilterCache*>(h->filter_cache);
}
// every entry of the handle gets the current version: one table launch per tile size (FFT family), one launch per entry otherwise
void refresh_filters(pcnn_handle h, FilterCache* fc) {
  for (int ti = 0; ti < 2; ++ti) {
    if (fc->host[ti].empty()) continue;
    int max_items = 1;
    for (const FwdParams& f : fc->host[ti]) max_items = std::max(max_items, f.groups);
    if (ti == 0) launch_fwd_fft32_multi(h, fc->dev[0], (int)fc->host[0].size(), max_items);
    else launch_fwd_fft64_multi(h, fc->dev[1], (int)fc->host[1].size(), max_items);
  }
  for (FilterEntry& en : fc->e) {
    if (en.xform != PCNN_XFORM_FFT && en.version != h->filter_version) {
      const Geom gm = geom_of(h, en.T);
      en.fp.tab = gm.tab;                                           // the tables move when the workspace grows
      launch_fwd(h, gm, en.fp, 1);
    }
    en.version = h->filter_version;
  }
  ++h->fc_refreshes;
}
// the spectrum of filter `w` ((kh, kw, Cin, Cout), output channel groups gout) for this call: a cached buffer, or `ws_slot` freshly filled
const float* filter_spectrum(pcnn_handle h, const Geom& gm, const float* w, float* ws_slot, int kh, int kw, int Cin, int Cout, int gout) {
  auto uncached = [&]() { launch_fwd(h, gm, filter_params(gm, w, ws_slot, kh, kw, Cin, Cout, gout), 1); return ws_slot; };
  if (h->filter_version == 0) return uncached();
  FilterCache* fc = cache_of(h);
  for (FilterEntry& en : fc->e)
    if (en.w == w && en.kh == kh && en.kw == kw && en.Cin == Cin && en.Cout == Cout && en.gout == gout && en.T == gm.T && en.xform == h->spectral_xform) {
      if (en.version != h->filter_version) refresh_filters(h, fc); else ++h->fc_hits;
      return en.buf;
    }
  // first sight of this filter: a buffer of its own - unless the stream is being captured (no allocation, no upload inside a capture)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return uncached(); }
  const size_t bytes = align256(sp_bytes((size_t)Cin * gout, gm.rows));
  void* buf = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess) { (void)hipGetLastError(); return uncached(); }
  FilterEntry en{w, kh, kw, Cin, Cout, gout, gm.T, h->spectral_xform, static_cast<float*>(buf), h->filter_version,
                 filter_params(gm, w, static_cast<float*>(buf), kh, kw, Cin, Cout, gout)};
  en.fp.ntile = 1;
  if (en.xform == PCNN_XFORM_FFT) {
    const int ti = gm.T == 64 ? 1 : 0;
    const size_t n = fc->host[ti].size() + 1;
    if (n > fc->cap[ti]) {
      const size_t cap = std::max<size_t>(64, 2 * n);
      void* t = nullptr;
      if (hipMalloc(&t, cap * sizeof(FwdParams)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(buf); return uncached(); }   // no table: the filter stays uncached
      if (fc->dev[ti]) pcnn_release(h, fc->dev[ti]);
      fc->dev[ti] = static_cast<FwdParams*>(t); fc->cap[ti] = cap;
    }
    fc->host[ti].push_back(en.fp);
    // (synchronous copy of the whole table: once per filter, in the first step that sees it; the stream may still be reading the old contents)
    (void)hipStreamSynchronize(h->stream);
    (void)hipMemcpy(fc->dev[ti], fc->host[ti].data(), n * sizeof(FwdParams), hipMemcpyHostToDevice);
  }
  launch_fwd(h, gm, en.fp, 1);
  fc->e.push_back(en);
  fc->bytes += bytes;
  ++h->fc_fills;
  return en.buf;
}
}  // namespace

void pcnn_filter_cache_free(pcnn_handle_s* h) {
  if (!h || !h->filter_cache) return;
  FilterCache* fc = static_cast<FilterCache*>(h->filter_cache);
  for (FilterEntry& en : fc->e) pcnn_release(h, en.buf);
  for (int ti = 0; ti < 2; ++ti) if (fc->dev[ti]) pcnn_release(h, fc->dev[ti]);
  delete fc;
  h->filter_cache = nullptr;
}

extern "C" int pcnn_set_filter_version(pcnn_handle h, uint64_t version) {
  if (!h) return 1;
  h->filter_version = version;
  return 0;
}

extern "C" int pcnn_filter_cache_clear(pcnn_handle h) {
  if (!h) return 1;
  pcnn_filter_cache_free(h);
  return 0;
}

extern "C" int pcnn_filter_cache_stats(pcnn_handle h, long long* entries, long long* bytes, long long* hits, long long* fills, long long* refreshes) {
  if (!h) return 1;
  const FilterCache* fc = static_cast<const FilterCache*>(h->filter_cache);
  if (entries) *entries = fc ? (long long)fc->e.size() : 0;
  if (bytes) *bytes = fc ? (long long)fc->bytes : 0;
  if (hits) *hits = h->fc_hits;
  if (fills) *fills = h->fc_fills;
  if (refreshes) *refreshes = h->fc_refreshes;
  return 0;
}

int pcnn_spectral_conv_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                           const float* bn_shift, const float* residual, float* y, float* act_out) {
  const int Tg = pick_tile(h, d);
  const int Vy = Tg - d->kh + 1, Vx = Tg - d->kw + 1;
  const int tiles_y = pcnn_cdiv(d->Ho, Vy), tiles_x = pcnn_cdiv(d->Wo, Vx);
  const int pack = Tg == 64 ? 1 : pack_for(d->Cin, d->Cout), cpt = 32 / pack, tgx = pcnn_cdiv(tiles_x, pack);
  const int64_t ntile = (int64_t)d->N * tiles_y * tgx;                       // tile groups (= tiles when pack == 1)
  PCNN_REQUIRE(h, ntile < (1ll << 30), "spectral convolution: too many tiles");
  const int gin = pcnn_cdiv(d->Cin, 32), gout = pcnn_cdiv(d->Cout, 32);
  const int rows = Tg * Tg;
  // workspace: [filter spectrum | input spectra | output spectra]
  const size_t wsp_b = align256(sp_bytes((size_t)d->Cin * gout, rows));
  auto xs_bytes = [&](int ch) { return align256(sp_bytes((size_t)ch * gin, rows)); };
  auto ys_bytes = [&](int ch) { return align256(sp_bytes((size_t)pad32(ch) * gout, rows)); };
  const int chunk = fit_chunk(h, (int)std::min<int64_t>(chunk_tiles(Tg) / (gin > gout ? gin : gout), ntile), [&](int ch) { return wsp_b + xs_bytes(ch) + ys_bytes(ch); });
  const size_t xs_b = xs_bytes(chunk), ys_b = ys_bytes(chunk);
  char* r;
  if (int rc = ensure_workspace(h, wsp_b + xs_b + ys_b, &r)) return rc;
  const Geom gm = geom_of(h, Tg);
  float* wsp = reinterpret_cast<float*>(r); r += wsp_b;
  float* xs = reinterpret_cast<float*>(r); r += xs_b;
  float* ys = reinterpret_cast<float*>(r);
  const float* fsp = filter_spectrum(h, gm, w, wsp, d->kh, d->kw, d->Cin, d->Cout, gout);
  PCNN_CHECK_LAUNCH(h, "spectral convolution (filter spectrum)");
  FwdParams fx;
  fx.x = x; fx.sp = xs; fx.tab = gm.tab; fx.H = d->H; fx.W = d->W; fx.C = d->Cin; fx.ld = d->ldx; fx.groups = gin; fx.cstride = 32; fx.cvalid = 32;
  fx.tiles_x = tiles_x; fx.tiles_y = tiles_y; fx.Vy = Vy; fx.Vx = Vx; fx.oy = d->pad_top; fx.ox = d->pad_left; fx.pad_mode = d->pad_mode;
  fx.pad_value = d->pad_value; fx.ylim = Tg; fx.xlim = Tg; fx.ext_y = 1 << 30; fx.ext_x = 1 << 30;
  fx.pack = pack; fx.cpt = cpt; fx.tgx = tgx;
  if (pack > 1) { fx.cstride = cpt; fx.cvalid = cpt; }
  InvParams iv;
  iv.sp = ys; iv.tab = gm.tab; iv.y = y; iv.bias = bias; iv.bn_scale = bn_scale; iv.bn_shift = bn_shift; iv.res = residual; iv.act_out = act_out;
  iv.absmax = reinterpret_cast<unsigned*>(h->y_absmax);
  iv.Ho = d->Ho; iv.Wo = d->Wo; iv.C = d->Cout; iv.ldy = d->ldy; iv.ld_res = d->ld_res; iv.ld_act = d->ld_act_out; iv.groups = gout; iv.cstride = 32; iv.cvalid = 32;
  iv.act = d->act; iv.alpha = d->act_alpha; iv.tiles_x = tiles_x; iv.tiles_y = tiles_y; iv.Vy = Vy; iv.Vx = Vx; iv.flip = 0;
  iv.pack = pack; iv.cpt = cpt; iv.tgx = tgx;
  if (pack > 1) { iv.cstride = cpt; iv.cvalid = cpt; }
  MixParams mx;
  mx.xs = xs; mx.ys = ys; mx.wsp = fsp; mx.gin = gin; mx.gout = gout; mx.Cin = d->Cin; mx.cpt = cpt;
  for (int64_t t0 = 0; t0 < ntile; t0 += chunk) {
    const int nt = (int)std::min<int64_t>(chunk, ntile - t0);
    fx.tile0 = (int)t0; iv.tile0 = (int)t0;
    launch_fwd(h, gm, fx, nt);
    launch_mix(h, gm, mx, gin, gout, nt);
    launch_inv(h, gm, iv, nt);
  }
  PCNN_CHECK_LAUNCH(h, "spectral convolution");
  return 0;
}

int pcnn_spectral_conv_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw) {
  const int Tg = pick_tile(h, d);
  const int Vy = Tg - d->kh + 1, Vx = Tg - d->kw + 1;
  const int tiles_y = pcnn_cdiv(d->Ho, Vy), tiles_x = pcnn_cdiv(d->Wo, Vx);
  const int pack = Tg == 64 ? 1 : pack_for(d->Cin, d->Cout), cpt = 32 / pack, tgx = pcnn_cdiv(tiles_x, pack);
  const int64_t ntile = (int64_t)d->N * tiles_y * tgx;
  PCNN_REQUIRE(h, ntile < (1ll << 30), "spectral convolution: too many tiles");
  const int gin = pcnn_cdiv(d->Cin, 32), S = wgrad_splits(Tg);
  const int rows = Tg * Tg, nslot = rows / 2;
  // workspace: [C^ | x window spectra | dz tile spectra | partial sums]
  const size_t csp_b = align256(sp_bytes((size_t)d->Cin, rows)), part_b = align256((size_t)S * nslot * gin * 4 * 1024 * 4);
  auto xs_bytes = [&](int ch) { return align256(sp_bytes((size_t)ch * gin, rows)); };
  auto zs_bytes = [&](int ch) { return align256(sp_bytes((size_t)ch, rows)); };
  const int chunk = fit_chunk(h, (int)std::min<int64_t>(chunk_tiles(Tg) / gin, ntile), [&](int ch) { return csp_b + part_b + xs_bytes(ch) + zs_bytes(ch); });
  const size_t xs_b = xs_bytes(chunk), zs_b = zs_bytes(chunk);
  char* r;
  if (int rc = ensure_workspace(h, csp_b + xs_b + zs_b + part_b, &r)) return rc;
  const Geom gm = geom_of(h, Tg);
  float* csp = reinterpret_cast<float*>(r); r += csp_b;
  float* xs = reinterpret_cast<float*>(r); r += xs_b;
  float* zs = reinterpret_cast<float*>(r); r += zs_b;
  float* part = reinterpret_cast<float*>(r);
  FwdParams fx;
  fx.x = x; fx.sp = xs; fx.tab = gm.tab; fx.H = d->H; fx.W = d->W; fx.C = d->Cin; fx.ld = d->ldx; fx.groups = gin; fx.cstride = 32; fx.cvalid = 32;
  fx.tiles_x = tiles_x; fx.tiles_y = tiles_y; fx.Vy = Vy; fx.Vx = Vx; fx.oy = d->pad_top; fx.ox = d->pad_left; fx.pad_mode = d->pad_mode;
  fx.pad_value = d->pad_value; fx.ylim = Tg; fx.xlim = Tg; fx.ext_y = 1 << 30; fx.ext_x = 1 << 30;
  fx.pack = pack; fx.cpt = cpt; fx.tgx = tgx;
  if (pack > 1) { fx.cstride = cpt; fx.cvalid = cpt; }
  FwdParams fz = fx;                                    // dz: the tile's own Vy x Vx outputs, zero elsewhere in the window
  fz.x = dz; fz.sp = zs; fz.H = d->Ho; fz.W = d->Wo; fz.C = d->Cout; fz.ld = d->ldy; fz.groups = 1; fz.oy = 0; fz.ox = 0;
  fz.pad_mode = PCNN_PAD_CONSTANT; fz.pad_value = 0.f; fz.ylim = Vy; fz.xlim = Vx;
  WMixParams wm;
  wm.xs = xs; wm.ds = zs; wm.part = part; wm.slots = gm.slots; wm.gin = gin; wm.S = S / 4; wm.rows = rows; wm.nslot = nslot;
  for (int64_t t0 = 0; t0 < ntile; t0 += chunk) {
    const int nt = (int)std::min<int64_t>(chunk, ntile - t0);
    fx.tile0 = (int)t0; fz.tile0 = (int)t0; wm.ntile = nt; wm.accumulate = t0 > 0;
    launch_fwd(h, gm, fx, nt);
    launch_fwd(h, gm, fz, nt);
    hipLaunchKernelGGL(spec_wmix_kernel, dim3(nslot, S / 4, gin), dim3(256), 0, h->stream, wm);
  }
  hipLaunchKernelGGL(spec_wcombine_kernel, dim3(nslot, gin), dim3(256), 0, h->stream, part, gm.slots, csp, S, gin, d->Cin, 1.0f, cpt, rows);
  launch_inv(h, gm, taps_params(gm, csp, dw, d->kh, d->kw, d->Cin, d->Cout, 0), 1);
  PCNN_CHECK_LAUNCH(h, "spectral weight gradient");
  return 0;
}

// Both gradients of one layer in one pass over the tile grid of the DATA gradient: the spectrum of dz's windows (with halo) is computed once
// and serves the data gradient (channel mixing with the flipped filter, inverse transform -> dx) AND the weight gradient, which is formed
// input-partitioned - dw[t] = sum over x tiles (their own Vy x Vx values, zero elsewhere: a cheap masked transform) of
// corr(x tile, dz window)[k-1-t] - so that neither x's windows nor a second, masked copy of dz have to be transformed.
// d: the forward convolution; dg: its data-gradient convolution (input dz, filter w_flipped (kh,kw,Cout,Cin), output dx or, for
// SYMMETRIC / REFLECT layers, the gradient on the padded domain).
// Diagnostic (tests, tools): the spectrum rows of ONE 64 x 64 window at the image origin, item layout of spectral_common.h (4096 rows x 32 floats per
// channel group), as the forward transform writes them - tests/test_gpu_spectral64.py compares the kernel forms (PCNN_FWD64_RADIX = 2 | 4) row by row.
// the debug exports hand the spectrum out in CANONICAL row order whatever the transform family wrote (spectral_common.h): out[item][r][c] = sp[item][row(r)][c]
__global__ __launch_bounds__(256) void spec_canonical_rows_kernel(const float* __restrict__ sp, float* __restrict__ out, int T, int rows, int P) {
  const int item = blockIdx.y;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * 32; e += gridDim.x * 256) {
    const int r = e >> 5, c = e & 31;
    out[(size_t)item * rows * 32 + e] = sp[pcnn_spec::sp_item(item, rows) + pcnn_spec::sp_row_from_canonical(T, r, P) * RS + c];
  }
}
static int canonical_copy(pcnn_handle h, const float* sp, float* out, int T, int items) {
  const int P = h->spectral_xform == PCNN_XFORM_FFT ? PCNN_SP_P : 64;
  hipLaunchKernelGGL(spec_canonical_rows_kernel, dim3(16, items), dim3(256), 0, h->stream, sp, out, T, T * T, P);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int pcnn_debug_tile_spectrum64(pcnn_handle h, int H, int W, int C, const float* x, int ylim, int xlim, float* out) {
  PCNN_REQUIRE(h, h && x && out && H >= 1 && W >= 1 && C >= 1 && C <= 64, "pcnn_debug_tile_spectrum64: bad argument");
  const int groups = pcnn_cdiv(C, 32);
  const size_t sp_b = align256(sp_bytes((size_t)groups, 4096));
  char* r;
  if (int rc = ensure_workspace(h, sp_b, &r)) return rc;
  const Geom gm = geom_of(h, 64);
  FwdParams f;
  f.x = x; f.sp = reinterpret_cast<float*>(r); f.tab = gm.tab; f.H = H; f.W = W; f.C = C; f.ld = C; f.groups = groups; f.cstride = 32; f.cvalid = 32;
  f.tiles_x = 1; f.tiles_y = 1; f.tile0 = 0; f.Vy = 64; f.Vx = 64; f.oy = 0; f.ox = 0; f.pad_mode = PCNN_PAD_CONSTANT; f.pad_value = 0.f;
  f.ylim = ylim; f.xlim = xlim; f.ext_y = 1 << 30; f.ext_x = 1 << 30; f.pack = 1; f.cpt = 32; f.tgx = 1;
  launch_fwd(h, gm, f, 1);
  PCNN_CHECK_LAUNCH(h, "pcnn_debug_tile_spectrum64");
  if (canonical_copy(h, f.sp, out, 64, groups)) PCNN_FAIL(h, "pcnn_debug_tile_spectrum64: copy failed");
  return 0;
}

// Diagnostic (tests): the 32-point tile spectra of ONE image exactly as the forward transform of a convolution writes them - tiles of Vy x Vx
// outputs, window origin (-oy, -ox), the given padding, optional masks (ylim / xlim < 32: gradient-style tiles) and tile packing (pack = 1, 2, 4, 8:
// that many x-adjacent tiles of a <= 32 / pack-channel image share the lanes) - with the transform kernels the handle currently selects
// (pcnn_set_spectral_transform).  out: [tile groups][channel groups][1024 rows][32] floats.  tests/test_gpu_spectral_fft.py compares the two kernel
// families row by row.
extern "C" int pcnn_debug_forward_spectrum32(pcnn_handle h, int H, int W, int C, const float* x, int Vy, int Vx, int oy, int ox, int pad_mode, float pad_value,
                                             int ylim, int xlim, int pack, float* out, size_t out_floats) {
  PCNN_REQUIRE(h, h && x && out && H >= 1 && W >= 1 && C >= 1 && C <= 64 && Vy >= 1 && Vy <= 32 && Vx >= 1 && Vx <= 32, "pcnn_debug_forward_spectrum32: bad argument");
  PCNN_REQUIRE(h, pack == 1 || ((pack == 2 || pack == 4 || pack == 8) && C <= 32 / pack), "pcnn_debug_forward_spectrum32: pack = %d with %d channels", pack, C);
  const int tiles_y = pcnn_cdiv(H, Vy), tiles_x = pcnn_cdiv(W, Vx), tgx = pcnn_cdiv(tiles_x, pack), groups = pack > 1 ? 1 : pcnn_cdiv(C, 32);
  const int ntile = tiles_y * tgx;
  PCNN_REQUIRE(h, out_floats >= (size_t)ntile * groups * 1024 * 32, "pcnn_debug_forward_spectrum32: output holds %zu floats, %zu needed", out_floats, (size_t)ntile * groups * 1024 * 32);
  const size_t sp_b = align256(sp_bytes((size_t)ntile * groups, 1024));
  char* r;
  if (int rc = ensure_workspace(h, sp_b, &r)) return rc;
  const Geom gm = geom_of(h, 32);
  FwdParams f;
  f.x = x; f.sp = reinterpret_cast<float*>(r); f.tab = gm.tab; f.H = H; f.W = W; f.C = C; f.ld = C; f.groups = groups; f.cstride = 32; f.cvalid = 32;
  f.tiles_x = tiles_x; f.tiles_y = tiles_y; f.tile0 = 0; f.Vy = Vy; f.Vx = Vx; f.oy = oy; f.ox = ox; f.pad_mode = pad_mode; f.pad_value = pad_value;
  f.ylim = ylim; f.xlim = xlim; f.ext_y = 1 << 30; f.ext_x = 1 << 30; f.pack = pack; f.cpt = 32 / pack; f.tgx = tgx;
  if (pack > 1) { f.cstride = f.cpt; f.cvalid = f.cpt; }
  launch_fwd(h, gm, f, ntile);
  PCNN_CHECK_LAUNCH(h, "pcnn_debug_forward_spectrum32");
  if (canonical_copy(h, f.sp, out, 32, ntile * groups)) PCNN_FAIL(h, "pcnn_debug_forward_spectrum32: copy failed");
  return 0;
}

extern "C" int pcnn_conv2d_bwd_spectral_eligible(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg) {
  if (!h || !d || !dg) return 0;
  if (d->Cout > 32 || d->Cin > 64) return 0;
  if (d->pad_mode == PCNN_PAD_CONSTANT && d->pad_value != 0.f) return 0;
  if (pcnn_conv_fwd_takes_narrow_route(h, dg)) return 0;
  if (pcnn_spectral_eligible(h, dg, false) && pcnn_spectral_eligible(h, d, true)) return 1;
  // Neither gradient alone may beat its direct kernel while the FUSED pass does: it transforms dz once for both (measured: 1.6-1.75 x one
  // spectral convolution for data + weight gradient together).  5 x 5, 20 -> 16 at 8 x 1024^2 (final/stage4/conv of hpnn.json): direct data
  // gradient 1.99 ms + direct weight gradient 1.73 ms against 2.97 ms fused (tools/probe_k5.py).
  double ts_g = 0.0, ta_g = 0.0, ts_w = 0.0, ta_w = 0.0;
  if (spectral_costs(h, dg, false, &ts_g, &ta_g) != 2 || spectral_costs(h, d, true, &ts_w, &ta_w) != 2) return 0;
  return 1.75 * std::max(ts_g, ts_w) < 0.9 * (ta_g + ta_w) ? 1 : 0;
}

extern "C" int pcnn_conv2d_bwd_spectral_post_eligible(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg) {
  if (!pcnn_conv2d_bwd_spectral_eligible(h, d, dg)) return 0;
  // POST indexes act_out / raw_out / dbias with the pixel offsets of dx.  For SYMMETRIC / REFLECT layers the fused pass produces the gradient on the
  // PADDED domain (Ho = H + kh - 1), whose offsets are not the producer's: the library refuses those itself, whatever the caller checked.
  if (d->pad_mode != PCNN_PAD_CONSTANT) return 0;
  return d->Cin <= 32 ? 1 : 0;                                       // both inverse kernels carry the POST epilogue; one channel group
}

static int bwd_spectral_impl(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg, const float* x, const float* dz, const float* w_flipped,
                             const float* residual, float* dx, float* dw, const pcnn_post_desc* post);

extern "C" int pcnn_conv2d_bwd_spectral(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg, const float* x, const float* dz, const float* w_flipped,
                                        const float* residual, float* dx, float* dw) {
  return bwd_spectral_impl(h, d, dg, x, dz, w_flipped, residual, dx, dw, nullptr);
}

extern "C" int pcnn_conv2d_bwd_spectral_post(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg, const float* x, const float* dz, const float* w_flipped,
                                             const float* residual, float* dx, float* dw, const pcnn_post_desc* post) {
  PCNN_REQUIRE(h, h && post && post->act_out, "pcnn_conv2d_bwd_spectral_post: null argument");
  PCNN_REQUIRE(h, pcnn_conv2d_bwd_spectral_post_eligible(h, d, dg), "pcnn_conv2d_bwd_spectral_post: layer is not eligible (ask pcnn_conv2d_bwd_spectral_post_eligible first)");
  return bwd_spectral_impl(h, d, dg, x, dz, w_flipped, residual, dx, dw, post);
}

static int bwd_spectral_impl(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg, const float* x, const float* dz, const float* w_flipped,
                             const float* residual, float* dx, float* dw, const pcnn_post_desc* post) {
  PCNN_REQUIRE(h, h && d && dg && x && dz && w_flipped && dx && dw, "pcnn_conv2d_bwd_spectral: null argument");
  PCNN_REQUIRE(h, pcnn_conv2d_bwd_spectral_eligible(h, d, dg), "pcnn_conv2d_bwd_spectral: layer is not eligible (ask pcnn_conv2d_bwd_spectral_eligible first)");
  PCNN_REQUIRE(h, dg->Cin == d->Cout && dg->Cout == d->Cin && dg->kh == d->kh && dg->kw == d->kw && dg->N == d->N, "pcnn_conv2d_bwd_spectral: descriptors do not match");
  const int Tg = pick_tile(h, dg);
  const int Vy = Tg - d->kh + 1, Vx = Tg - d->kw + 1;
  const int tiles_y = pcnn_cdiv(dg->Ho, Vy), tiles_x = pcnn_cdiv(dg->Wo, Vx);
  const int pack = Tg == 64 ? 1 : pack_for(d->Cin, d->Cout), cpt = 32 / pack, tgx = pcnn_cdiv(tiles_x, pack);
  const int64_t ntile = (int64_t)d->N * tiles_y * tgx;
  PCNN_REQUIRE(h, ntile < (1ll << 30), "spectral convolution: too many tiles");
  const int gz = 1, gx = pcnn_cdiv(d->Cin, 32), S = wgrad_splits(Tg);       // channel groups of dz (<= 32 channels) and of x / dx
  const int rows = Tg * Tg, nslot = rows / 2;
  // workspace: [filter spectrum | dz spectra (gz) | dx spectra (gx) | x-tile spectra (gx) | partial sums]; C^ reuses the filter-spectrum slot
  const size_t wsp_b = align256(sp_bytes((size_t)std::max(dg->Cin * gx, d->Cin), rows));
  const size_t part_b = align256((size_t)S * nslot * gx * 4 * 1024 * 4);
  auto zs_bytes = [&](int ch) { return align256(sp_bytes((size_t)ch * gz, rows)); };
  auto ys_bytes = [&](int ch) { return align256(sp_bytes((size_t)pad32(ch) * gx, rows)); };
  const int chunk = fit_chunk(h, (int)std::min<int64_t>(chunk_tiles(Tg) / gx, ntile), [&](int ch) { return wsp_b + part_b + 4096 + zs_bytes(ch) + 2 * ys_bytes(ch); });
  const size_t zs_b = zs_bytes(chunk), ys_b = ys_bytes(chunk);
  char* r;
  const size_t bs_b = post ? (size_t)256 * 8 * 64 * 4 * sizeof(float) : 0;   // POST: one partial bias sum per (workgroup, wave, lane) - four at 64 points (a lane holds four channels)
  if (int rc = ensure_workspace(h, wsp_b + zs_b + 2 * ys_b + part_b + bs_b + 4096, &r)) return rc;
  const Geom gm = geom_of(h, Tg);
  float* wsp = reinterpret_cast<float*>(r); r += wsp_b;
  float* zs = reinterpret_cast<float*>(r); r += zs_b;
  float* ys = reinterpret_cast<float*>(r); r += ys_b;
  float* xs = reinterpret_cast<float*>(r); r += ys_b;
  float* part = reinterpret_cast<float*>(r); r += part_b;
  float* bsum = post ? reinterpret_cast<float*>(r) : nullptr;
  // flipped filter spectrum -> mixing matrices of the data gradient (input groups: dz's, output groups: dx's)
  PCNN_REQUIRE(h, gx == 1 || dg->Cout == 64, "pcnn_conv2d_bwd_spectral: %d input channels unsupported (<= 32 or 64)", d->Cin);
  const float* fsp = filter_spectrum(h, gm, w_flipped, wsp, d->kh, d->kw, dg->Cin, dg->Cout, gx);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_bwd_spectral (filter spectrum)");
  FwdParams fz;                                          // dz windows with halo: the data gradient's input transform
  fz.x = dz; fz.sp = zs; fz.tab = gm.tab; fz.H = dg->H; fz.W = dg->W; fz.C = dg->Cin; fz.ld = dg->ldx; fz.groups = gz; fz.cstride = 32; fz.cvalid = 32;
  fz.tiles_x = tiles_x; fz.tiles_y = tiles_y; fz.Vy = Vy; fz.Vx = Vx; fz.oy = dg->pad_top; fz.ox = dg->pad_left; fz.pad_mode = dg->pad_mode;
  fz.pad_value = dg->pad_value; fz.ylim = Tg; fz.xlim = Tg; fz.ext_y = 1 << 30; fz.ext_x = 1 << 30;
  fz.pack = pack; fz.cpt = cpt; fz.tgx = tgx;
  if (pack > 1) { fz.cstride = cpt; fz.cvalid = cpt; }
  FwdParams fxm = fz;                                    // x tiles: own Vy x Vx values (boundary-condition padded where the grid is the padded domain)
  const bool padded_domain = d->pad_mode != PCNN_PAD_CONSTANT;
  fxm.x = x; fxm.sp = xs; fxm.H = d->H; fxm.W = d->W; fxm.C = d->Cin; fxm.ld = d->ldx; fxm.groups = gx;
  fxm.oy = padded_domain ? d->pad_top : 0; fxm.ox = padded_domain ? d->pad_left : 0; fxm.pad_mode = d->pad_mode; fxm.pad_value = 0.f;
  fxm.ylim = Vy; fxm.xlim = Vx; fxm.ext_y = dg->Ho; fxm.ext_x = dg->Wo;
  InvParams iv;
  iv.sp = ys; iv.tab = gm.tab; iv.y = dx; iv.bias = nullptr; iv.bn_scale = nullptr; iv.bn_shift = nullptr; iv.res = residual; iv.act_out = nullptr; iv.absmax = nullptr;
  iv.Ho = dg->Ho; iv.Wo = dg->Wo; iv.C = dg->Cout; iv.ldy = dg->ldy; iv.ld_res = dg->ld_res; iv.ld_act = 0; iv.groups = gx; iv.cstride = 32; iv.cvalid = 32;
  iv.act = PCNN_ACT_LINEAR; iv.alpha = 0.f; iv.tiles_x = tiles_x; iv.tiles_y = tiles_y; iv.Vy = Vy; iv.Vx = Vx; iv.flip = 0;
  iv.pack = pack; iv.cpt = cpt; iv.tgx = tgx;
  if (pack > 1) { iv.cstride = cpt; iv.cvalid = cpt; }
  if (post) {
    iv.gact = post->act_out; iv.ld_gact = post->ld_act_out; iv.gmode = post->act; iv.galpha = post->act_alpha;
    iv.y2 = post->raw_out; iv.ld_y2 = post->ld_raw; iv.bsum = post->dbias ? bsum : nullptr;
    if (iv.bsum && hipMemsetAsync(bsum, 0, bs_b, h->stream) != hipSuccess) PCNN_FAIL(h, "pcnn_conv2d_bwd_spectral_post: memset failed");
  }
  MixParams mx;
  mx.xs = zs; mx.ys = ys; mx.wsp = fsp; mx.gin = gz; mx.gout = gx; mx.Cin = dg->Cin; mx.cpt = cpt;
  WMixParams wm;
  wm.xs = xs; wm.ds = zs; wm.part = part; wm.slots = gm.slots; wm.gin = gx; wm.S = S / 4; wm.rows = rows; wm.nslot = nslot;
  static const int fused_mix = getenv("PCNN_SPEC_MIXW") ? atoi(getenv("PCNN_SPEC_MIXW")) : 1;   // developer switch (A/B timing): 0 = two kernels
  MixWParams mw;
  mw.zs = zs; mw.xs = xs; mw.ys = ys; mw.part = part; mw.wsp = fsp; mw.slots = gm.slots; mw.gx = gx; mw.rows = rows; mw.nslot = nslot;
  mw.Cz = dg->Cin; mw.cpt = cpt;
  for (int64_t t0 = 0; t0 < ntile; t0 += chunk) {
    const int nt = (int)std::min<int64_t>(chunk, ntile - t0);
    fz.tile0 = (int)t0; fxm.tile0 = (int)t0; iv.tile0 = (int)t0; wm.ntile = nt; wm.accumulate = t0 > 0;
    launch_fwd(h, gm, fz, nt);
    if (fused_mix) {                                     // one pass over D^ for both gradients (spec_mixw_kernel)
      launch_fwd(h, gm, fxm, nt);
      mw.ntile = nt; mw.accumulate = t0 > 0;
      hipLaunchKernelGGL(spec_mixw_kernel, dim3(nslot, S / 4, gx), dim3(256), 0, h->stream, mw);
      launch_inv(h, gm, iv, nt);
      continue;
    }
    launch_mix(h, gm, mx, gz, gx, nt);
    launch_inv(h, gm, iv, nt);
    launch_fwd(h, gm, fxm, nt);
    hipLaunchKernelGGL(spec_wmix_kernel, dim3(nslot, S / 4, gx), dim3(256), 0, h->stream, wm);
  }
  if (post && post->dbias) {
    if (Tg == 64 && h->spectral_xform == PCNN_XFORM_FFT) launch_post_bias_fft64(h, bsum, 256, dg->Cout, post->dbias);
    else if (Tg == 64) launch_post_bias64(h, bsum, 256, dg->Cout, post->dbias);
    else if (h->spectral_xform == PCNN_XFORM_FFT) launch_post_bias_fft32(h, bsum, pack, cpt, dg->Cout, post->dbias);
    else hipLaunchKernelGGL(spec_post_bias_kernel, dim3((unsigned)dg->Cout), dim3(256), 0, h->stream, bsum, 256 * 8, pack, cpt, post->dbias);
  }
  float* csp = wsp;                                      // the workspace's filter-spectrum slot: every mixing launch above has read it (same stream), or the spectrum came from the cache
  hipLaunchKernelGGL(spec_wcombine_kernel, dim3(nslot, gx), dim3(256), 0, h->stream, part, gm.slots, csp, S, gx, d->Cin, -1.0f, cpt, rows);
  launch_inv(h, gm, taps_params(gm, csp, dw, d->kh, d->kw, d->Cin, d->Cout, 1), 1);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_bwd_spectral");
  return 0;
}
