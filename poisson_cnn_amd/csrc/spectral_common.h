// Shared declarations of the tiled spectral convolution: spectral_conv.hip (32-point tiles, mixing kernels, host side) and spectral64.hip
// (64-point tiles).  Spectrum layout for a tile size T (T = 32 or 64): items (tile x channel group of 32) are T*T rows of 32 floats
// (128-byte channel vectors); rows [0,T): column fx = 0 (real in x, half-complex in y: Re fy = 0..T/2, then Im fy = 1..T/2-1),
// [T,2T): fx = T/2, then for fx = 1..T/2-1: 2T + 2T (fx-1) + {fy | T + fy} = Re | Im of the complex column (tools/spectral_model.py,
// tools/spectral_model64.py).  T*T/2 mixing slots.
#pragma once
#include "pcnn_internal.h"

namespace pcnn_spec {

constexpr int RS = 32;                       // floats between consecutive spectrum rows of an item
// floats between consecutive items: rows x RS plus SP_PAD - the per-frequency kernels gather the same row of many items, i.e. they walk
// memory with the item stride; a power-of-two stride maps those accesses onto few HBM channels
#ifndef PCNN_SP_PAD
#define PCNN_SP_PAD 32
#endif
constexpr int SP_PAD = PCNN_SP_PAD;
__host__ __device__ __forceinline__ int64_t sp_item(int64_t item, int rows) { return item * ((int64_t)rows * RS + SP_PAD); }
__host__ __device__ __forceinline__ size_t sp_bytes(size_t items, int rows) { return items * ((size_t)rows * RS + SP_PAD) * sizeof(float); }

// ---- row order inside an item.  CANONICAL order (the matrix-core transform family, the debug exports and the tests): the header comment above.
// The FFT family (spectral_fft.hip, the default) interleaves the real and the imaginary row of a frequency in blocks of PCNN_SP_P frequencies, so that
// the per-frequency kernels - which read / write the (Re, Im) row pair of ONE frequency of every item - find the pair inside one 256 P-byte piece of the
// item instead of T rows (4 / 8 KB) apart: row(k, part) = 2 P (k / P) + P part + k % P inside a column block, k = frequency index of the column (complex
// column: fy = 0..T-1; real column: the pair index, k = 0 the two real frequencies fy = 0 | T/2, k >= 1: Re | Im of fy = k).  P = T (T/2 for the real
// columns) IS the canonical order.  The mixing kernels only follow the slot tables (rr, ri), built from the same function for each family.
#ifndef PCNN_SP_P
#define PCNN_SP_P 1
#endif
__host__ __device__ constexpr int sp_blk(int P, int k, int part) { return 2 * P * (k / P) + P * part + (k % P); }
__host__ __device__ constexpr int sp_pc(int T, int P) { return P < T ? P : T; }             // block size of a complex column
__host__ __device__ constexpr int sp_pr(int T, int P) { return P < T / 2 ? P : T / 2; }     // ... of a real column
// complex column of a T-point tile: row (inside the column's 2 T rows) of part (0 real, 1 imaginary) of frequency fy
__host__ __device__ constexpr int sp_row_c(int T, int fy, int part, int P = PCNN_SP_P) { return sp_blk(sp_pc(T, P), fy, part); }
// real column: row (inside the column's T rows) of half-complex entry s (s <= T/2: Re fy = s; s > T/2: Im fy = s - T/2)
__host__ __device__ constexpr int sp_row_r(int T, int s, int P = PCNN_SP_P) {
  return s == T / 2 ? sp_blk(sp_pr(T, P), 0, 1) : (s < T / 2 ? sp_blk(sp_pr(T, P), s, 0) : sp_blk(sp_pr(T, P), s - T / 2, 1));
}
// canonical row r of an item -> its row in the order with block size P
__host__ __device__ constexpr int sp_row_from_canonical(int T, int r, int P = PCNN_SP_P) {
  return r < 2 * T ? (r / T) * T + sp_row_r(T, r % T, P) : 2 * T + ((r - 2 * T) / (2 * T)) * 2 * T + sp_row_c(T, (r - 2 * T) % T, ((r - 2 * T) % (2 * T)) / T, P);
}
static_assert(sp_row_from_canonical(32, 5, 64) == 5 && sp_row_from_canonical(32, 16, 64) == 16 && sp_row_from_canonical(32, 64 + 37, 64) == 64 + 37 && sp_row_from_canonical(64, 128 + 64 + 9, 64) == 128 + 64 + 9, "P >= T is the canonical order");
static_assert(sp_row_from_canonical(32, 16, 1) == 1 && sp_row_from_canonical(32, 3, 1) == 6 && sp_row_from_canonical(32, 16 + 3, 1) == 7 && sp_row_from_canonical(32, 64 + 32 + 5, 1) == 64 + 11, "P = 1: Re | Im adjacent");
// the mixing slots (rr, ri, kind, 0) of a tile size in the order with block size P: T^2 / 2 of them (kind 1: the slot packs two REAL frequencies)
inline void sp_build_slots(int T, int P, int* slots) {
  int n = 0;
  auto put = [&](int rr, int ri, int kind) { slots[4 * n] = rr; slots[4 * n + 1] = ri; slots[4 * n + 2] = kind; slots[4 * n + 3] = 0; ++n; };
  for (int b = 0; b < 2; ++b) {
    put(b * T + sp_row_r(T, 0, P), b * T + sp_row_r(T, T / 2, P), 1);
    for (int fy = 1; fy < T / 2; ++fy) put(b * T + sp_row_r(T, fy, P), b * T + sp_row_r(T, T / 2 + fy, P), 0);
  }
  for (int fx = 1; fx < T / 2; ++fx)
    for (int fy = 0; fy < T; ++fy) put(2 * T + 2 * T * (fx - 1) + sp_row_c(T, fy, 0, P), 2 * T + 2 * T * (fx - 1) + sp_row_c(T, fy, 1, P), 0);
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
// All LDS reads issued so far have landed; nothing moves across.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Workgroup barrier that orders LDS traffic only (__syncthreads() also drains vmcnt, i.e. waits for the next item's prefetch loads and the
// spectrum stores in flight; the prefetched registers are waited for where they are consumed, global stores need no ordering here).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// accumulator register r of lane (half) <-> row of the 32x32 tile
__device__ __forceinline__ int acc_row(int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); }

// 4 x 4 transpose between the four lanes of a quad and four registers: lane i, register j  <->  lane j, register i.  A 16-byte load gives a
// lane four consecutive CHANNELS of one pixel (or spectrum row), the MFMA operand layout wants a lane to hold ONE channel of four pixels:
// two DPP exchange steps (neighbours, then pairs) convert between the two, so that every global access of the transform kernels moves
// 16 bytes per lane - the texture-address path takes a wave instruction every ~16 cycles whatever its width, and at 4 bytes per lane the
// transforms' loads and stores occupied it for as long as their MFMAs occupy the matrix pipe (in-kernel stamps).
__device__ __forceinline__ float dpp_quad_swap1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); }   // [1,0,3,2]
__device__ __forceinline__ float dpp_quad_swap2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)); }   // [2,3,0,1]
__device__ __forceinline__ void quad_transpose(float& r0, float& r1, float& r2, float& r3, bool odd, bool upper) {
  { const float t = dpp_quad_swap1(odd ? r0 : r1); if (odd) r0 = t; else r1 = t; }
  { const float t = dpp_quad_swap1(odd ? r2 : r3); if (odd) r2 = t; else r3 = t; }
  { const float t = dpp_quad_swap2(upper ? r0 : r2); if (upper) r0 = t; else r2 = t; }
  { const float t = dpp_quad_swap2(upper ? r1 : r3); if (upper) r1 = t; else r3 = t; }
}

struct FwdParams {
  const float* x; float* sp; const float* tab;
  int H, W, C, ld, groups, cstride, cvalid;
  int tiles_x, tiles_y, tile0, ntile;
  int Vy, Vx, oy, ox, pad_mode; float pad_value;
  int ylim, xlim;        // the window holds values only in its first ylim x xlim entries (gradient / input tiles of the backward pass)
  int ext_y, ext_x;      // ... and only where the tile grid coordinate (ty Vy + r, tx Vx + c) lies inside ext_y x ext_x
  int pack, cpt, tgx;    // layers of <= 16 channels: `pack` x-adjacent tiles share the 32 lanes (lane = cpt * tile + channel, cpt = 32 / pack);
                         // tgx = tile groups per tile row; an "item" is then a tile GROUP and tile0 / ntile count groups
};

struct InvParams {
  const float* sp; const float* tab;
  float* y; const float* bias; const float* bn_scale; const float* bn_shift; const float* res; float* act_out; unsigned* absmax;
  int Ho, Wo, C, ldy, ld_res, ld_act, groups, cstride, cvalid, act; float alpha;
  int tiles_x, tiles_y, tile0, ntile, Vy, Vx;
  int flip;              // store output pixel (y, x) at (Ho-1-y, Wo-1-x): the input-partitioned weight gradient comes out tap-reversed
  int pack, cpt, tgx;    // tile packing, as in FwdParams (flip requires pack == 1)
  // POST (data-gradient launches only): the activation backward of the layer that PRODUCED this convolution's input, applied to the gradient
  // before it is stored - v = (conv + residual); y2 (if given) receives v, y receives v * act'(gact) where gact is that layer's saved
  // activation output, and every lane adds what it stored into bsum[(block 8 + wave) 64 + lane] (the bias gradient's partial sums)
  const float* gact = nullptr; float* y2 = nullptr; float* bsum = nullptr;
  int ld_gact = 0, ld_y2 = 0, gmode = 0; float galpha = 1.f;
};

// 64-point tiles (spectral64.hip): table block, slots are built by build_tables64; the launchers take the same parameter blocks (pack = 1)
constexpr int TAB64_FLOATS = 32768;
void build_tables64(float* tab, int* slots);                 // TAB64_FLOATS floats, 2048 x int4 slots
void launch_fwd64(pcnn_handle h, FwdParams p, int ntile);     // p.tab: the 64-point table block
void launch_inv64(pcnn_handle h, InvParams p, int ntile);
void launch_post_bias64(pcnn_handle h, const float* bsum, int nblocks, int C, float* dbias);   // POST at 64 points: bsum holds 4 floats per (block, wave, lane)

// 32-point tiles as in-register FFTs on the vector ALUs (spectral_fft.hip); same parameter blocks, same spectrum layout.  16 waves per workgroup:
// the POST partial sums hold one float per (block, wave, lane) with FFT_WAVES waves per block
constexpr int FFT_WAVES = 16;
void launch_fwd_fft32(pcnn_handle h, FwdParams p, int ntile);
void launch_fwd_fft32_multi(pcnn_handle h, const FwdParams* tab, int count, int max_items);   // `count` one-tile transforms (filters) from a device table of parameter blocks
void launch_fwd_fft64_multi(pcnn_handle h, const FwdParams* tab, int count, int max_items);
void launch_inv_fft32(pcnn_handle h, InvParams p, int ntile);
void launch_post_bias_fft32(pcnn_handle h, const float* bsum, int pack, int cpt, int C, float* dbias);     // POST partial sums of the FFT inverse in use -> dbias
void launch_fwd_fft64(pcnn_handle h, FwdParams p, int ntile);       // 64-point tiles, item = (tile, 16 channels), 8 waves x 2 units
void launch_inv_fft64(pcnn_handle h, InvParams p, int ntile);
void launch_post_bias_fft64(pcnn_handle h, const float* bsum, int nblocks, int C, float* dbias);   // POST at 64 points: one float per (block, wave, lane)

}  // namespace pcnn_spec
