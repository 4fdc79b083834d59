// The transforms of the tiled spectral convolution as IN-REGISTER FFTs on the vector ALUs (round 5).
//
// spectral_conv.hip / spectral64.hip apply the DFT as a GEMM on the matrix cores.  On gfx950 an fp32 MFMA issues at exactly the fp32 vector
// rate (64 FLOP/clk/SIMD), so that form spends T multiply-adds per point where an FFT spends ~2.5 log2 T at the same issue rate - and its
// accumulator tiles cost 256 registers per lane, one workgroup of 8 waves per CU, whose memory and matrix phases add up (profiles/r04: every
// transform kernel at ~0.4 of HBM AND ~0.4 matrix-pipe busy).  Here the CHANNEL stays in the lane (every global access is still a pixel's /
// spectrum row's 128-byte channel vector) and a lane holds a whole 32-point row or column OF ITS CHANNEL in registers: the butterflies never
// cross lanes (fft_regs.h), a tile costs ~7 k vector instructions instead of 1 024 MFMAs (= 65 k SIMD cycles), the kernels need <= 128 registers
// and run 16 waves per CU, so one wave's loads and stores fly under the other waves' arithmetic.
//
//   fft32_fwd_kernel   item = (tile, 32 channels).  x axis: wave w owns window rows 2w, 2w + 1 (one per lane half): 32 loads per lane (prefetched
//                      one item ahead), real FFT in registers (rfft_fwd<32>), half-complex result to LDS U[y][s][c] (128 KB).  y axis: wave 0 owns
//                      the two real columns (fx = 0 / 16, one per lane half: a real FFT again), wave fx = 1..15 the complex column fx with the
//                      output-frequency PARITY in the lane half (one radix-2 decimation-in-frequency step while reading LDS, then cfft_dif<16>);
//                      32 stores per lane, each instruction two whole 128-byte spectrum rows.
//   fft32_inv_kernel   the mirror image + the fused convolution epilogue of spec_inv_kernel: y axis straight from global (wave 0: the real
//                      columns by rfft_inv<32>; wave fx: the even / odd input frequencies per lane half -> E / O halves of a decimation-in-time
//                      step in LDS), x axis: u = E +- O while reading LDS, rfft_inv<32>, epilogue from registers.
// Spectrum layout, parameter blocks, tables of slots: spectral_common.h - identical to the matrix-core kernels, which stay selectable
// (pcnn_set_spectral_transform / PCNN_SPEC_XFORM) and are the reference of tests/test_gpu_spectral_fft.py (spectra equal to <= 1e-6 row by row).
#include "spectral_common.h"
#include "fft_regs.h"
#include <algorithm>
#include <stdlib.h>

#ifndef PCNN_NT
#define PCNN_NT 17
#endif
#define NT_LOAD(bit, p) ((PCNN_NT & (bit)) ? __builtin_nontemporal_load(p) : *(p))
#define NT_STORE(bit, v, p) do { if (PCNN_NT & (bit)) __builtin_nontemporal_store(v, p); else *(p) = (v); } while (0)

// Removal studies (what a kernel costs without its loads / stores / arithmetic / LDS traffic) exist only in a diagnostic build: -DPCNN_FFT_STUDY, bits
// from the environment variable PCNN_FFT_STUDY (1 no spectrum / pixel stores, 2 no global loads, 4 no FFT arithmetic, 8 no LDS reads, 16 no LDS
// writes).  The shipped library compiles every test out.
#ifdef PCNN_FFT_STUDY
__constant__ int g_fft_study;
#define FFT_STUDY(bit) (g_fft_study & (bit))
#else
#define FFT_STUDY(bit) 0
#endif

namespace pcnn_spec {

namespace {
using namespace pcnn_fft;

constexpr int T = 32, ROWS = 1024;
constexpr int EO = 16 * 32 * 32;                               // floats between the E and the O half of the inverse kernel's LDS image
constexpr size_t LDS_BYTES = (size_t)T * T * 32 * sizeof(float);   // 128 KB
__host__ __device__ __forceinline__ int64_t sp_item32(int64_t item) { return pcnn_spec::sp_item(item, ROWS); }

// tf.pad index map without control flow (selects only); constant padding: any valid pixel (replaced when the value is consumed)
__device__ __forceinline__ int pad_sel(int i, int n, int mode) {
  const int refl = mode == PCNN_PAD_SYMMETRIC ? (i < 0 ? -i - 1 : 2 * n - 1 - i) : (i < 0 ? -i : 2 * n - 2 - i);
  const int rc = min(max(refl, 0), n - 1);
  return (unsigned)i < (unsigned)n ? i : (mode == PCNN_PAD_CONSTANT ? 0 : rc);
}

// ------------------------------------------------------------------------------------------------------------------ forward transform
// (Measured and removed, round 5 - commit "Forward transform loads through hand-managed vmcnt": the window loads issued from inline assembly and waited for
// by hand with s_waitcnt vmcnt(32), so that the x phase starts while the previous item's 32 spectrum stores are still draining - the compiler's own
// waits are merged conservatively over the kernel's paths.  Correct in the shipped build (all spectral tests) and not faster: 7 taps forward 1.86 vs
// 1.84 ms - the kernel sits at the copy ceiling either way.  Removed because it is fragile: the compiler does not know the
// destination registers are in flight and may copy or reuse them; the diagnostic build of the very same source (-DPCNN_FFT_STUDY) faulted.)
// what a lane keeps of an item between requesting its window row and consuming it
struct FwdRow {
  float fill;            // lane: value of a row that is not read from memory (0: beyond ylim / no such channel; the padding constant)
  int cl, cr;            // lane: window columns x < cl or x >= cr are constant padding (boundary windows with CONSTANT padding; else 0, T)
  int xlim;              // lane: columns >= xlim are zero (gradient / input tiles of the backward pass)
  bool use_fill;         // lane
  bool fast;             // uniform: every window of the item lies inside the image in x
};

// requests window row y (lane: y = 2 wave + half) of `item`: 32 loads with always-valid addresses; padding and masks are applied on consumption
__device__ __forceinline__ void fwd_request(const FwdParams& p, int item, int wave, int y, int c, FwdRow& it, float (&R)[32]) {
  const int g = item % p.groups;
  int t = p.tile0 + item / p.groups;
  const int txg = t % p.tgx; t /= p.tgx;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int sub = p.pack > 1 ? c / p.cpt : 0, cc = c - sub * p.cpt;            // this lane's tile of the group, and its channel
  const int tx = txg * p.pack + sub;
  const int chan = g * p.cstride + cc;
  const bool cok = cc < p.cvalid && chan < p.C && tx < p.tiles_x;
  const float* img = p.x + (int64_t)n * p.H * p.W * p.ld;
  const int wy0 = ty * p.Vy - p.oy, wx0 = tx * p.Vx - p.ox;
  const int ylim = min(p.ylim, p.ext_y - ty * p.Vy);                           // uniform
  it.xlim = min(p.xlim, p.ext_x - tx * p.Vx);
  const int gy = wy0 + y;
  const bool rowconst = p.pad_mode == PCNN_PAD_CONSTANT && (unsigned)gy >= (unsigned)p.H;
  const bool rowzero = !cok || y >= ylim;
  it.use_fill = rowzero || rowconst;
  it.fill = rowzero ? 0.f : p.pad_value;
  it.cl = 0; it.cr = T;
  const int wx_first = txg * p.pack * p.Vx - p.ox, wx_last = wx_first + (p.pack - 1) * p.Vx;
  it.fast = wx_first >= 0 && wx_last + T <= p.W;                               // uniform
  if (2 * wave >= ylim || FFT_STUDY(2)) return;                                // uniform: both rows of this wave are zero rows - nothing to fetch
  const int sy = pad_sel(gy, p.H, p.pad_mode);
  const unsigned ch = (unsigned)(cok ? chan : 0);
  if (it.fast) {
    // one lane offset for all 32 loads; the uniform pointer steps from pixel to pixel on the scalar ALU
    const unsigned lo = (unsigned)((sy * p.W + wx0) * p.ld) + ch;
    const float* rp = img;
#pragma unroll
    for (int x = 0; x < T; ++x) { R[x] = rp[lo]; rp += p.ld; }
  } else {
    const unsigned rowoff = (unsigned)(sy * p.W * p.ld) + ch;
    if (p.pad_mode == PCNN_PAD_CONSTANT) { it.cl = -wx0; it.cr = p.W - wx0; }
#pragma unroll
    for (int x = 0; x < T; ++x) R[x] = img[rowoff + (unsigned)(pad_sel(wx0 + x, p.W, p.pad_mode) * p.ld)];
  }
}

template <bool MASKED>
__device__ __forceinline__ void fwd_consume(const FwdParams& p, const FwdRow& it, float (&R)[32]) {
#pragma unroll
  for (int x = 0; x < T; ++x) {
    float v = R[x];
    if (!it.fast && (x < it.cl || x >= it.cr)) v = p.pad_value;
    if (it.use_fill) v = it.fill;
    if (MASKED && x >= it.xlim) v = 0.f;
    R[x] = v;
  }
}

// The values are needed NOW: without this the compiler sinks a whole sub-transform to its first use - a phase later - and keeps (spills) its 64
// inputs instead of its 32 results.  No instruction is emitted.
template <int N> __device__ __forceinline__ void pin(float* a) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]));
}

// half-complex entry s of a transformed row held as rfft_fwd<32> leaves it: s <= 16: Re X[s], s > 16: Im X[s - 16]
__device__ __forceinline__ float hc_get(const float (&R)[32], int s) {
  bool neg = false;
  const int pos = s <= 16 ? rfft_pos(32, s, false, neg) : rfft_pos(32, s - 16, true, neg);
  return neg ? -R[pos] : R[pos];
}
__device__ __forceinline__ void hc_put(float (&R)[32], int s, float v) {
  bool neg = false;
  const int pos = s <= 16 ? rfft_pos(32, s, false, neg) : rfft_pos(32, s - 16, true, neg);
  R[pos] = neg ? -v : v;
}

// (the kernel body as a function of (parameter block, first item, item stride): fft32_fwd_kernel walks one layer, fft32_fwd_multi_kernel one entry
// of a table of layers per blockIdx.y - the filter spectra of a whole model in one launch, pcnn_set_filter_version)
template <bool MASKED>
__device__ __forceinline__ void fft32_fwd_body(const FwdParams& p, float* U, const int bid, const int gdim) {          // U[(y*32 + s)*32 + c]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = p.ntile * p.groups;
  int item = bid;
  if (item >= total) return;
  const int y = 2 * wave + half;
  float R[32];
  FwdRow cur;
  fwd_request(p, item, wave, y, c, cur, R);
  // ---- x axis: this lane's window row, real -> half-complex, into LDS
  auto x_phase = [&]() {
    fwd_consume<MASKED>(p, cur, R);
    if (!FFT_STUDY(4)) rfft_fwd<32>(R);
    float* u = U + (y * 32) * 32 + c;
#pragma unroll
    for (int s = 0; s < T; ++s) u[s * 32] = hc_get(R, s);
  };
  // (first x phase outside the loop: every x phase inside it then meets the same memory-counter state - the next item's loads, then this item's
  // stores - as in spec_fwd_kernel)
  x_phase();
  for (;;) {
    const int next = item + gdim;
    if (next < total) fwd_request(p, next, wave, y, c, cur, R);      // lands under the y phase below
    lds_barrier();
    float* out = p.sp + sp_item32(item);
    float V[32];
    if (wave == 0) {
      // the two real columns fx = 0 (lanes 0-31) and fx = 16 (lanes 32-63): half-complex along y as well
      const float* u = U + (half ? 16 : 0) * 32 + c;
#pragma unroll
      for (int yy = 0; yy < T; ++yy) V[yy] = u[yy * 1024];
      if (!FFT_STUDY(4)) rfft_fwd<32>(V);
      // (stores: uniform base + 32-bit lane offset + immediate - no 64-bit address per 4 KB window in vector registers)
      const unsigned lo = (unsigned)((half ? 32 : 0) * RS + c);
#pragma unroll
      for (int s = 0; s < T; ++s) if (!FFT_STUDY(1)) NT_STORE(1, hc_get(V, s), &(out + (sp_row_r(T, s) * RS))[lo]);
    } else {
      // complex column fx = wave; lane half = parity of the output frequencies: Z[2m + par] = FFT16( (u[y] +- u[y + 16]) W32^(par y) )[m]
      const float* ur = U + wave * 32 + c, *ui = U + (16 + wave) * 32 + c;
      float* vr = V, *vi = V + 16;
      int par = half;
      asm volatile("" : "+v"(par));                                  // opaque: the 30 per-lane twiddle selects below are formed here, item by item -
                                                                     // left alone they are hoisted out of the persistent loop into 30 registers (spills)
      const float osign = par ? -1.f : 1.f;
#pragma unroll
      for (int yy = 0; yy < 16; ++yy) {
        const float lr = ur[yy * 1024], hr = ur[(yy + 16) * 1024], li = ui[yy * 1024], hi = ui[(yy + 16) * 1024];
        const float ar = fma_(osign, hr, lr), ai = fma_(osign, hi, li);
        if (yy == 0) { vr[yy] = ar; vi[yy] = ai; }
        else {
          const float wr = par ? tw_re<32>(yy) : 1.f, wi = par ? tw_im<32>(yy) : 0.f;
          vr[yy] = fma_(ar, wr, -(ai * wi));
          vi[yy] = fma_(ar, wi, ai * wr);
        }
      }
      if (!FFT_STUDY(4)) cfft_dif<16, -1>(vr, vi);
      // register q holds Z[2 bitrev(q) + par]: rows sp_row_c(fy = 2m + par, part) of the column's 64 (spectral_common.h; the odd frequency of a
      // pair lies a constant number of rows behind the even one: the lane half's offset)
      float* o = out + (64 + 64 * (wave - 1)) * RS;                   // uniform
      const unsigned lo = (unsigned)(half * (sp_row_c(T, 1, 0) - sp_row_c(T, 0, 0)) * RS + c);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (FFT_STUDY(1)) continue;
        NT_STORE(1, vr[bitrev(m, 16)], &(o + (sp_row_c(T, 2 * m, 0) * RS))[lo]);
        NT_STORE(1, vi[bitrev(m, 16)], &(o + (sp_row_c(T, 2 * m, 1) * RS))[lo]);
      }
    }
    if (next >= total) break;
    item = __builtin_amdgcn_readfirstlane(next);
    lds_barrier();                                                       // U is free for the next item
    x_phase();
  }
}

template <bool MASKED>
__global__ __launch_bounds__(1024) void fft32_fwd_kernel(FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float U[];
  fft32_fwd_body<MASKED>(p, U, blockIdx.x, gridDim.x);
}

// a parameter block from a device table into scalar registers (the table index is uniform; readfirstlane says so to the compiler)
__device__ __forceinline__ FwdParams load_fwd_params(const FwdParams* q) {
  static_assert(sizeof(FwdParams) % 4 == 0, "FwdParams is copied word by word");
  FwdParams p;
  const int* s = reinterpret_cast<const int*>(q);
  int* d = reinterpret_cast<int*>(&p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(FwdParams) / 4); ++i) d[i] = __builtin_amdgcn_readfirstlane(s[i]);
  return p;
}

// blockIdx.y = table entry (one filter: a one-tile "image" with Cin x Cout channels), blockIdx.x walks its items
__global__ __launch_bounds__(1024) void fft32_fwd_multi_kernel(const FwdParams* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) float U[];
  const FwdParams p = load_fwd_params(tab + blockIdx.y);
  fft32_fwd_body<true>(p, U, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------ inverse transform + epilogue
// The fused convolution epilogue of one output row held in registers (lane = channel: per-channel constants are per-lane scalars), NPIX pixels,
// the first vx of them real.  bias -> activation -> act_out -> BN affine -> residual -> y, or - POST, the data-gradient launches - residual -> y2 ->
// times act'(producer's activation) -> y, with the bias-gradient partial sum.
// Written for the memory counter (vmcnt is in order, and a store's data register may not be rewritten before the store has read it):
//   * per burst of BURST pixels: every input load first, then every value into a register OF ITS OWN, then the stores back to back;
//   * NO store sits behind a branch: a pixel beyond vx re-stores the row's first pixel (same address, same value) - with a static number of memory
//     operations per burst the compiler's waits name exactly what they need.  (First version: `if (x < vx) store` - every store in a basic block of
//     its own, re-using one data register: an s_waitcnt vmcnt(0) per pixel, i.e. every store COMPLETED before the next pixel was formed; the
//     64-point inverse spent half its time there - tools/study_fft.sh.)
// Round 6: a whole burst beyond the row's end is skipped (the round-5 form re-stored the row's first pixel for every pixel beyond vx so that the number of
// memory operations per row stayed static; with bursts of four, a quarter of the bursts of an 11-tap layer and an eighth of a 7-tap layer's lie wholly beyond
// the end).  One uniform branch per burst; measured on the 32-point inverse (profiles/r06_step_ab_epilogue_skip.txt): -3 % on the pipelined variants (bit 1),
// -2...3 % on the others (bit 2); on the 64-point inverse (bit 4; bursts of eight: a 15-tap row has 50 of 64 pixels) -3...7 % per variant - round 5 had measured
// -1.3 % forward / +2 % backward for the same idea, before the stores went non-temporal.
#ifndef PCNN_EPI_SKIP
#define PCNN_EPI_SKIP 7
#endif
struct NoBetween { __device__ __forceinline__ void operator()(int) const {} };
// `between(b)` runs after the stores of burst b: the 32-point inverse issues a slice of the NEXT item's spectrum loads there (inv32_pipe)
template <bool TANH, bool RES, bool POST, int NPIX, int BURST, bool NT, typename Between = NoBetween>
__device__ __forceinline__ void epilogue_row(const InvParams& p, const float* X, float scale, int vx, unsigned pix0, int sgn, unsigned chv, float bias, float sc, float sh,
                                             float* yimg, float* aimg, const float* rimg, const float* gimg, float* y2img, float& ymax, float& bsum, Between between = Between(),
                                             int vx_uniform = NPIX) {
  auto st = [](float v, float* q) { if (NT) __builtin_nontemporal_store(v, q); else *q = v; };
  float first_y = 0.f, first_a = 0.f, first_y2 = 0.f;              // what the row's first pixel stores (pixels beyond vx repeat it)
#pragma unroll
  for (int x0b = 0; x0b < NPIX; x0b += BURST) {
#if PCNN_EPI_SKIP
    if (x0b >= vx_uniform) { between(x0b / BURST); continue; }      // a whole burst beyond the row's end (uniform): nothing to re-store
#endif
    float rv[BURST], gv[BURST], oy[BURST], oa[BURST], oy2[BURST];
    if (RES) {
#pragma unroll
      for (int r = 0; r < BURST; ++r) rv[r] = rimg[(pix0 + (unsigned)(x0b + r < vx ? sgn * (x0b + r) : 0)) * (unsigned)p.ld_res + chv];
    }
    if (POST) {
#pragma unroll
      for (int r = 0; r < BURST; ++r) gv[r] = gimg[(pix0 + (unsigned)(x0b + r < vx ? sgn * (x0b + r) : 0)) * (unsigned)p.ld_gact + chv];
    }
#pragma unroll
    for (int r = 0; r < BURST; ++r) {
      const bool ok = x0b + r < vx;
      float v = X[x0b + r] * scale;
      if (!POST) {                                                   // (a data-gradient launch has no bias, activation, BN or act_out)
        v += bias;
        v = TANH ? tanhf(v) : (v > 0.f ? v : v * p.alpha);
        oa[r] = v;
        v = v * sc + sh;
      }
      if (RES) v += rv[r];
      if (POST) {
        oy2[r] = v;
        const float gq = gv[r];
        v *= p.gmode == PCNN_ACT_TANH ? 1.f - gq * gq : (gq > 0.f ? 1.f : p.galpha);
        bsum += ok ? v : 0.f;
      }
      oy[r] = v;
      ymax = fmaxf(ymax, ok ? fabsf(v) : 0.f);
      if (x0b + r == 0) { first_y = v; if (!POST) first_a = oa[0]; if (POST) first_y2 = oy2[0]; }
    }
    __builtin_amdgcn_sched_barrier(0);                               // the values first, then the stores: nothing of the next burst in between
    // (the optional second output is tested once per burst, not per pixel: a branch between two stores makes their number unknown to the waits)
    float* const second = POST ? y2img : aimg;
    const unsigned ld2 = (unsigned)(POST ? p.ld_y2 : p.ld_act);
    if (second) {
#pragma unroll
      for (int r = 0; r < BURST; ++r) {
        const bool ok = x0b + r < vx;
        const unsigned pix = pix0 + (unsigned)(ok ? sgn * (x0b + r) : 0);
        st(ok ? (POST ? oy2[r] : oa[r]) : (POST ? first_y2 : first_a), &second[pix * ld2 + chv]);
        st(ok ? oy[r] : first_y, &yimg[pix * (unsigned)p.ldy + chv]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < BURST; ++r) {
        const bool ok = x0b + r < vx;
        const unsigned pix = pix0 + (unsigned)(ok ? sgn * (x0b + r) : 0);
        st(ok ? oy[r] : first_y, &yimg[pix * (unsigned)p.ldy + chv]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    between(x0b / BURST);
  }
}

// requests this wave's column of an item (`in`: its spectrum): wave 0 - lane half = real column 0 / 16, its 32 half-complex entries in row order;
// wave fx - lane half = parity, B[m] = Re Z[2m + par], B[16 + m] = Im Z[2m + par] (spectrum rows 2j + par of the column's 64, j = 0..31).
// ONE code path for both (a uniform row step on the scalar ALU): the prefetch registers then have a single definition point in the item loop.
__device__ __forceinline__ void inv_request(const float* in, int wave, int half, int c, float (&B)[32]) {
  const float* sp = in + (wave == 0 ? 0 : (64 + 64 * (wave - 1)) * RS);
  const unsigned lo = (unsigned)((wave == 0 ? 32 : sp_row_c(T, 1, 0) - sp_row_c(T, 0, 0)) * RS * half + c);
  if constexpr (PCNN_SP_P == 1) {
    // Re | Im adjacent: entry j = (m = j & 15, part = j >> 4) sits at row 2 m + part of a real column and at row 4 m + part (+ 2 par: the lane offset) of
    // a complex one - ONE code path with a uniform row step on the scalar ALU, as with the canonical order (a per-entry select between two row constants
    // costs a scalar select per load: the inverse kernels ran 4-9 % slower with it)
    static_assert(sp_row_r(T, 17, 1) == 3 && sp_row_r(T, 16, 1) == 1 && sp_row_c(T, 6, 1, 1) == 13, "row order the stepping form assumes");
    const int step = (wave == 0 ? 2 : 4) * RS;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const float* q = sp + part * RS;
#pragma unroll
      for (int m = 0; m < 16; ++m) { B[16 * part + m] = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &q[lo]); q += step; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < T; ++j) {
      // wave 0: half-complex entry j of the real column; wave fx: Re (j < 16) / Im of Z[2 (j & 15) + par] - a uniform (scalar) choice between two row constants
      const int row = wave == 0 ? sp_row_r(T, j) : sp_row_c(T, 2 * (j & 15), j >> 4);
      B[j] = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &(sp + row * RS)[lo]);
    }
  }
}

// entries [J0, J0 + NJ) of inv_request's B (interleaved row order: a uniform pointer step): the slice of the next item's loads that one epilogue burst issues
template <int J0, int NJ>
__device__ __forceinline__ void inv_request_slice(const float* in, int wave, int half, int c, float (&B)[32]) {
  const float* sp = in + (wave == 0 ? 0 : (64 + 64 * (wave - 1)) * RS);
  const unsigned lo = (unsigned)((wave == 0 ? 32 : 2) * RS * half + c);
  const int step = (wave == 0 ? 2 : 4) * RS;
#pragma unroll
  for (int j = J0; j < J0 + NJ; ++j) B[j] = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &(sp + (j >> 4) * RS + (j & 15) * step)[lo]);
}

// Prefetch of the next item's spectrum column (32 registers that stay live through the x axis and the epilogue): with it the epilogue variants spill
// 20-26 registers, and a spill costs more than the prefetch hides (scratch traffic shares the memory pipe) - off; measured both ways, DESIGN.md 4.8.
#ifndef PCNN_INV32_PREFETCH
#define PCNN_INV32_PREFETCH 0
#endif
constexpr bool INV32_PREFETCH = PCNN_INV32_PREFETCH;
// Round 6: the next item's 32 spectrum loads issued in slices BETWEEN the epilogue's store bursts (each slice into the registers of the pixels the burst has just
// stored), so that their latency runs under the rest of the epilogue and the barrier instead of behind this item's stores in the in-order memory counter.  Measured per
// epilogue variant on one box (profiles/r06_step_ab_inv32_pipe.txt, ms per train step, plain / bursts of 4 / bursts of 4 + pipelined loads): plain epilogue
// 16.1 / 16.1 / 15.6, residual + POST 3.30 / 3.29 / 3.2 - but POST alone 7.4 / 8.0 / 8.2 and residual alone 4.1 / 4.5 / 4.7 (their bursts of 8 hide more than the
// pipelining returns).  So: pipelined with bursts of four where the epilogue has neither or both extra inputs, the round-5 form otherwise.  PCNN_INV32_PIPE=0: off.
#ifndef PCNN_INV32_PIPE
#define PCNN_INV32_PIPE 1
#endif
template <bool RES, bool POST> constexpr bool inv32_pipe() { return PCNN_INV32_PIPE && !PCNN_INV32_PREFETCH && PCNN_SP_P == 1 && RES == POST; }

#ifndef PCNN_ST32_NT
#define PCNN_ST32_NT 0
#endif
// TANH = false: linear / relu / leaky-relu as one select with the negative-side slope in p.alpha (1 / 0 / alpha)
template <bool TANH, bool RES, bool POST>
__global__ __launch_bounds__(1024) void fft32_inv_kernel(InvParams p) {
  extern __shared__ __attribute__((aligned(16))) float U[];          // E[(y*32 + s)*32 + c], y < 16, then O
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = p.ntile * p.groups;
  int item = blockIdx.x;
  if (item >= total) return;
  constexpr bool INV32_PIPE = inv32_pipe<RES, POST>();
  float B[32];
  if (INV32_PREFETCH || INV32_PIPE) inv_request(p.sp + sp_item32(item), wave, half, c, B);
  float ymax = 0.f, bsum = 0.f;
  constexpr int BURST = (RES == POST) ? 4 : 8;                       // pixels per epilogue burst (registers: 128 per lane; epilogue_row keeps up to 5 values per pixel)
  for (;;) {
    const int next = item + gridDim.x;
    // ---- y axis inverse (unnormalised: the 1 / 1024 of both axes is applied once, after the x axis)
    {
      float V[32];
      if (!INV32_PREFETCH && !INV32_PIPE) inv_request(p.sp + sp_item32(item), wave, half, c, B);
#pragma unroll
      for (int i = 0; i < 32; ++i) V[i] = B[i];
      if (INV32_PREFETCH && next < total) inv_request(p.sp + sp_item32(next), wave, half, c, B);      // lands under the rest of this item
      if (wave == 0) {
        float W[32];
#pragma unroll
        for (int s = 0; s < T; ++s) hc_put(W, s, V[s]);
        if (!FFT_STUDY(4)) rfft_inv<32>(W);                                             // W[y] = 32 u[y]
        float* e = U + (half ? 16 : 0) * 32 + c;
#pragma unroll
        for (int yy = 0; yy < 16; ++yy) {
          e[yy * 1024] = 0.5f * (W[yy] + W[yy + 16]);                // the x axis forms u[y] = E + O, u[y + 16] = E - O for every column alike
          e[EO + yy * 1024] = 0.5f * (W[yy] - W[yy + 16]);
        }
      } else {
        float* vr = V, *vi = V + 16;
        if (!FFT_STUDY(4)) cfft_dif<16, +1>(vr, vi);                                    // register q: E (par = 0) or O-before-twiddle (par = 1) at y = bitrev(q)
        float* er = U + (half ? EO : 0) + wave * 32 + c, *ei = er + 16 * 32;
        int par = half;
        asm volatile("" : "+v"(par));                                // opaque: the per-lane twiddle selects are formed here, not hoisted out of the item loop
#pragma unroll
        for (int yy = 0; yy < 16; ++yy) {
          const float ar = vr[bitrev(yy, 16)], ai = vi[bitrev(yy, 16)];
          if (yy == 0) { er[0] = ar; ei[0] = ai; }
          else {
            const float wr = par ? tw_re<32>(yy) : 1.f, wi = par ? -tw_im<32>(yy) : 0.f;        // conj(W32^y) for the odd half
            er[yy * 1024] = fma_(ar, wr, -(ai * wi));
            ei[yy * 1024] = fma_(ar, wi, ai * wr);
          }
        }
      }
    }
    lds_barrier();
    // ---- x axis inverse of this lane's output row + the fused epilogue (lane = channel: per-channel constants are per-lane scalars)
    {
      const int g = item % p.groups;
      int t = p.tile0 + item / p.groups;
      const int txg = t % p.tgx; t /= p.tgx;
      const int ty = t % p.tiles_y;
      const int n = t / p.tiles_y;
      const int sub = p.pack > 1 ? c / p.cpt : 0, cc = c - sub * p.cpt;
      const int subx = sub * p.Vx;
      const int y0 = ty * p.Vy, x0 = txg * p.pack * p.Vx;
      const int vy = min(p.Vy, p.Ho - y0), vx = min(p.Vx, p.Wo - x0 - subx);
      const int chan = g * p.cstride + cc;
      const bool cok = cc < p.cvalid && chan < p.C;
      // (pipelined variants) the spectrum of the item after this one; the last item re-requests itself: a static number of loads per pass, no branch around them
      const float* nin = p.sp + sp_item32(next < total ? next : item);
      bool piped = false;
      if (2 * wave < vy) {                                            // uniform: at least the even row of this wave is an output row
        const int yy = 2 * wave + half;
        const bool rowok = cok && yy < vy && vx > 0;                  // (vx <= 0: this lane's packed tile lies beyond the image - its first pixel does not exist)
        float X[32];
        {
          // u = E +- O in batches of eight entries: left alone, the scheduler requests all 64 operands first (64 registers beside the 32 of the
          // next item's prefetch: spills)
          const float* e = U + ((yy & 15) * 32) * 32 + c;
          const float osign = wave < 8 ? 1.f : -1.f;
#pragma unroll
          for (int s0 = 0; s0 < T; s0 += 8) {
#pragma unroll
            for (int s = s0; s < s0 + 8; ++s) hc_put(X, s, fma_(osign, e[EO + s * 32], e[s * 32]));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (!FFT_STUDY(4)) rfft_inv<32>(X);                                             // X[x] = 1024 * pixel (yy, x)
        const float bias = (p.bias && cok) ? p.bias[chan] : 0.f;
        const float sc = (p.bn_scale && cok) ? p.bn_scale[chan] : 1.f, sh = (p.bn_scale && cok) ? p.bn_shift[chan] : 0.f;
        // addresses: a uniform IMAGE base per tensor plus an unsigned 32-bit lane offset (scalar base + vector offset form: no 64-bit address
        // lives in vector registers); the lane's first pixel, then one pixel to the right (left when the output is stored flipped) per x
        const int sgn = p.flip ? -1 : 1;
        const int prow = p.flip ? p.Ho - 1 - y0 - yy : y0 + yy, pcol = p.flip ? p.Wo - 1 - x0 - subx : x0 + subx;
        unsigned pix0 = (unsigned)(prow * p.Wo + pcol);
        asm volatile("" : "+v"(pix0));                               // opaque: per-pixel offsets are recomputed, not hoisted into 32 registers per tensor
        const int64_t ipix = (int64_t)n * p.Ho * p.Wo;
        float* yimg = p.y + ipix * p.ldy;
        float* aimg = (!POST && p.act_out) ? p.act_out + ipix * p.ld_act : nullptr;
        const float* rimg = RES ? p.res + ipix * p.ld_res : nullptr;
        const float* gimg = POST ? p.gact + ipix * p.ld_gact : nullptr;
        float* y2img = (POST && p.y2) ? p.y2 + ipix * p.ld_y2 : nullptr;
        const unsigned chv = (unsigned)chan;
        if (INV32_PIPE) {
          auto between = [&](int b) {
            if constexpr (BURST == 8) {
              if (b == 0) inv_request_slice<0, 8>(nin, wave, half, c, B); else if (b == 1) inv_request_slice<8, 8>(nin, wave, half, c, B);
              else if (b == 2) inv_request_slice<16, 8>(nin, wave, half, c, B); else inv_request_slice<24, 8>(nin, wave, half, c, B);
            } else {
              if (b == 0) inv_request_slice<0, 4>(nin, wave, half, c, B); else if (b == 1) inv_request_slice<4, 4>(nin, wave, half, c, B);
              else if (b == 2) inv_request_slice<8, 4>(nin, wave, half, c, B); else if (b == 3) inv_request_slice<12, 4>(nin, wave, half, c, B);
              else if (b == 4) inv_request_slice<16, 4>(nin, wave, half, c, B); else if (b == 5) inv_request_slice<20, 4>(nin, wave, half, c, B);
              else if (b == 6) inv_request_slice<24, 4>(nin, wave, half, c, B); else inv_request_slice<28, 4>(nin, wave, half, c, B);
            }
          };
          if (rowok && !FFT_STUDY(1)) {
            epilogue_row<TANH, RES, POST, T, BURST, PCNN_ST32_NT != 0>(p, X, 1.f / 1024.f, vx, pix0, sgn, chv, bias, sc, sh, yimg, aimg, rimg, gimg, y2img, ymax, bsum, between,
                                                                       p.pack > 1 ? T : __builtin_amdgcn_readfirstlane(min(p.Vx, p.Wo - x0)));
          } else {
            inv_request(nin, wave, half, c, B);
          }
          piped = true;
        } else
        if (rowok && !FFT_STUDY(1))
          epilogue_row<TANH, RES, POST, T, BURST, PCNN_ST32_NT != 0>(p, X, 1.f / 1024.f, vx, pix0, sgn, chv, bias, sc, sh, yimg, aimg, rimg, gimg, y2img, ymax, bsum, NoBetween(),
                                                                     (PCNN_EPI_SKIP & 2) ? (p.pack > 1 ? T : __builtin_amdgcn_readfirstlane(min(p.Vx, p.Wo - x0))) : T);
      }
      if (INV32_PIPE && !piped) inv_request(nin, wave, half, c, B);    // waves without an output row in this tile
    }
    if (next >= total) break;
    item = __builtin_amdgcn_readfirstlane(next);
    lds_barrier();                                                       // the LDS image is free for the next item
  }
  if (POST && p.bsum) p.bsum[(blockIdx.x * 16 + wave) * 64 + lane] += bsum;   // own slot: launches of one call follow each other on the stream
  if (p.absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if (lane == 0) {
      const unsigned bits = __float_as_uint(ymax <= 3.0e38f ? ymax : 3.0e38f);
      if (bits > __atomic_load_n(p.absmax, __ATOMIC_RELAXED)) atomicMax(p.absmax, bits);
    }
  }
}

// (Measured and rejected, round 5: a software pipeline over the items - request(i + 1), x axis (i), barrier, y axis (i + 1), epilogue (i) - so that no load
// is issued behind the epilogue's stores (vmcnt is one in-order counter) and nothing is held through the epilogue.  Correct, but the 32 pixel values
// carried across the next item's y axis pushed every variant to 13-24 spilled registers, and scratch traffic shares the memory pipe: 7 taps forward
// 2.04 vs 1.86 ms at 8 x 1024^2.  The plain order - loads at the top of an item - stays.)
// (Measured and rejected, round 5 - commit "Experiment: 32-point FFT kernels on 16-channel items": the same kernels on 16-CHANNEL items, 8 waves and
// 64 KB of LDS per workgroup, TWO workgroups per CU, so that one computes while the other waits for its loads.  Correct - all 79 spectral tests - and
// 6 % slower: 7 taps forward 1.96 vs 1.84 ms, 11 taps 2.47 vs 2.16 ms at 8 x 1024^2 (profiles/r05_probe_xform_16ch_items_rejected.txt).  The 64-byte
// half-vector accesses cost more than the overlap brings; the 32-channel form below stays.)

// POST (fft32_inv_kernel): dbias[ch] from the lanes' partial sums, in a fixed order - thread t sums the slots (block, wave) = t, t + 256, ... of the
// lanes that carry the channel (lane & 31 = sub cpt + channel for each packed tile, both halves).  One workgroup per channel.
__global__ __launch_bounds__(256) void fft32_post_bias_kernel(const float* __restrict__ bsum, int nslots, int pack, int cpt, float* __restrict__ dbias) {
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

// =================================================================================================================== 64-point tiles
// A 64 x 64 x 32-channel tile is 512 KB - the whole register file of a CU.  item = (tile, 16 channels): the lanes of a wave are 4 x 16 channels
// (every global access a 64-byte half of a pixel's / spectrum row's channel vector; the two halves of a 32-channel group run at the same time on
// workgroups b and b + 8, i.e. on one XCD, so the second half of every 128-byte line comes from its L2), and the tile passes through LDS in TWO
// phases of 128 KB, one per parity of the x frequency:
//   x axis   wave w, lane group r: window row 4w + r, rfft_fwd<64> in 64 registers.  Its first step IS the parity split: the even bins are the real
//            FFT of e[n] = x[n] + x[n + 32] (-> phase 0), the odd bins X[4m + 1] = one complex FFT of 16 points (-> phase 1; X[4m + 3] = conj X[61 - 4m]).
//   y axis   per phase 16 columns x 64 rows x 16 channels in LDS, one column per wave; the lane group is the CLASS cl = fy mod 4 of the output
//            frequencies: Z[4m + cl] = FFT16( W64^(cl y) sum_q (-i)^(q cl) u[y + 16 q] )[m] - a radix-4 decimation-in-frequency step formed while reading
//            LDS, then cfft_dif<16> in 32 registers.  The two real columns (fx = 0, 32: phase 0) take one wave: lane groups (column, half) with
//            half E: the even fy as rfft_fwd<32>(u[y] + u[y + 32]), half D: the odd fy as the complex 16-point FFT of the real split.
constexpr int T64 = 64, ROWS64 = 4096;
// virtual work item v of the persistent grid -> (tile-and-group index, 16-channel half): v and v + 8 are the two halves of one 32-channel group
__device__ __forceinline__ void item64(int v, int& tg, int& hf) { tg = (v >> 4) * 8 + (v & 7); hf = (v >> 3) & 1; }

__host__ __device__ __forceinline__ int64_t sp_item64(int64_t item) { return pcnn_spec::sp_item(item, ROWS64); }


// what a wave keeps of an item between requesting its window rows and consuming them
struct FwdItem64 {
  const float* img;      // uniform: image n
  int cl, cr;            // uniform: window columns x < cl or x >= cr are constant padding
  int xlim;              // uniform: columns >= xlim are zero
  unsigned voff;         // lane = window column x: float offset of that column's source pixel in an image row (tf.pad index map applied)
};
struct FwdRow64 {
  float fill;            // lane: value of a row that is not read from memory (0: beyond ylim / no such channel; the padding constant)
  unsigned lo;           // lane: float offset of (source row, channel)
  bool use_fill;         // lane
  bool fetch;            // uniform: at least one of the unit's four rows is read from memory
};

// describes virtual item v (window, column map) and the unit's row y (lane: y = 4 unit + lane group)
template <int NU>
__device__ __forceinline__ void fwd64_describe_t(const FwdParams& p, int v, int lane, int c16, FwdItem64& it, FwdRow64 (&row)[2], int unit0) {
  int tg, hf;
  item64(v, tg, hf);
  const int g = tg % p.groups;
  int t = p.tile0 + tg / p.groups;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int cc = 16 * hf + c16, chan = g * p.cstride + cc;
  const bool cok = cc < p.cvalid && chan < p.C;
  it.img = p.x + (int64_t)n * p.H * p.W * p.ld;
  const int wy0 = ty * p.Vy - p.oy, wx0 = tx * p.Vx - p.ox;
  const int ylim = min(p.ylim, p.ext_y - ty * p.Vy);
  it.xlim = min(p.xlim, p.ext_x - tx * p.Vx);
  it.cl = 0; it.cr = T64;
  if (p.pad_mode == PCNN_PAD_CONSTANT) { it.cl = -wx0; it.cr = p.W - wx0; }
  // the column map is the same for every row and lane of the item: lane x forms entry x ONCE (vector selects), the loads read it back as a
  // scalar (v_readlane) and add it to the uniform image pointer - interior and boundary windows take the same path, one load + three scalar
  // instructions per pixel, and no index arithmetic per load
  it.voff = (unsigned)(pad_sel(wx0 + lane, p.W, p.pad_mode) * p.ld);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int y = 4 * (unit0 + u) + (lane >> 4);
    const int gy = wy0 + y;
    const bool rowconst = p.pad_mode == PCNN_PAD_CONSTANT && (unsigned)gy >= (unsigned)p.H;
    const bool rowzero = !cok || y >= ylim;
    row[u].use_fill = rowzero || rowconst;
    row[u].fill = rowzero ? 0.f : p.pad_value;
    row[u].fetch = 4 * (unit0 + u) < ylim;
    row[u].lo = (unsigned)(pad_sel(gy, p.H, p.pad_mode) * p.W * p.ld) + (unsigned)(cok ? chan : 0);
  }
}

// requests the window row of unit Un: always-valid addresses; padding and masks are applied on consumption
template <int Un>
__device__ __forceinline__ void fwd64_request(const FwdItem64& it, const FwdRow64 (&row)[2], float (&R)[2][64]) {
#pragma unroll
  for (int x = 0; x < T64; ++x) {
    const float* rp = it.img + (unsigned)__builtin_amdgcn_readlane((int)it.voff, x);
    R[Un][x] = (row[Un].fetch && !FFT_STUDY(2)) ? rp[row[Un].lo] : 0.f;                   // (defined on every path: otherwise R is carried around the item loop)
  }
}

template <bool MASKED>
__device__ __forceinline__ void fwd64_consume(const FwdParams& p, const FwdItem64& it, const FwdRow64& row, float (&R)[64]) {
#pragma unroll
  for (int x = 0; x < T64; ++x) {
    float v = R[x];
    if (x < it.cl || x >= it.cr) v = p.pad_value;
    if (row.use_fill) v = row.fill;
    if (MASKED && x >= it.xlim) v = 0.f;
    R[x] = v;
  }
}

// tw: this lane's row of the twiddle table in LDS, tw[2 y] + i tw[2 y + 1] = W64^(cl y) (gfx950 VOP3 selects take no literal operands: as a select
// chain over compile-time constants the 90 twiddles sit in 90 registers)
__device__ __forceinline__ void y64_gather(const float* U, int offr, int offi, int cl, float isign, const float* tw, float* vr, float* vi) {
  asm volatile("" : "+v"(offr), "+v"(offi));                          // opaque: LDS addresses are lane constants - left alone they are hoisted out of the
  const float* ur = U + offr, *ui = U + offi;                         // item loop, one register per 64 KB window and use
  const bool odd = cl & 1;
  const float sg2 = odd ? -1.f : 1.f;                                // a = u0 + sg2 u2, w = u1 + sg2 u3
  const float s = (cl & 2) ? -1.f : 1.f;                             // v = a + s b,  b = w (cl even) | -i w (cl odd; with s: cl = 1: -i w, cl = 3: +i w)
#pragma unroll
  for (int y = 0; y < 16; ++y) {
    const float u0r = ur[y * 512], u1r = ur[(y + 16) * 512], u2r = ur[(y + 32) * 512], u3r = ur[(y + 48) * 512];
    const float u0i = isign * ui[y * 512], u1i = isign * ui[(y + 16) * 512], u2i = isign * ui[(y + 32) * 512], u3i = isign * ui[(y + 48) * 512];
    const float ar = fma_(sg2, u2r, u0r), ai = fma_(sg2, u2i, u0i);
    const float wr_ = fma_(sg2, u3r, u1r), wi_ = fma_(sg2, u3i, u1i);
    const float br = odd ? wi_ : wr_, bi = odd ? -wr_ : wi_;
    const float xr = fma_(s, br, ar), xi = fma_(s, bi, ai);
    if (y == 0) { vr[0] = xr; vi[0] = xi; }
    else {
      const float tr = tw[2 * y], ti = tw[2 * y + 1];
      vr[y] = fma_(xr, tr, -(xi * ti));
      vi[y] = fma_(xr, ti, xi * tr);
    }
    if (y & 1) __builtin_amdgcn_sched_barrier(0);                    // batches of two rows: 16 LDS operands in flight, not 128
  }
}

// 8 waves of up to 256 registers, two units per wave and phase (unit u of wave w: window rows 4 (2w + u) + lane group, column 2w + u); an item's two
// rows of 64 values per lane are requested at the top of the item (no prefetch: see PCNN_PF64_NONE below).  (A 16-wave build of the same phases - 128
// registers - cannot hold a row of 64 beside the y-axis state: it spilled 50-110 registers in every arrangement tried.)
#ifndef PCNN_PF64_ONE
#define PCNN_PF64_ONE 0
#endif
#ifndef PCNN_PF64_FULL
#define PCNN_PF64_FULL 0
#endif
#ifndef PCNN_PF64_NONE
#define PCNN_PF64_NONE 1
#endif
constexpr bool PF64_NONE = PCNN_PF64_NONE; // 8-wave form WITHOUT any prefetch (both rows requested at the top of the item): the shipped form since the end of round 5 -
                                           // unit 0's row held across the even y phase (PCNN_PF64_NONE=0) costs 29 spilled registers, and the spill-free kernel is 2.4 %
                                           // faster per 15-tap forward although its loads are exposed (profiles/r05_probe_fwd64_no_prefetch.txt)
constexpr bool PF64_FULL = PCNN_PF64_FULL; // 8-wave form: request BOTH rows of the next item under the even y phase (128 registers in flight: spills ~60)
constexpr bool PF64_ONE = PCNN_PF64_ONE;   // 16-wave form: request the next item's row under the even y phase (64 registers in flight)
template <bool MASKED, int NU>
__device__ __forceinline__ void fft64_fwd_body(const FwdParams& p, const int nvirt, float* U, const int bid, const int gdim) {   // U[(y*32 + s)*16 + c16], y < 64, s < 32: one x-parity phase
  const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, c16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntg = p.ntile * p.groups;
  int v = bid;
  if (v >= nvirt) return;
  float* const TW = U + 64 * 32 * 16;                                // W64^(cl y), cl < 4, y < 16: 128 floats behind the phase image
  if (tid < 64) { TW[2 * tid] = cos64(((tid >> 4) * (tid & 15)) & 63); TW[2 * tid + 1] = -sin64(((tid >> 4) * (tid & 15)) & 63); }
  const float* const tw = TW + lg * 32;
  float R[2][64];
  FwdItem64 cur;
  FwdRow64 row[2];
  // the padded tail of the virtual item list (tg >= ntg) runs as a copy of a real item that stores nothing: the workgroup keeps its barriers
  auto safe = [&](int vv) { int tg, hf; item64(vv, tg, hf); return tg < ntg ? vv : (vv & 8); };
  fwd64_describe_t<NU>(p, safe(v), lane, c16, cur, row, NU * wave);
  if ((NU == 2 && !PF64_NONE) || PF64_ONE) fwd64_request<0>(cur, row, R);
  if (NU == 2 && PF64_FULL) fwd64_request<1>(cur, row, R);
  for (;;) {
    const int next = v + gdim;
    const bool more = next < nvirt;
    // (unit 0's row was requested a phase ahead - below; unit 1's here: the registers cannot hold both beside the y-axis state)
    if (NU == 2 && PF64_NONE) fwd64_request<0>(cur, row, R);
    if (NU == 2 && !PF64_FULL) fwd64_request<1>(cur, row, R);
    else if (NU == 1 && !PF64_ONE) fwd64_request<0>(cur, row, R);
    // ---- x axis: the whole real FFT of this lane's two rows.  The ODD bins go through LDS first (16 complex columns, every unit the same path)
    // while the even bins wait in R[u][0..32).
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      __builtin_amdgcn_sched_barrier(0);                                 // one unit after the other: interleaved, the two transforms need twice the registers
      fwd64_consume<MASKED>(p, cur, row[u], R[u]);
      if (!FFT_STUDY(4)) rfft_fwd<64>(R[u]);
      pin<32>(R[u]);
      int uoff = ((4 * (NU * wave + u) + lg) * 32) * 16 + c16;
      asm volatile("" : "+v"(uoff));
      float* urow = U + uoff;
#pragma unroll
      for (int m = 0; m < 16; ++m) {                                     // C[m] = X[4m + 1]: Re at s = m, Im at s = 16 + m
        if (FFT_STUDY(16)) continue;
        urow[m * 16] = R[u][32 + bitrev(m, 16)];
        urow[(16 + m) * 16] = R[u][48 + bitrev(m, 16)];
      }
    }
    lds_barrier();
    int tg, hf;
    item64(v, tg, hf);
    const bool store = tg < ntg && !FFT_STUDY(1);
    float* out = p.sp + sp_item64(store ? tg : 0) + 16 * hf;              // uniform; the lane adds (class row) * RS + c16
    int cl = lg;
    asm volatile("" : "+v"(cl));                                     // opaque: the per-lane constants of a phase are formed in the phase, not hoisted
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      // ---- y axis, odd fx.  column index q = 2 wave + u: fx = 2q + 1; q even: fx = 4j + 1 = C[j] (j = q / 2); q odd: fx = 4j + 3 = conj C[15 - j]
      __builtin_amdgcn_sched_barrier(0);
      const int q = NU * wave + u;
      const int mcol = (q & 1) ? 15 - (q >> 1) : (q >> 1);
      float V[32];
      float* vr = V, *vi = V + 16;
      if (FFT_STUDY(8)) { for (int i = 0; i < 32; ++i) V[i] = 1.f; } else y64_gather(U, mcol * 16 + c16, (16 + mcol) * 16 + c16, cl, (q & 1) ? -1.f : 1.f, tw, vr, vi);
      if (!FFT_STUDY(4)) cfft_dif<16, -1>(vr, vi);
      if (store) {
        float* o = out + (128 + 128 * (2 * q)) * RS;
        const unsigned lo = (unsigned)(sp_row_c(64, cl, 0) * RS + c16);     // row(4 m + cl) = row(4 m) + row(cl) for every block size (spectral_common.h)
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          NT_STORE(1, vr[bitrev(m, 16)], &(o + (sp_row_c(64, 4 * m, 0) * RS))[lo]);
          NT_STORE(1, vi[bitrev(m, 16)], &(o + (sp_row_c(64, 4 * m, 1) * RS))[lo]);
        }
      }
    }
    lds_barrier();                                                       // the odd phase has been read: U is free
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int uoff = ((4 * (NU * wave + u) + lg) * 32) * 16 + c16;
      asm volatile("" : "+v"(uoff));
      float* urow = U + uoff;
      float (&E)[32] = *reinterpret_cast<float (*)[32]>(&R[u][0]);
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) if (!FFT_STUDY(16)) urow[s2 * 16] = hc_get(E, s2);     // s <= 16: Re X[2s], s > 16: Im X[2 (s - 16)]
    }
    // R is free: unit 0's row of the next item is requested here and lands under the even y phase (both rows - 128 registers - beside the y-axis
    // state spill 66 registers; during the odd phase the even bins still occupy 64)
    if (more) {
      fwd64_describe_t<NU>(p, safe(next), lane, c16, cur, row, NU * wave);
      if ((NU == 2 && !PF64_NONE) || PF64_ONE) fwd64_request<0>(cur, row, R);
      if (NU == 2 && PF64_FULL) fwd64_request<1>(cur, row, R);
    }
    lds_barrier();
    asm volatile("" : "+v"(cl));
    // ---- y axis, even fx.  q = 0..14: complex column fx = 2 (q + 1) (Re at s = q + 1, Im at s = 17 + q); q = 15: the two real columns
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      const int q = NU * wave + u;
      float V[32];
      float* vr = V, *vi = V + 16;
      if (q < 15) {
        if (FFT_STUDY(8)) { for (int i = 0; i < 32; ++i) V[i] = 1.f; } else y64_gather(U, (q + 1) * 16 + c16, (17 + q) * 16 + c16, cl, 1.f, tw, vr, vi);
        if (!FFT_STUDY(4)) cfft_dif<16, -1>(vr, vi);
        if (store) {
          float* o = out + (128 + 128 * (2 * (q + 1) - 1)) * RS;
          const unsigned lo = (unsigned)(sp_row_c(64, cl, 0) * RS + c16);     // row(4 m + cl) = row(4 m) + row(cl) for every block size (spectral_common.h)
#pragma unroll
          for (int m = 0; m < 16; ++m) {
            NT_STORE(1, vr[bitrev(m, 16)], &(o + (sp_row_c(64, 4 * m, 0) * RS))[lo]);
            NT_STORE(1, vi[bitrev(m, 16)], &(o + (sp_row_c(64, 4 * m, 1) * RS))[lo]);
          }
        }
      } else {
        // lane groups: (column fx = 0, E), (0, D), (32, E), (32, D).  E: even fy = real FFT of u[y] + u[y + 32]; D: odd fy by the real split
        int uuoff = ((cl & 2) ? 16 : 0) * 16 + c16;
        asm volatile("" : "+v"(uuoff));
        const float* uu = U + uuoff;
        float* o = out;
        const unsigned lo = (unsigned)(((cl & 2) ? 64 : 0) * RS + c16);
        if ((cl & 1) == 0) {
#pragma unroll
          for (int yy = 0; yy < 32; ++yy) V[yy] = uu[yy * 512] + uu[(yy + 32) * 512];
          if (!FFT_STUDY(4)) rfft_fwd<32>(V);
          if (store) {
#pragma unroll
            for (int k = 0; k <= 16; ++k) NT_STORE(1, hc_get(V, k), &(o + (sp_row_r(64, 2 * k) * RS))[lo]);                 // Re Z[2k]: half-complex entry 2k of the column
#pragma unroll
            for (int k = 1; k < 16; ++k) NT_STORE(1, hc_get(V, 16 + k), &(o + (sp_row_r(64, 32 + 2 * k) * RS))[lo]);        // Im Z[2k]: entry 32 + 2k
          }
        } else {
          // d[n] = u[n] - u[n + 32]; c[n] = (d[n] - i d[n + 16]) W64^n, n < 16; C[m] = Z[4m + 1]; Z[4m + 3] = conj C[15 - m]
#pragma unroll
          for (int n2 = 0; n2 < 16; ++n2) {
            const float dr = uu[n2 * 512] - uu[(n2 + 32) * 512], di = uu[(n2 + 48) * 512] - uu[(n2 + 16) * 512];
            mul_tw<64, -1>(n2, dr, di, vr[n2], vi[n2]);
          }
          if (!FFT_STUDY(4)) cfft_dif<16, -1>(vr, vi);
          if (store) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
              NT_STORE(1, vr[bitrev(m, 16)], &(o + (sp_row_r(64, 4 * m + 1) * RS))[lo]);
              NT_STORE(1, vi[bitrev(m, 16)], &(o + (sp_row_r(64, 32 + 4 * m + 1) * RS))[lo]);
              NT_STORE(1, vr[bitrev(15 - m, 16)], &(o + (sp_row_r(64, 4 * m + 3) * RS))[lo]);
              NT_STORE(1, -vi[bitrev(15 - m, 16)], &(o + (sp_row_r(64, 32 + 4 * m + 3) * RS))[lo]);
            }
          }
        }
      }
    }
    if (!more) break;
    v = __builtin_amdgcn_readfirstlane(next);
    lds_barrier();                                                       // U is free for the next item
  }
}

template <bool MASKED, int NU>
__global__ __launch_bounds__(1024 / NU) void fft64_fwd_kernel(FwdParams p, int nvirt) {
  extern __shared__ __attribute__((aligned(16))) float U[];
  fft64_fwd_body<MASKED, NU>(p, nvirt, U, blockIdx.x, gridDim.x);
}

// the table form (fft32_fwd_multi_kernel): blockIdx.y = entry; gridDim.x is a multiple of 16 (a workgroup keeps one 16-channel half)
__global__ __launch_bounds__(512) void fft64_fwd_multi_kernel(const FwdParams* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) float U[];
  const FwdParams p = load_fwd_params(tab + blockIdx.y);
  const int ntg = p.ntile * p.groups;
  fft64_fwd_body<true, 2>(p, 2 * ((ntg + 7) & ~7), U, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------ 64-point inverse + epilogue
// The mirror image: item = (tile, 16 channels), 8 waves x 2 units.  Per x-parity phase (odd fx first, then even fx): unit q owns one column - lane
// group = class cl of the INPUT frequencies fy = 4m + cl: 32 spectrum loads per lane, G_cl[y] = conj(W64^(cl y)) IFFT16(Z[4m + cl])[y] into LDS (four
// class planes of 16 rows); then unit q owns output rows 4q .. 4q + 3 (lane group = row): u[y + 16 j] = sum_cl i^(j cl) G_cl[y] - the radix-4 decimation-
// in-time step, formed while reading LDS - lands in its place of the row's rfft_inv<64> input.  The two real columns fx = 0 / 32 are ONE complex column
// Z = A + i B (their Hermitian halves are read back from the half-complex rows), so every unit runs the same code.  After both phases: rfft_inv<64> of
// the row in registers and the fused convolution epilogue of spec64_inv_kernel.
constexpr int G_YS = 528, G_CS = 16 * G_YS + 16;                     // floats between rows / class planes of the LDS image (odd multiples of 16: no bank conflicts)
constexpr size_t LDS64I_BYTES = (4 * G_CS + 128) * sizeof(float);

#ifndef PCNN_ST64_NT
#define PCNN_ST64_NT 1
#endif
#define ST64(v, p) do { if (PCNN_ST64_NT) __builtin_nontemporal_store(v, p); else *(p) = (v); } while (0)
template <bool TANH, bool RES, bool POST>
__global__ __launch_bounds__(512) void fft64_inv_kernel(InvParams p, int nvirt) {
  extern __shared__ __attribute__((aligned(16))) float G[];          // G[cl * G_CS + y * G_YS + s * 16 + c16], y < 16, s < 32
  const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, c16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntg = p.ntile * p.groups;
  int v = blockIdx.x;
  if (v >= nvirt) return;
  float* const TW = G + 4 * G_CS;                                    // W64^(cl y), cl < 4, y < 16
  if (tid < 64) { TW[2 * tid] = cos64(((tid >> 4) * (tid & 15)) & 63); TW[2 * tid + 1] = -sin64(((tid >> 4) * (tid & 15)) & 63); }
  const float* const tw = TW + lg * 32;
  float ymax = 0.f, bsum = 0.f;
#ifndef PCNN_INV64_BURST
#define PCNN_INV64_BURST 8
#endif
  constexpr int BURST = PCNN_INV64_BURST;
  for (;;) {
    const int next = v + gridDim.x;
    int tg, hf;
    item64(v, tg, hf);
    const bool live = tg < ntg;                                      // the padded tail of the virtual item list: a copy of a real item that stores nothing
    const float* in = p.sp + sp_item64(live ? tg : 0) + 16 * hf;      // uniform
    float R[2][64];
    int cl = lg;
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {                          // 0: odd fx (C[m] = X[4m + 1] of the rows), 1: even fx
      lds_barrier();                                                     // the image of the previous phase / item has been read
      asm volatile("" : "+v"(cl));                                   // opaque: per-lane constants are formed in the phase, not hoisted out of the item loop
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        const int q = 2 * wave + u;                                    // column index of the phase
        float V[32];
        float* vr = V, *vi = V + 16;
        int sre;                                                       // s of the column's real part in the phase image (imaginary part: 16 + sre)
        if (phase == 1 && q == 15) {
          // the two real columns as one complex column Z = A + i B: A[fy] from rows [0, 64), B[fy] from rows [64, 128) (half-complex; fy > 32: the
          // conjugate of 64 - fy)
          sre = 0;
          const unsigned lo = (unsigned)c16;
#pragma unroll
          for (int m = 0; m < 16; ++m) {
            const int fy = 4 * m + cl, fp = fy <= 32 ? fy : 64 - fy;
            const float sg = fy <= 32 ? 1.f : -1.f;
            const bool hasim = fp != 0 && fp != 32;
            const unsigned rr = (unsigned)(sp_row_r(64, fp) * RS), ri = (unsigned)(sp_row_r(64, 32 + (hasim ? fp : 1)) * RS);     // (per-lane: fp depends on the class)
            const float ar = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &in[lo + rr]), br = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &(in + 64 * RS)[lo + rr]);
            float ai = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &in[lo + ri]), bi = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &(in + 64 * RS)[lo + ri]);
            ai = hasim ? sg * ai : 0.f; bi = hasim ? sg * bi : 0.f;
            vr[m] = ar - bi; vi[m] = ai + br;
          }
        } else {
          int fx;
          if (phase == 0) {                                            // q even: fx = 4j + 1 -> C[j]; q odd: fx = 4j + 3 = conj C[15 - j]
            fx = 2 * q + 1;
            sre = (q & 1) ? 15 - (q >> 1) : (q >> 1);
          } else { fx = 2 * (q + 1); sre = q + 1; }
          const float* src = in + (128 + 128 * (fx - 1)) * RS;       // uniform
          const unsigned lo = (unsigned)(sp_row_c(64, cl, 0) * RS + c16);
#pragma unroll
          for (int m = 0; m < 16; ++m) {
            vr[m] = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &(src + sp_row_c(64, 4 * m, 0) * RS)[lo]);
            vi[m] = FFT_STUDY(2) ? 1.f : NT_LOAD(8, &(src + sp_row_c(64, 4 * m, 1) * RS)[lo]);
          }
        }
        if (!FFT_STUDY(4)) cfft_dif<16, +1>(vr, vi);                                      // register j: y = bitrev(j)
        int goff = cl * G_CS + sre * 16 + c16;
        asm volatile("" : "+v"(goff));                                 // opaque: LDS addresses are lane constants - left alone, ~100 of them are hoisted
        float* g = G + goff;                                           // out of the item loop into registers of their own (everything else then spills)
#pragma unroll
        for (int y = 0; y < 16; ++y) {
          const float ar = vr[bitrev(y, 16)], ai = vi[bitrev(y, 16)];
          float gr, gi;
          if (y == 0) { gr = ar; gi = ai; }
          else {
            const float tr = tw[2 * y], ti = -tw[2 * y + 1];          // conj(W64^(cl y))
            gr = fma_(ar, tr, -(ai * ti));
            gi = fma_(ar, ti, ai * tr);
          }
          g[y * G_YS] = gr;
          g[y * G_YS + 256] = gi;                                      // s = 16 + sre
        }
      }
      lds_barrier();
      // ---- this unit's rows: u[y + 16 j] = sum_cl i^(j cl) G_cl[y], j = row >> 4 (uniform per unit), into the row's place of the x-axis inverse
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        const int unit = 2 * wave + u, j = unit >> 2;                  // rows 4 unit + lg = 16 j + y
        const int y = (4 * unit + lg) & 15;
        int goff = y * G_YS + c16;
        asm volatile("" : "+v"(goff));                                 // opaque (see above)
        const float* g0 = G + goff;
        // i^(j cl): cl = 1: (1, i, -1, -i)[j]; cl = 2: (-1)^j; cl = 3: (1, -i, -1, i)[j]
        const bool swp = j & 1;                                        // odd j: the cl = 1, 3 terms swap real and imaginary part
        const float s2 = (j & 1) ? -1.f : 1.f;
        const float s1r = (j == 0) ? 1.f : (j == 1 ? -1.f : (j == 2 ? -1.f : 1.f)), s1i = (j == 0) ? 1.f : (j == 1 ? 1.f : (j == 2 ? -1.f : -1.f));
        const float s3r = (j == 0) ? 1.f : (j == 1 ? 1.f : (j == 2 ? -1.f : -1.f)), s3i = (j == 0) ? 1.f : (j == 1 ? -1.f : (j == 2 ? -1.f : 1.f));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float a0 = g0[k * 16], b0 = g0[k * 16 + 256];
          const float a1 = g0[G_CS + k * 16], b1 = g0[G_CS + k * 16 + 256];
          const float a2 = g0[2 * G_CS + k * 16], b2 = g0[2 * G_CS + k * 16 + 256];
          const float a3 = g0[3 * G_CS + k * 16], b3 = g0[3 * G_CS + k * 16 + 256];
          // i^j (a1 + i b1): j = 1: (-b1, a1), j = 2: (-a1, -b1), j = 3: (b1, -a1);  (-i)^j (a3 + i b3): j = 1: (b3, -a3), j = 2: (-a3, -b3), j = 3: (-b3, a3)
          const float p1 = swp ? b1 : a1, q1 = swp ? a1 : b1, p3 = swp ? b3 : a3, q3 = swp ? a3 : b3;
          float re = fma_(s2, a2, a0), im = fma_(s2, b2, b0);
          re = fma_(s1r, p1, re); im = fma_(s1i, q1, im);
          re = fma_(s3r, p3, re); im = fma_(s3i, q3, im);
          asm volatile("" : "+v"(re), "+v"(im));                       // needed NOW: otherwise the sums are sunk to the x-axis transform and their 8 operands kept
          if (phase == 0) {                                            // C[k]: Re at 32 + bitrev(k), Im at 48 + bitrev(k) of the rfft_inv<64> input;
            R[u][32 + bitrev(k, 16)] = re;                             // k >= 8 came through the columns fx = 4j + 3 = conj C[15 - j]: conjugate the SUM
            R[u][48 + bitrev(k, 16)] = k >= 8 ? -im : im;              // (the class terms carry factors i^(j cl): conjugation does not commute with them)
          } else {                                                     // pair k: Re X[2k] | Im X[2k] (k = 1..15); k = 0: the real columns fx = 0 (re) and 32 (im)
            float (&E)[32] = *reinterpret_cast<float (*)[32]>(&R[u][0]);
            if (k == 0) { hc_put(E, 0, re); hc_put(E, 16, im); }
            else { hc_put(E, k, re); hc_put(E, 16 + k, im); }
          }
          if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- x axis inverse of this lane's two output rows + the fused epilogue (lane = channel: per-channel constants are per-lane scalars)
    {
      const int g = tg % p.groups;
      int t = p.tile0 + (live ? tg : 0) / p.groups;
      const int tx = t % p.tiles_x; t /= p.tiles_x;
      const int ty = t % p.tiles_y;
      const int n = t / p.tiles_y;
      const int y0 = ty * p.Vy, x0 = tx * p.Vx;
      const int vy = min(p.Vy, p.Ho - y0), vx = min(p.Vx, p.Wo - x0);
      const int cc = 16 * hf + c16, chan = g * p.cstride + cc;
      const bool cok = live && cc < p.cvalid && chan < p.C;
      const float bias = (p.bias && cok) ? p.bias[chan] : 0.f;
      const float sc = (p.bn_scale && cok) ? p.bn_scale[chan] : 1.f, sh = (p.bn_scale && cok) ? p.bn_shift[chan] : 0.f;
      const int sgn = p.flip ? -1 : 1;
      const int64_t ipix = (int64_t)n * p.Ho * p.Wo;
      float* yimg = p.y + ipix * p.ldy;
      float* aimg = (!POST && p.act_out) ? p.act_out + ipix * p.ld_act : nullptr;
      const float* rimg = RES ? p.res + ipix * p.ld_res : nullptr;
      const float* gimg = POST ? p.gact + ipix * p.ld_gact : nullptr;
      float* y2img = (POST && p.y2) ? p.y2 + ipix * p.ld_y2 : nullptr;
      const unsigned chv = (unsigned)chan;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        const int unit = 2 * wave + u;
        if (4 * unit >= vy) continue;                                  // uniform: none of the unit's four rows is an output row
        const int yy = 4 * unit + lg;
        if (!FFT_STUDY(4)) rfft_inv<64>(R[u]);                                            // R[u][x] = 4096 * pixel (yy, x)
        const int prow = p.flip ? p.Ho - 1 - y0 - yy : y0 + yy, pcol = p.flip ? p.Wo - 1 - x0 : x0;
        unsigned pix0 = (unsigned)(prow * p.Wo + pcol);
        asm volatile("" : "+v"(pix0));                               // opaque: per-pixel offsets are recomputed, not hoisted into registers per tensor
        if (cok && yy < vy && !FFT_STUDY(1))
          epilogue_row<TANH, RES, POST, T64, BURST, PCNN_ST64_NT != 0>(p, R[u], 1.f / 4096.f, vx, pix0, sgn, chv, bias, sc, sh, yimg, aimg, rimg, gimg, y2img, ymax, bsum, NoBetween(),
                                                                       (PCNN_EPI_SKIP & 4) ? __builtin_amdgcn_readfirstlane(vx) : T64);
      }
    }
    if (next >= nvirt) break;
    v = __builtin_amdgcn_readfirstlane(next);
  }
  if (POST && p.bsum) p.bsum[(blockIdx.x * 8 + wave) * 64 + lane] += bsum;    // own slot; a workgroup keeps ONE 16-channel half (the grid is a multiple of 16)
  if (p.absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    if (lane == 0) {
      const unsigned bits = __float_as_uint(ymax <= 3.0e38f ? ymax : 3.0e38f);
      if (bits > __atomic_load_n(p.absmax, __ATOMIC_RELAXED)) atomicMax(p.absmax, bits);
    }
  }
}

// POST at 64 points: dbias[ch] from the lanes' partial sums, in a fixed order.  Channel ch = 16 hf + c16 lives in the workgroups with (block >> 3) & 1 == hf,
// lanes with lane & 15 == c16 (four lane groups = four rows).  One workgroup per channel.
__global__ __launch_bounds__(256) void fft64_post_bias_kernel(const float* __restrict__ bsum, int nblocks, float* __restrict__ dbias) {
  __shared__ float red[256];
  const int ch = blockIdx.x, t = threadIdx.x, hf = ch >> 4, c16 = ch & 15;
  float a = 0.f;
  for (int sl = t; sl < nblocks * 8 * 4; sl += 256) {                 // (block, wave, lane group)
    const int lgq = sl & 3, bw = sl >> 2, block = bw >> 3;
    if (((block >> 3) & 1) == hf) a += bsum[(size_t)bw * 64 + lgq * 16 + c16];
  }
  red[t] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (t < st) red[t] += red[t + st]; __syncthreads(); }
  if (t == 0) dbias[ch] = red[0];
}

constexpr size_t LDS64_BYTES = LDS_BYTES + 128 * sizeof(float);       // + the twiddle table of the radix-4 step
template <typename K>
void set_lds(K kernel, size_t bytes = LDS_BYTES) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); }

template <bool TANH, bool RES, bool POST>
void launch_inv_t(pcnn_handle h, const InvParams& p, const dim3& grid) {
  set_lds(fft32_inv_kernel<TANH, RES, POST>);
  hipLaunchKernelGGL((fft32_inv_kernel<TANH, RES, POST>), grid, dim3(1024), LDS_BYTES, h->stream, p);
}

}  // namespace

// persistent kernels: one 16-wave workgroup per CU (128 KB of LDS) walking the (tile, channel group) items
#ifdef PCNN_FFT_STUDY
static void study_init() { static int once = 0; if (!once) { once = 1; const int v = getenv("PCNN_FFT_STUDY") ? atoi(getenv("PCNN_FFT_STUDY")) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fft_study), &v, sizeof(int)); } }
#else
static void study_init() {}
#endif

void launch_fwd_fft32(pcnn_handle h, FwdParams p, int ntile) {
  study_init();
  p.ntile = ntile;
  const dim3 grid((unsigned)std::min(ntile * p.groups, 256));
  if (p.ylim < T || p.xlim < T) {
    set_lds(fft32_fwd_kernel<true>);
    hipLaunchKernelGGL((fft32_fwd_kernel<true>), grid, dim3(1024), LDS_BYTES, h->stream, p);
  } else {
    set_lds(fft32_fwd_kernel<false>);
    hipLaunchKernelGGL((fft32_fwd_kernel<false>), grid, dim3(1024), LDS_BYTES, h->stream, p);
  }
}

// `count` filters (table entries in device memory, ntile = 1 each) in one launch; max_items: the largest groups value among them
void launch_fwd_fft32_multi(pcnn_handle h, const FwdParams* tab, int count, int max_items) {
  set_lds(fft32_fwd_multi_kernel);
  hipLaunchKernelGGL(fft32_fwd_multi_kernel, dim3((unsigned)std::min(max_items, 64), (unsigned)count), dim3(1024), LDS_BYTES, h->stream, tab);
}
void launch_fwd_fft64_multi(pcnn_handle h, const FwdParams* tab, int count, int max_items) {
  set_lds(fft64_fwd_multi_kernel, LDS64_BYTES);
  const int nvirt = 2 * ((max_items + 7) & ~7);
  hipLaunchKernelGGL(fft64_fwd_multi_kernel, dim3((unsigned)std::min((nvirt + 15) & ~15, 64), (unsigned)count), dim3(512), LDS64_BYTES, h->stream, tab);
}

// POST partial sums of the 32-point FFT inverse -> dbias
void launch_post_bias_fft32(pcnn_handle h, const float* bsum, int pack, int cpt, int C, float* dbias) {
  hipLaunchKernelGGL(fft32_post_bias_kernel, dim3((unsigned)C), dim3(256), 0, h->stream, bsum, 256 * FFT_WAVES, pack, cpt, dbias);
}

void launch_fwd_fft64(pcnn_handle h, FwdParams p, int ntile) {
  p.ntile = ntile;
  study_init();
  const int ntg = ntile * p.groups;
  const int nvirt = 2 * ((ntg + 7) & ~7);                            // (tile-and-group) x two 16-channel halves, in blocks of 8 + 8 (item64)
  const dim3 grid((unsigned)std::min((nvirt + 15) & ~15, 256));
  const bool masked = p.ylim < T64 || p.xlim < T64 || p.ext_y < (1 << 29) || p.ext_x < (1 << 29);
  static const int nu = getenv("PCNN_FFT64_UNITS") ? atoi(getenv("PCNN_FFT64_UNITS")) : 2;    // developer switch (A/B): 1 = 16 waves x 1 unit, 2 = 8 waves x 2 units
  if (nu == 1) {
    if (masked) { set_lds(fft64_fwd_kernel<true, 1>, LDS64_BYTES); hipLaunchKernelGGL((fft64_fwd_kernel<true, 1>), grid, dim3(1024), LDS64_BYTES, h->stream, p, nvirt); }
    else { set_lds(fft64_fwd_kernel<false, 1>, LDS64_BYTES); hipLaunchKernelGGL((fft64_fwd_kernel<false, 1>), grid, dim3(1024), LDS64_BYTES, h->stream, p, nvirt); }
    return;
  }
  if (masked) {
    set_lds(fft64_fwd_kernel<true, 2>, LDS64_BYTES);
    hipLaunchKernelGGL((fft64_fwd_kernel<true, 2>), grid, dim3(512), LDS64_BYTES, h->stream, p, nvirt);
  } else {
    set_lds(fft64_fwd_kernel<false, 2>, LDS64_BYTES);
    hipLaunchKernelGGL((fft64_fwd_kernel<false, 2>), grid, dim3(512), LDS64_BYTES, h->stream, p, nvirt);
  }
}

template <bool TANH, bool RES, bool POST>
static void launch_inv64_t(pcnn_handle h, const InvParams& p, const dim3& grid, int nvirt) {
  set_lds(fft64_inv_kernel<TANH, RES, POST>, LDS64I_BYTES);
  hipLaunchKernelGGL((fft64_inv_kernel<TANH, RES, POST>), grid, dim3(512), LDS64I_BYTES, h->stream, p, nvirt);
}

void launch_inv_fft64(pcnn_handle h, InvParams p, int ntile) {
  study_init();
  p.ntile = ntile;
  const int ntg = ntile * p.groups;
  const int nvirt = 2 * ((ntg + 7) & ~7);
  const dim3 grid((unsigned)std::min((nvirt + 15) & ~15, 256));
  if (p.gact) {
    p.alpha = 1.f;
    p.galpha = p.gmode == PCNN_ACT_LINEAR ? 1.f : (p.gmode == PCNN_ACT_RELU ? 0.f : p.galpha);
    if (p.res) launch_inv64_t<false, true, true>(h, p, grid, nvirt); else launch_inv64_t<false, false, true>(h, p, grid, nvirt);
    return;
  }
  if (p.act == PCNN_ACT_TANH) {
    if (p.res) launch_inv64_t<true, true, false>(h, p, grid, nvirt); else launch_inv64_t<true, false, false>(h, p, grid, nvirt);
  } else {
    p.alpha = p.act == PCNN_ACT_LINEAR ? 1.f : (p.act == PCNN_ACT_RELU ? 0.f : p.alpha);
    if (p.res) launch_inv64_t<false, true, false>(h, p, grid, nvirt); else launch_inv64_t<false, false, false>(h, p, grid, nvirt);
  }
}

void launch_post_bias_fft64(pcnn_handle h, const float* bsum, int nblocks, int C, float* dbias) {
  hipLaunchKernelGGL(fft64_post_bias_kernel, dim3((unsigned)C), dim3(256), 0, h->stream, bsum, nblocks, dbias);
}

void launch_inv_fft32(pcnn_handle h, InvParams p, int ntile) {
  study_init();
  p.ntile = ntile;
  const dim3 grid((unsigned)std::min(ntile * p.groups, 256));
  if (p.gact) {                                                      // data gradient + the producer's activation backward (linear conv epilogue)
    p.alpha = 1.f;
    p.galpha = p.gmode == PCNN_ACT_LINEAR ? 1.f : (p.gmode == PCNN_ACT_RELU ? 0.f : p.galpha);
    if (p.res) launch_inv_t<false, true, true>(h, p, grid); else launch_inv_t<false, false, true>(h, p, grid);
    return;
  }
  if (p.act == PCNN_ACT_TANH) {
    if (p.res) launch_inv_t<true, true, false>(h, p, grid); else launch_inv_t<true, false, false>(h, p, grid);
  } else {
    p.alpha = p.act == PCNN_ACT_LINEAR ? 1.f : (p.act == PCNN_ACT_RELU ? 0.f : p.alpha);     // slope of the negative side
    if (p.res) launch_inv_t<false, true, false>(h, p, grid); else launch_inv_t<false, false, false>(h, p, grid);
  }
}

}  // namespace pcnn_spec
