// "3 x fp16 split" variant of the fused pad+conv kernel: fp32-accurate products on the fp16 matrix cores.
//
// Every fp32 operand v (scaled by a power of two so that |v| < 2^13) is written as hi + lo with hi = fp16(v), lo = fp16(v - hi)
// (22 significant bits).  a*b is accumulated in fp32 as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (the dropped a_lo*b_lo term is
// <= 2^-22 relative): three v_mfma_f32_32x32x16_f16 per 16-deep K step instead of eight v_mfma_f32_32x32x2_f32 of twice the
// issue time each, i.e. 5.3x fewer matrix-pipe cycles.  Measured against fp64 on K = 7200 dot products
// (tools/split_f16_test.hip, MI355X): rel-L2 5.4e-7 for the split vs 1.5e-6 for the plain fp32 MFMA fmaf chain - the fp16
// MFMA sums its 16 products more accurately than 16 sequential fp32 fmas - also on data spanning e^(+-8) in magnitude.
//
// Mapping (differences from conv_fwd.hip):
//   * the halo tile is staged per 8-channel chunk as TWO fp16 planes (hi, lo), 16 bytes per pixel each: pixel stride 16 B makes
//     the ds_read_b128 fragment reads conflict-free without padding;
//   * scaling is per (tile, chunk): the loader keeps the chunk's raw fp32 values in registers, reduces max|x| over the
//     workgroup and scales by the power of two that puts the maximum into [2^12, 2^13); the chunk's accumulator is rescaled
//     when it is added to the running total, so activations of any magnitude are safe from fp16 overflow/underflow;
//   * one MFMA K-step (16) = 2 filter taps x 8 channels: lanes 0-31 (k = 0..7) read tap 2t, lanes 32-63 tap 2t+1;
//   * the filter is packed (per call) into matching fp16 hi/lo fragments, scaled by a per-tensor power of two.
#include "pcnn_internal.h"
#include "conv_epilogue.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TW = 32, WAVES = 4;   // tile = (WAVES*MT) rows x 32 columns; MT = 4 (16 rows), or 2 for layers with little work per tile
// halo pixels per thread per chunk: MT = 4: (16+14)*(32+14)/256 = 5.4 for k = 15; MT = 2 (k <= 5): (8+4)*(32+4)/256 = 1.7
constexpr int max_pix(int MT) { return MT == 2 ? 2 : 6; }

struct SplitParams {
  const float* x; const float* bias; const float* bn_scale; const float* bn_shift; const float* res; float* y; float* act_out;
  const f16x8* wp;            // packed filter: [chunk][tap pair][half][co][8 fp16], hi plane then lo plane
  const float* wscale;        // wscale[0] = 1 / s_w
  int64_t wplane;             // f16x8 elements per plane
  int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, kh, kw, pt, pl, pad_mode; float pad_value; int act; float alpha;
  int ld_res, ld_act, tiles_x, tiles_y, vec_ok, epi_vec;
  unsigned* y_absmax;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// MT = 2 halves the accumulator registers: 4 workgroups per CU instead of 2 hide the load -> convert -> MFMA -> store latency chain
// of the small layers (3x3 / 5x5, <= 20 channels), whose tiles hold only a few MFMA steps
template <int NT, int MT>
__global__ __launch_bounds__(256, NT == 1 ? (MT == 2 ? 4 : 2) : 1) void conv_fwd_split_kernel(SplitParams p) {
  constexpr int TH = WAVES * MT, MAXPIX = max_pix(MT);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
  // XCD-aware order: workgroup ids go round-robin to the 8 XCDs, so XCD x takes the x-th contiguous eighth of the tiles -
  // neighbouring tiles (shared halo rows and columns) meet in one L2
  int tile;
  {
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;
    tile = xcd * per + min(xcd, rem) + (blockIdx.x >> 3);
  }
  const int tx = tile % p.tiles_x; tile /= p.tiles_x;
  const int ty = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int TR = TH + p.kh - 1, TC = TW + p.kw - 1, npix = TR * TC;
  f16x8* hi_plane = reinterpret_cast<f16x8*>(smem);
  f16x8* lo_plane = hi_plane + npix;
  float* red = reinterpret_cast<float*>(lo_plane + npix);      // 4 floats: per-wave maxima
  const int cin_pad = (p.Cin + 7) & ~7;
  const int T = p.kh * p.kw, nT2 = (T + 1) >> 1;
  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
  const float inv_sw = p.wscale[0];

  // two-level summation: `acc` sums one 8-channel chunk (in units of that chunk's activation scale), `tot` the chunks
  f32x16 acc[MT][NT], tot[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[m][t][i] = 0.f; tot[m][t][i] = 0.f; }

  // source pixel of every halo slot of this thread (boundary-condition index maps applied), computed once: the same for all chunks.
  // -1: outside the image under CONSTANT padding; -2: slot beyond the tile
  int soff[MAXPIX];
#pragma unroll
  for (int q = 0; q < MAXPIX; ++q) {
    const int u = tid + 256 * q;
    const int uu = u < npix ? u : 0;
    const int r = uu / TC, c = uu - r * TC;
    const int sy = pcnn_pad_index(y0 + r - p.pt, p.H, p.pad_mode);
    const int sx = pcnn_pad_index(x0 + c - p.pl, p.W, p.pad_mode);
    soff[q] = u >= npix ? -2 : ((sy >= 0 && sx >= 0) ? sy * p.W + sx : -1);
  }

  for (int c0 = 0; c0 < cin_pad; c0 += 8) {
    // ---- load this chunk's halo pixels (8 channels each) into registers, padding applied
    f32x4 va[MAXPIX], vb[MAXPIX];
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < MAXPIX; ++q) {
      const bool inimg = soff[q] >= 0;
      const float* src = xin + (int64_t)(inimg ? soff[q] : 0) * p.ldx;
      f32x4 a, b;
      if (p.vec_ok && c0 + 7 < p.Cin) {
        a = *reinterpret_cast<const f32x4*>(src + c0);
        b = *reinterpret_cast<const f32x4*>(src + c0 + 4);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = src[c0 + j < p.Cin ? c0 + j : p.Cin - 1];
          b[j] = src[c0 + 4 + j < p.Cin ? c0 + 4 + j : p.Cin - 1];
        }
      }
      if (!inimg) { const float pv = p.pad_value; a = (f32x4){pv, pv, pv, pv}; b = a; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c0 + j >= p.Cin) a[j] = 0.f;
        if (c0 + 4 + j >= p.Cin) b[j] = 0.f;
      }
      if (soff[q] == -2) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }
      va[q] = a; vb[q] = b;
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fmaxf(fabsf(a[j]), fabsf(b[j])));
    }
    mx = wave_max(mx);
    __syncthreads();                       // previous chunk's fragment reads are done (LDS planes and `red` are free)
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (mx > 0.f) (void)frexpf(mx, &e);    // mx = m * 2^e, m in [0.5, 1)
    const bool okx = mx > 0.f && e > -100 && e < 100;
    const float s = okx ? ldexpf(1.0f, 13 - e) : 1.0f, inv_s = okx ? ldexpf(1.0f, e - 13) : 1.0f;
#pragma unroll
    for (int q = 0; q < MAXPIX; ++q) {
      const int u = tid + 256 * q;
      if (u < npix) {
        f16x8 h8, l8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = va[q][j] * s, b = vb[q][j] * s;
          const _Float16 ah = (_Float16)a, bh = (_Float16)b;
          h8[j] = ah; l8[j] = (_Float16)(a - (float)ah);
          h8[4 + j] = bh; l8[4 + j] = (_Float16)(b - (float)bh);
        }
        hi_plane[u] = h8; lo_plane[u] = l8;
      }
    }
    __syncthreads();

    // ---- K loop: one step = 2 taps x 8 channels; B fragments (hi, lo) prefetched one step ahead (ping-pong)
    const f16x8* wl = p.wp + ((int64_t)(c0 >> 3) * nT2 * 2 + half) * (NT * 32) + col;
    auto load_b = [&](f16x8 (&bh)[NT], f16x8 (&bl)[NT], int t2) {
      const f16x8* src = wl + (int64_t)t2 * 2 * (NT * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) { bh[t] = src[t * 32]; bl[t] = src[p.wplane + t * 32]; }
    };
    const int lane_pix = (wave * MT) * TC + col;       // pixel index of this lane's A row for tap (0,0), M-tile 0
    // A fragments: the hi halves of step t+1 are requested (LDS) before the MFMAs of step t, the lo halves at the start of
    // their own step (first needed by the third MFMA); B fragments (L2) of step t+1 before the MFMAs of step t.  Registers:
    // acc + tot (128) leave room for one extra hi set only.
    int ki0 = 0, kj0 = 0;                              // first tap of the step whose hi fragments are loaded next
    auto tap_offset = [&]() {
      if (ki0 >= p.kh) { ki0 = 0; kj0 = 0; }           // prefetch past the last step: any valid offset, the values are unused
      int ki1 = ki0, kj1 = kj0 + 1;
      if (kj1 == p.kw) { kj1 = 0; ki1 = ki0 + 1; }
      if (ki1 >= p.kh) { ki1 = ki0; kj1 = kj0; }       // odd tap count: the pad tap reads a valid pixel (its filter is zero)
      const int off = half ? ki1 * TC + kj1 : ki0 * TC + kj0;
      kj0 += 2;
      while (kj0 >= p.kw) { kj0 -= p.kw; ki0 += 1; }
      return lane_pix + off;
    };
    auto step = [&](const f16x8 (&bh)[NT], const f16x8 (&bl)[NT], const f16x8 (&ah)[MT], int a_cur,
                    f16x8 (&nbh)[NT], f16x8 (&nbl)[NT], f16x8 (&nah)[MT], int& a_next, int t2) {
      load_b(nbh, nbl, t2 + 1);                        // unconditional: the packed filter has spare zero steps past the end
      f16x8 al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) al[m] = lo_plane[a_cur + m * TC];
      a_next = tap_offset();
#pragma unroll
      for (int m = 0; m < MT; ++m) nah[m] = hi_plane[a_next + m * TC];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[t], acc[m][t], 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[t], acc[m][t], 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[t], acc[m][t], 0, 0, 0);
        }
    };
    f16x8 b0h[NT], b0l[NT], b1h[NT], b1l[NT], a0h[MT], a1h[MT];
    int a0 = tap_offset(), a1 = 0;
    load_b(b0h, b0l, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) a0h[m] = hi_plane[a0 + m * TC];
    int t2 = 0;
    for (; t2 + 2 <= nT2; t2 += 2) {
      step(b0h, b0l, a0h, a0, b1h, b1l, a1h, a1, t2);
      step(b1h, b1l, a1h, a1, b0h, b0l, a0h, a0, t2 + 1);
    }
    if (t2 < nT2) step(b0h, b0l, a0h, a0, b1h, b1l, a1h, a1, t2);
    // ---- fold the chunk into the running total with its scale
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { tot[m][t][i] = fmaf(acc[m][t][i], inv_s, tot[m][t][i]); acc[m][t][i] = 0.f; }
  }

  // ---- epilogue (conv_epilogue.h): staged through the LDS the planes no longer need
  ConvEpilogue e;
  e.bias = p.bias; e.bn_scale = p.bn_scale; e.bn_shift = p.bn_shift; e.res = p.res; e.y = p.y; e.act_out = p.act_out;
  e.Ho = p.Ho; e.Wo = p.Wo; e.Cout = p.Cout; e.ldy = p.ldy; e.ld_res = p.ld_res; e.ld_act = p.ld_act; e.act = p.act; e.alpha = p.alpha;
  e.vec = p.epi_vec; e.absmax = p.y_absmax;
  conv_epilogue_store<MT, NT>(tot, inv_sw, e, n, y0, x0, reinterpret_cast<float*>(smem));
}

// per-tensor power-of-two scale of the filter: out[0] = 1/s, out[1] = s with max|w|*s in [2^12, 2^13)
__global__ __launch_bounds__(1024) void filter_scale_kernel(const float* __restrict__ w, int64_t n, float* __restrict__ out) {
  __shared__ float red[16];
  float mx = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, fabsf(w[i]));
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
    int e = 0;
    if (mx > 0.f) (void)frexpf(mx, &e);
    out[0] = mx > 0.f ? ldexpf(1.0f, e - 13) : 1.0f;
    out[1] = mx > 0.f ? ldexpf(1.0f, 13 - e) : 1.0f;
  }
}

// w (kh,kw,Cin,Cout) fp32 -> packed fp16 fragments [chunk g][tap pair t2][half h][co (NT*32)][8], hi plane then lo plane:
//   element j of (g, t2, h, co) = w[tap 2*t2 + h][8g + j][co] * s   (zero for taps / channels / columns outside the filter)
__global__ void pack_split_kernel(const float* __restrict__ w, const float* __restrict__ scale, _Float16* __restrict__ wp, int64_t plane_halfs, int T,
                                  int Cin, int Cout, int ng, int nT2, int NT) {
  const int64_t total = (int64_t)ng * nT2 * 2 * NT * 32 * 8;
  const float s = scale[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = i & 7; int64_t r = i >> 3; const int co = r % (NT * 32); r /= (NT * 32); const int h = r & 1; r >>= 1;
    const int t2 = r % nT2; const int g = r / nT2;
    const int tap = 2 * t2 + h, ci = 8 * g + j;
    float v = 0.f;
    if (tap < T && ci < Cin && co < Cout) v = w[((int64_t)tap * Cin + ci) * Cout + co] * s;
    const _Float16 hi = (_Float16)v;
    wp[i] = hi;
    wp[plane_halfs + i] = (_Float16)(v - (float)hi);
  }
}

}  // namespace

// Returns 0 on success, -1 if the shape is outside what this path covers (caller falls back to the fp32-MFMA kernel).
int pcnn_conv2d_fwd_split(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                          const float* bn_shift, const float* residual, float* y, float* act_out) {
  if (d->kh > 15 || d->kw > 15 || d->Cout > 64) return -1;
  const int NT = d->Cout <= 32 ? 1 : 2;
  const int cin_pad = (d->Cin + 7) & ~7, ng = cin_pad >> 3, T = d->kh * d->kw, nT2 = (T + 1) >> 1;
  static const int mt2_max_k = getenv("PCNN_SPLIT_MT2_MAXK") ? atoi(getenv("PCNN_SPLIT_MT2_MAXK")) : 640;
  const int MTsel = (NT == 1 && T * cin_pad <= mt2_max_k && (2 * WAVES + d->kh - 1) * (TW + d->kw - 1) <= 256 * max_pix(2)) ? 2 : 4;       // 8-row tiles for layers with few MFMA steps per tile
  const int TH = WAVES * MTsel;
  const int TR = TH + d->kh - 1, TC = TW + d->kw - 1;
  if (TR * TC > 256 * max_pix(MTsel)) return -1;
  const int64_t plane_halfs = ((int64_t)ng * nT2 + 4) * 2 * NT * 32 * 8;       // + spare steps for the prefetch past the end
  const size_t need = 256 + (size_t)plane_halfs * 2 * sizeof(_Float16);
  if (h->scratch_bytes < need) {
    if (h->scratch) { pcnn_release(h, h->scratch); h->scratch = nullptr; h->scratch_bytes = 0; }
    const size_t cap = need < (4u << 20) ? (4u << 20) : need;
    if (hipMalloc(&h->scratch, cap) != hipSuccess) PCNN_FAIL(h, "pcnn_conv2d_fwd: cannot allocate %zu B of filter scratch", cap);
    h->scratch_bytes = cap;
  }
  float* scale = static_cast<float*>(h->scratch);
  _Float16* wp = reinterpret_cast<_Float16*>(static_cast<unsigned char*>(h->scratch) + 256);
  const int64_t nw = (int64_t)T * d->Cin * d->Cout;
  hipLaunchKernelGGL(filter_scale_kernel, dim3(1), dim3(1024), 0, h->stream, w, nw, scale);
  const int64_t total = (int64_t)ng * nT2 * 2 * NT * 32 * 8;
  hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(total, 256), 2048)), dim3(256), 0, h->stream, w, scale, wp, plane_halfs, T,
                     d->Cin, d->Cout, ng, nT2, NT);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_fwd(split pack)");
  SplitParams p;
  p.x = x; p.bias = bias; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.res = residual; p.y = y; p.act_out = act_out;
  p.wp = reinterpret_cast<const f16x8*>(wp); p.wscale = scale; p.wplane = plane_halfs / 8;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.ldy = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.act = d->act; p.alpha = d->act_alpha; p.ld_res = d->ld_res; p.ld_act = d->ld_act_out;
  p.tiles_x = pcnn_cdiv(d->Wo, TW); p.tiles_y = pcnn_cdiv(d->Ho, TH);
  p.vec_ok = (d->ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.epi_vec = conv_epilogue_vec_ok(d->Cout, y, d->ldy, residual, d->ld_res, act_out, d->ld_act_out);
  p.y_absmax = reinterpret_cast<unsigned*>(h->y_absmax);
  const size_t lds = std::max((size_t)TR * TC * 32 + 64, conv_epilogue_lds_bytes(d->Cout));
  const int64_t nblk = (int64_t)d->N * p.tiles_x * p.tiles_y;
  if (nblk >= (1ll << 31)) return -1;
  if (NT == 1 && MTsel == 2) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_split_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_fwd_split_kernel<1, 2>), dim3((unsigned)nblk), dim3(256), lds, h->stream, p);
  } else if (NT == 1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_split_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_fwd_split_kernel<1, 4>), dim3((unsigned)nblk), dim3(256), lds, h->stream, p);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_split_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_fwd_split_kernel<2, 4>), dim3((unsigned)nblk), dim3(256), lds, h->stream, p);
  }
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_fwd(split)");
  return 0;
}
