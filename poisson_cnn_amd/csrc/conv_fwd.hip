// Fused BC-padding + 2-D convolution + bias + activation (+BN affine)(+residual) as an fp32 implicit GEMM on
// the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
//
// Mapping (MI355X-first, not a cuDNN-style im2col):
//   * workgroup = 256 threads = 4 waves, output tile 16 rows x 32 columns of one image, all Cout (<= 64).
//   * GEMM M = 32 consecutive output pixels of one row (one MFMA tile), N = Cout, K = (tap, Cin chunk).
//     Wave w owns rows 4w..4w+3 of the tile: 4 accumulators of 32x32 (64 VGPRs).
//   * The input halo tile (16+kh-1) x (32+kw-1) x CK channels is staged ONCE per Cin chunk in LDS with the
//     boundary-condition padding (CONSTANT / SYMMETRIC / REFLECT, tf.pad semantics) applied by the loader, so the
//     padded tensor never exists in HBM.  Pixel stride PS = 4*odd floats makes the ds_read_b128 A-fragment reads
//     conflict free (lane = pixel, 4 consecutive channels per lane feed 4 MFMAs).
//   * Filter fragments (B) are read straight from global/L2 in the Keras HWIO layout: for one (tap, ci) the 32 (co)
//     values are 128 contiguous bytes, so the B operand needs no LDS and no repacking; they are prefetched one
//     K-step ahead in registers.
//   * Epilogue from the accumulators: bias, activation, optional BN affine, optional residual, NHWC store with an
//     arbitrary channel stride (so a conv can write straight into a channel slice of a concat buffer).
#include "pcnn_internal.h"
#include "conv_epilogue.h"

int pcnn_conv2d_fwd_split(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                          const float* bn_shift, const float* residual, float* y, float* act_out);   // conv_fwd_split.hip
int pcnn_spectral_conv_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                           const float* bn_shift, const float* residual, float* y, float* act_out);  // spectral_conv.hip
bool pcnn_spectral_eligible(pcnn_handle h, const pcnn_conv_desc* d, bool wgrad);
int pcnn_conv_small_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                        const float* bn_shift, const float* residual, float* y, float* act_out);              // conv_small.hip
bool pcnn_conv_fwd_takes_narrow_route(pcnn_handle h, const pcnn_conv_desc* d);                            // spectral_conv.hip

namespace {

constexpr int TH = 16, TW = 32, WAVES = 4, MT = TH / WAVES;
constexpr int MAX_LDS_BYTES = 80 * 1024;   // two workgroups per CU (160 KiB LDS)

struct ConvParams {
  const float* x; const float* w; const float* bias; const float* bn_scale; const float* bn_shift; const float* res;
  float* y; float* act_out;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, kh, kw, pt, pl, pad_mode; float pad_value; int act; float alpha;
  int ld_res, ld_act;
  int tiles_x, tiles_y, CK, PS, vec_ok, epi_vec;
  unsigned* y_absmax;
  const float* wp;   // packed filter, see pack_weights_kernel
};

template <int NT>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void conv_fwd_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, col = lane & 31;
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
  const int TR = TH + p.kh - 1, TC = TW + p.kw - 1;
  const int PS = p.PS;
  const int cin_pad = (p.Cin + 7) & ~7;

  // Two-level (blocked) summation: `acc` collects one filter row of one Cin chunk (K <= kw*CK terms) inside the MFMA
  // fmaf chain, then is added into `tot`.  This cuts the fp32 rounding error of the K ~ 7200 reductions by ~3x compared
  // with one long chain, at the cost of 64 v_add per ~240 MFMAs.
  f32x16 acc[MT][NT], tot[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[m][t][i] = 0.f; tot[m][t][i] = 0.f; }

  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;

  for (int c0 = 0; c0 < cin_pad; c0 += p.CK) {
    const int ck = min(p.CK, cin_pad - c0);
    const int G = ck >> 2;   // float4 groups per pixel in this chunk
    __syncthreads();         // previous chunk fully consumed
    // ---- stage the halo tile (padding applied here).  Units of 4 channels; 8 loads per lane are issued back to back from
    // always-valid (select-ed) addresses and only then padded and written to LDS, so a lane keeps 8 global loads in flight
    // instead of paying one full memory round trip per element.
    {
      const int upr = TC * G, total = TR * upr;
      const float inv_upr = 1.0f / (float)upr, inv_G = 1.0f / (float)G;   // exact small-integer division via (u + 0.5) / d
      const bool vec = p.vec_ok != 0;
      for (int base = tid; base < total; base += 256 * 8) {
        f32x4 v[8];
        int off[8];   // LDS float offset; -1: nothing to do; <= -2: out-of-image (constant padding), offset = -2 - off
        int chs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int u = base + 256 * q;
          const int uu = u < total ? u : 0;
          const int r = (int)(((float)uu + 0.5f) * inv_upr);
          const int w = uu - r * upr;
          const int c = (int)(((float)w + 0.5f) * inv_G);
          const int g = w - c * G;
          const int sy = pcnn_pad_index(y0 + r - p.pt, p.H, p.pad_mode);
          const int sx = pcnn_pad_index(x0 + c - p.pl, p.W, p.pad_mode);
          const bool inimg = sy >= 0 && sx >= 0;
          const int ch = c0 + 4 * g;
          const float* src = xin + ((int64_t)(inimg ? sy : 0) * p.W + (inimg ? sx : 0)) * p.ldx;
          if (vec && ch + 3 < p.Cin) {
            v[q] = *reinterpret_cast<const f32x4*>(src + ch);
          } else {   // ragged / unaligned channel tail: clamp every element's address, mask afterwards
            v[q][0] = src[ch + 0 < p.Cin ? ch + 0 : p.Cin - 1]; v[q][1] = src[ch + 1 < p.Cin ? ch + 1 : p.Cin - 1];
            v[q][2] = src[ch + 2 < p.Cin ? ch + 2 : p.Cin - 1]; v[q][3] = src[ch + 3 < p.Cin ? ch + 3 : p.Cin - 1];
          }
          const int o = (r * TC + c) * PS + 4 * g;
          off[q] = u < total ? (inimg ? o : -2 - o) : -1;
          chs[q] = ch;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (off[q] != -1) {
            f32x4 t = v[q];
            int o = off[q];
            if (o < 0) { o = -2 - o; const float pv = p.pad_value; t[0] = pv; t[1] = pv; t[2] = pv; t[3] = pv; }
            const int ch = chs[q];
            if (ch + 3 >= p.Cin) {   // zero the channels beyond Cin (K padding)
              if (ch + 0 >= p.Cin) t[0] = 0.f;
              if (ch + 1 >= p.Cin) t[1] = 0.f;
              if (ch + 2 >= p.Cin) t[2] = 0.f;
              t[3] = 0.f;
            }
            *reinterpret_cast<f32x4*>(&lds[o]) = t;
          }
        }
      }
    }
    __syncthreads();

    // ---- K loop over (tap, 8-channel sub-chunk).  The packed filter is consumed strictly sequentially ([tap][group]), so
    // the B operand is a pointer that advances by one step; steps are unrolled by two with ping-pong registers so that the
    // load of step s+1 stays in flight (counted vmcnt) under the 16 MFMAs of step s and no register copies are needed.
    const int nsub = ck >> 3;
    const int nsub_tot = cin_pad >> 3;
    const int nsteps = p.kh * p.kw * nsub;
    const f32x4* wlane = reinterpret_cast<const f32x4*>(p.wp) + half * (NT * 32) + col;
    int tap = 0, sub = 0, ki = 0, kj = 0;
    auto load_b = [&](f32x4 (&b)[NT], int tp, int sb) {
      const f32x4* src = wlane + (int64_t)(tp * nsub_tot + (c0 >> 3) + sb) * (2 * NT * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = src[t * 32];
    };
    auto step = [&](const f32x4 (&bc)[NT], f32x4 (&bn)[NT], bool more) {
      int ntap = tap, nsb = sub + 1, nki = ki, nkj = kj;
      if (nsb == nsub) { nsb = 0; ntap = tap + 1; nkj = kj + 1; if (nkj == p.kw) { nkj = 0; nki = ki + 1; } }
      load_b(bn, ntap, nsb);   // unconditional (the scratch has one spare step past the end): keeps the in-flight count static
      const float* abase = &lds[((wave * MT + ki) * TC + (col + kj)) * PS + sub * 8 + 4 * half];
      f32x4 a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(abase + m * TC * PS);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], bc[t][j], acc[m][t], 0, 0, 0);
      if (nki != ki || !more) {   // end of a filter row: flush the block sum
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot[m][t][i] += acc[m][t][i]; acc[m][t][i] = 0.f; }
      }
      tap = ntap; sub = nsb; ki = nki; kj = nkj;
    };
    f32x4 b0[NT], b1[NT];
    load_b(b0, 0, 0);
    int s = 0;
    for (; s + 2 <= nsteps; s += 2) {
      step(b0, b1, true);
      step(b1, b0, s + 2 < nsteps);
    }
    if (s < nsteps) step(b0, b1, false);
  }

  // ---- epilogue (conv_epilogue.h): staged through the LDS the halo tile no longer needs
  ConvEpilogue e;
  e.bias = p.bias; e.bn_scale = p.bn_scale; e.bn_shift = p.bn_shift; e.res = p.res; e.y = p.y; e.act_out = p.act_out;
  e.Ho = p.Ho; e.Wo = p.Wo; e.Cout = p.Cout; e.ldy = p.ldy; e.ld_res = p.ld_res; e.ld_act = p.ld_act; e.act = p.act; e.alpha = p.alpha;
  e.vec = p.epi_vec; e.absmax = p.y_absmax;
  conv_epilogue_store<MT, NT>(tot, 1.0f, e, n, y0, x0, lds);
}

// w (kh,kw,Cin,Cout) HWIO -> wp [tap][cin_pad/8][2][NT*32][4], zero padded: wp[...][g][h][co][j] = w[tap][8g + 4h + j][co]
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int taps, int Cin, int Cout, int cin_pad, int NT) {
  const int64_t total = (int64_t)taps * cin_pad * NT * 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = i & 3; int64_t r = i >> 2; const int co = r % (NT * 32); r /= (NT * 32); const int h = r & 1; r >>= 1;
    const int g = r % (cin_pad >> 3); const int tap = r / (cin_pad >> 3);
    const int ci = 8 * g + 4 * h + j;
    wp[i] = (ci < Cin && co < Cout) ? w[((int64_t)tap * Cin + ci) * Cout + co] : 0.f;
  }
}

__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int kh, int kw, int Cin, int Cout) {
  const int64_t total = (int64_t)kh * kw * Cin * Cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // i indexes wt (kh,kw,Cout,Cin)
    int ci = i % Cin; int64_t r = i / Cin; int co = r % Cout; r /= Cout; int j = r % kw; int ii = r / kw;
    wt[i] = w[(((int64_t)(kh - 1 - ii) * kw + (kw - 1 - j)) * Cin + ci) * Cout + co];
  }
}

// the same for a whole table of filters in one launch (pcnn_conv2d_flip_transpose_table): element i of the concatenated outputs belongs to the
// last entry whose `start` is <= i
__global__ void flip_transpose_table_kernel(const pcnn_flip_item* __restrict__ tab, int n, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].start <= i) lo = mid; else hi = mid - 1; }
    const pcnn_flip_item it = tab[lo];
    const int64_t e = i - it.start;
    int ci = e % it.Cin; int64_t r = e / it.Cin; int co = r % it.Cout; r /= it.Cout; int j = r % it.kw; int ii = r / it.kw;
    it.wt[e] = it.w[(((int64_t)(it.kh - 1 - ii) * it.kw + (it.kw - 1 - j)) * it.Cin + ci) * it.Cout + co];
  }
}

}  // namespace

extern "C" int pcnn_conv2d_fwd_absmax(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias,
                                      const float* bn_scale, const float* bn_shift, const float* residual, float* y, float* act_out, float* y_absmax) {
  PCNN_REQUIRE(h, h != nullptr, "pcnn_conv2d_fwd_absmax: null handle");
  if (y_absmax) (void)hipMemsetAsync(y_absmax, 0, sizeof(float), h->stream);
  h->y_absmax = y_absmax;
  const int rc = pcnn_conv2d_fwd(h, d, x, w, bias, bn_scale, bn_shift, residual, y, act_out);
  h->y_absmax = nullptr;
  return rc;
}

extern "C" int pcnn_conv2d_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias,
                               const float* bn_scale, const float* bn_shift, const float* residual, float* y, float* act_out) {
  PCNN_REQUIRE(h, h && d && x && w && y, "pcnn_conv2d_fwd: null argument");
  PCNN_REQUIRE(h, d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "pcnn_conv2d_fwd: empty tensor");
  PCNN_REQUIRE(h, d->Cin >= 1 && d->Cout >= 1 && d->Cout <= 64, "pcnn_conv2d_fwd: Cout=%d unsupported (1..64)", d->Cout);
  PCNN_REQUIRE(h, d->kh >= 1 && d->kw >= 1 && d->kh <= 31 && d->kw <= 31, "pcnn_conv2d_fwd: kernel %dx%d unsupported", d->kh, d->kw);
  PCNN_REQUIRE(h, d->ldx >= d->Cin && d->ldy >= d->Cout, "pcnn_conv2d_fwd: channel stride smaller than channel count");
  PCNN_REQUIRE(h, d->pad_mode >= 0 && d->pad_mode <= 2, "pcnn_conv2d_fwd: bad pad_mode %d", d->pad_mode);
  PCNN_REQUIRE(h, (bn_scale == nullptr) == (bn_shift == nullptr), "pcnn_conv2d_fwd: bn_scale and bn_shift go together");
  // every read must resolve inside the image: the farthest tap of the last output must be within pad reach
  if (d->pad_mode != PCNN_PAD_CONSTANT) {
    const int lim_y = d->pad_mode == PCNN_PAD_SYMMETRIC ? d->H : d->H - 1, lim_x = d->pad_mode == PCNN_PAD_SYMMETRIC ? d->W : d->W - 1;
    const int pb = d->Ho - 1 - d->pad_top + d->kh - 1 - (d->H - 1), pr = d->Wo - 1 - d->pad_left + d->kw - 1 - (d->W - 1);
    PCNN_REQUIRE(h, d->pad_top <= lim_y && pb <= lim_y && d->pad_left <= lim_x && pr <= lim_x,
                 "pcnn_conv2d_fwd: padding exceeds what tf.pad allows for a %dx%d image", d->H, d->W);
  }
  if (pcnn_conv_fwd_takes_narrow_route(h, d)) return pcnn_conv_small_fwd(h, d, x, w, bias, bn_scale, bn_shift, residual, y, act_out);
  if (pcnn_spectral_eligible(h, d, false)) return pcnn_spectral_conv_fwd(h, d, x, w, bias, bn_scale, bn_shift, residual, y, act_out);
  if (h->math_mode == PCNN_MATH_SPLIT_F16) {
    const int rc = pcnn_conv2d_fwd_split(h, d, x, w, bias, bn_scale, bn_shift, residual, y, act_out);
    if (rc >= 0) return rc;   // -1: shape not covered by the split kernel -> exact fp32 path below
  }
  ConvParams p;
  p.x = x; p.w = w; p.bias = bias; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.res = residual; p.y = y; p.act_out = act_out;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.ldy = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.act = d->act; p.alpha = d->act_alpha; p.ld_res = d->ld_res; p.ld_act = d->ld_act_out;
  p.tiles_x = pcnn_cdiv(d->Wo, TW); p.tiles_y = pcnn_cdiv(d->Ho, TH);
  p.vec_ok = (d->ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const int cin_pad = (d->Cin + 7) & ~7;
  const int TR = TH + d->kh - 1, TC = TW + d->kw - 1;
  int CK = 8;
  for (int cand = 64; cand >= 8; cand >>= 1) {
    if (cin_pad % cand) continue;
    const int ps = ((cand / 4) % 2 == 0) ? cand + 4 : cand;
    if ((size_t)TR * TC * ps * 4 <= MAX_LDS_BYTES) { CK = cand; break; }
  }
  p.CK = CK; p.PS = ((CK / 4) % 2 == 0) ? CK + 4 : CK;
  p.epi_vec = conv_epilogue_vec_ok(d->Cout, y, d->ldy, residual, d->ld_res, act_out, d->ld_act_out);
  p.y_absmax = reinterpret_cast<unsigned*>(h->y_absmax);
  const size_t lds = std::max((size_t)TR * TC * p.PS * 4, conv_epilogue_lds_bytes(d->Cout));
  PCNN_REQUIRE(h, lds <= 160 * 1024, "pcnn_conv2d_fwd: halo tile needs %zu B of LDS", lds);
  {
    const int NTh = d->Cout <= 32 ? 1 : 2;
    // + one full tap of spare K-steps: with several Cin chunks the last chunk's unconditional prefetch reads up to cin_pad / 8 - 1 steps past
    // the end (16 spare channels were only safe under the 4 MB minimum allocation)
    const size_t need = ((size_t)d->kh * d->kw * cin_pad + cin_pad + 16) * NTh * 32 * sizeof(float);
    if (h->scratch_bytes < need) {
      // grown on demand; stream-ordered reuse is safe because pack and conv run back to back on the handle's stream
      if (h->scratch) { pcnn_release(h, h->scratch); h->scratch = nullptr; h->scratch_bytes = 0; }
      size_t cap = need < (4u << 20) ? (4u << 20) : need;
      if (hipMalloc(&h->scratch, cap) != hipSuccess) PCNN_FAIL(h, "pcnn_conv2d_fwd: cannot allocate %zu B of filter scratch", cap);
      h->scratch_bytes = cap;
    }
    const int64_t total = (int64_t)d->kh * d->kw * cin_pad * NTh * 32;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(total, 256), 2048)), dim3(256), 0, h->stream, w,
                       static_cast<float*>(h->scratch), d->kh * d->kw, d->Cin, d->Cout, cin_pad, NTh);
    PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_fwd(pack)");
    p.wp = static_cast<const float*>(h->scratch);
  }
  const int64_t nblk = (int64_t)d->N * p.tiles_x * p.tiles_y;
  PCNN_REQUIRE(h, nblk < (1ll << 31), "pcnn_conv2d_fwd: grid too large");
  dim3 grid((unsigned)nblk), block(256);
  if (d->Cout <= 32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(conv_fwd_kernel<1>, grid, block, lds, h->stream, p);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(conv_fwd_kernel<2>, grid, block, lds, h->stream, p);
  }
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_fwd");
  return 0;
}

extern "C" int pcnn_conv2d_flip_transpose_table(pcnn_handle h, const pcnn_flip_item* table_dev, int n, int64_t total) {
  PCNN_REQUIRE(h, h && table_dev && n > 0 && total > 0, "pcnn_conv2d_flip_transpose_table: bad argument");
  const int blocks = (int)std::min<int64_t>(pcnn_cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(flip_transpose_table_kernel, dim3(blocks), dim3(256), 0, h->stream, table_dev, n, total);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_flip_transpose_table");
  return 0;
}

extern "C" int pcnn_conv2d_flip_transpose_weights(pcnn_handle h, const float* w, float* wt, int kh, int kw, int Cin, int Cout) {
  PCNN_REQUIRE(h, h && w && wt && kh > 0 && kw > 0 && Cin > 0 && Cout > 0, "pcnn_conv2d_flip_transpose_weights: bad argument");
  const int64_t total = (int64_t)kh * kw * Cin * Cout;
  const int blocks = (int)std::min<int64_t>(pcnn_cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(blocks), dim3(256), 0, h->stream, w, wt, kh, kw, Cin, Cout);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_flip_transpose_weights");
  return 0;
}
