// Narrow convolutions (<= 16 channels on both sides, 3x3 / 5x5) - the tail of the final stack (16->12, 12->12, 12->8, 8->8, 8->4, 4->1) and
// the Scaling convolutions.  Their arithmetic intensity (3.6 ... 27 FLOP/B) sits at or below the fp32 ridge, so the matrix cores have
// nothing to offer: a 32-wide MFMA tile would idle on 50-97 % of its lanes.  These kernels are built for the HBM roofline instead:
//
//   conv_small_fwd_kernel   thread = one output pixel with all its output channels in registers; the halo tile (boundary-condition
//                           padding applied by the loader) is staged once in LDS as whole pixels (conflict-free 16-byte reads), the
//                           zero-padded filter is read through the scalar cache (uniform addresses -> s_load, the FMAs take the weight
//                           as their SGPR operand), and the fused epilogue stores 16 bytes per lane.  Exact fp32 FMA chain in both math
//                           modes.  Also the data gradient (flipped / transposed filter).
//   conv_small_wgrad_kernel lane = pixel, wave = a group of filter taps whose Cin x Cout partial sums stay in registers while the
//                           workgroup streams its tiles; one cross-lane reduction per workgroup at the very end, partials summed in a
//                           fixed order (deterministic).  x and dz are each read once from HBM.
#include "pcnn_internal.h"
#include "conv_epilogue.h"

int pcnn_conv_small_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                        const float* bn_shift, const float* residual, float* y, float* act_out);
int pcnn_conv_small_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes);
bool pcnn_conv_small_fwd_eligible(const pcnn_conv_desc* d);
bool pcnn_conv_small_wgrad_eligible(const pcnn_conv_desc* d);
size_t pcnn_conv_small_wgrad_workspace(const pcnn_conv_desc* d);

namespace {

constexpr int STH = 8, STW = 32;                       // output tile: 8 rows x 32 columns = 256 pixels = 256 threads
__host__ __device__ constexpr int lds_stride(int CI) { return CI == 16 ? 20 : CI; }   // floats per staged pixel (16 would hit 4 banks 4-fold)

struct SmallParams {
  const float* x; const float* wp; const float* bias; const float* bn_scale; const float* bn_shift; const float* res; float* y; float* act_out;
  unsigned* absmax;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, pt, pl, pad_mode; float pad_value; int act; float alpha;
  int ld_res, ld_act, tiles_x, tiles_y, vec_in, vec_out;
};

// stages the (STH + K - 1) x (STW + K - 1) halo tile of image n as [pixel][CI (stride CIS)] floats, padding applied
template <int K, int CI>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ xin, int H, int W, int Cin, int ldx, int y0, int x0, int pt, int pl,
                                           int pad_mode, float pad_value, int vec) {
  constexpr int TR = STH + K - 1, TC = STW + K - 1, Q = CI / 4, CIS = lds_stride(CI);
  for (int u = threadIdx.x; u < TR * TC * Q; u += blockDim.x) {
    const int q = u % Q, pix = u / Q, r = pix / TC, c = pix - r * TC;
    const int sy = pcnn_pad_index(y0 + r - pt, H, pad_mode), sx = pcnn_pad_index(x0 + c - pl, W, pad_mode);
    f32x4 v;
    if (sy < 0 || sx < 0) {
      v = (f32x4){pad_value, pad_value, pad_value, pad_value};
    } else {
      const float* src = xin + ((int64_t)sy * W + sx) * ldx + 4 * q;
      if (vec && 4 * q + 3 < Cin) v = *reinterpret_cast<const f32x4*>(src);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 4 * q + j < Cin ? src[j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * q + j >= Cin) v[j] = 0.f;                 // channels beyond Cin (also under constant padding): zero
    *reinterpret_cast<f32x4*>(lds + pix * CIS + 4 * q) = v;
  }
}

// launch bound: the HBM-bound shapes (CI * CO <= 64) are compiled for 8 waves per SIMD (<= 64 VGPRs) - what hides the load -> compute ->
// store latency chain of a tile is the number of resident workgroups, not instruction-level tricks (a persistent variant with register
// prefetch of the next tile and a 4-rows-per-thread variant were measured: 2.3x and 1.1x SLOWER)
template <int K, int CI, int CO>
__global__ __launch_bounds__(256, (CI * CO <= 64 ? 8 : 4)) void conv_small_fwd_kernel(SmallParams p) {
  constexpr int TC = STW + K - 1, CIS = lds_stride(CI);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int tile = blockIdx.x;
  const int tx = tile % p.tiles_x; tile /= p.tiles_x;
  const int ty = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int y0 = ty * STH, x0 = tx * STW;
  stage_tile<K, CI>(lds, p.x + (int64_t)n * p.H * p.W * p.ldx, p.H, p.W, p.Cin, p.ldx, y0, x0, p.pt, p.pl, p.pad_mode, p.pad_value, p.vec_in);
  __syncthreads();
  const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  const float* wp = p.wp;                                 // [K*K][CI][CO], zero padded: uniform addresses -> scalar loads
#pragma unroll 1
  for (int i = 0; i < K; ++i)                                // filter rows stay a loop: K*K*CI*CO unrolled FMAs would not fit the register file
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float* px = lds + ((r + i) * TC + (c + j)) * CIS;
      float xv[CI];
#pragma unroll
      for (int q = 0; q < CI / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(px + 4 * q);
        xv[4 * q] = v[0]; xv[4 * q + 1] = v[1]; xv[4 * q + 2] = v[2]; xv[4 * q + 3] = v[3];
      }
#pragma unroll
      for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = fmaf(xv[ci], wp[((i * K + j) * CI + ci) * CO + o], acc[o]);
    }
  // ---- fused epilogue: bias -> activation -> (act_out) -> BN affine -> residual -> y
  const int oy = y0 + r, ox = x0 + c;
  float ymax = 0.f;
  if (oy < p.Ho && ox < p.Wo) {
    const int64_t pix = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
    float out[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      float v = 0.f;
      if (o < p.Cout) {
        v = pcnn_act(acc[o] + (p.bias ? p.bias[o] : 0.f), p.act, p.alpha);
        if (p.act_out && !p.vec_out) p.act_out[pix * p.ld_act + o] = v;
      }
      out[o] = v;
    }
    if (p.vec_out) {
      if (p.act_out) {
#pragma unroll
        for (int q = 0; q < CO / 4; ++q)
          if (4 * q < p.Cout) *reinterpret_cast<f32x4*>(p.act_out + pix * p.ld_act + 4 * q) = (f32x4){out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
      }
#pragma unroll
      for (int q = 0; q < CO / 4; ++q) {
        if (4 * q < p.Cout) {
          f32x4 v = {out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
          if (p.bn_scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] * p.bn_scale[4 * q + j] + p.bn_shift[4 * q + j];
          }
          if (p.res) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(p.res + pix * p.ld_res + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rr[j];
          }
          *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + 4 * q) = v;
#pragma unroll
          for (int j = 0; j < 4; ++j) ymax = fmaxf(ymax, fabsf(v[j]));
        }
      }
    } else {
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        if (o < p.Cout) {
          float v = out[o];
          if (p.bn_scale) v = v * p.bn_scale[o] + p.bn_shift[o];
          if (p.res) v += p.res[pix * p.ld_res + o];
          p.y[pix * p.ldy + o] = v;
          ymax = fmaxf(ymax, fabsf(v));
        }
      }
    }
  }
  conv_epilogue_absmax(p.absmax, ymax);
}

// w (K,K,Cin,Cout) -> wp [K*K][CI][CO] zero padded
__global__ void pack_small_kernel(const float* __restrict__ w, float* __restrict__ wp, int taps, int Cin, int Cout, int CI, int CO) {
  const int total = taps * CI * CO;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int o = i % CO; int r = i / CO; const int ci = r % CI; const int t = r / CI;
    wp[i] = (ci < Cin && o < Cout) ? w[((int64_t)t * Cin + ci) * Cout + o] : 0.f;
  }
}

template <int K, int CI, int CO>
void launch_small_fwd(pcnn_handle h, const SmallParams& p, int64_t nblk) {
  constexpr size_t lds = (size_t)(STH + K - 1) * (STW + K - 1) * lds_stride(CI) * sizeof(float);
  hipLaunchKernelGGL((conv_small_fwd_kernel<K, CI, CO>), dim3((unsigned)nblk), dim3(256), lds, h->stream, p);
}

template <int K, int CI>
bool dispatch_co(pcnn_handle h, const SmallParams& p, int64_t nblk, int CO) {
  switch (CO) {
    case 4: launch_small_fwd<K, CI, 4>(h, p, nblk); return true;
    case 8: launch_small_fwd<K, CI, 8>(h, p, nblk); return true;
    case 12: launch_small_fwd<K, CI, 12>(h, p, nblk); return true;
    case 16: launch_small_fwd<K, CI, 16>(h, p, nblk); return true;
  }
  return false;
}
template <int K>
bool dispatch_ci(pcnn_handle h, const SmallParams& p, int64_t nblk, int CI, int CO) {
  switch (CI) {
    case 4: return dispatch_co<K, 4>(h, p, nblk, CO);
    case 8: return dispatch_co<K, 8>(h, p, nblk, CO);
    case 12: return dispatch_co<K, 12>(h, p, nblk, CO);
    case 16: return dispatch_co<K, 16>(h, p, nblk, CO);
  }
  return false;
}

// ---------------------------------------------------------------------------------------------------------------- weight gradient (3x3)
struct SmallWgradParams {
  const float* x; const float* dz; float* part;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, pt, pl, pad_mode; float pad_value;
  int tiles_x, tiles_y, ntiles, S, vec_in, vec_dz;
};

// Workgroup = 3 waves; wave w of tap-group block yi owns taps (3 yi + w) TPW ... + TPW - 1 (< 9).  The ny workgroups that sweep the same
// tiles for different taps are decoded from blockIdx.x so that they sit on one XCD, adjacent in dispatch order: x and dz then come from
// HBM once and from that XCD's L2 for the siblings.
template <int CI, int CO, int TPW>
__global__ __launch_bounds__(192) void conv_small_wgrad_kernel(SmallWgradParams p) {
  constexpr int K = 3, TC = STW + K - 1, CIS = lds_stride(CI), COS = lds_stride(CO);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* zs = lds + (STH + K - 1) * TC * CIS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NY = (9 + 3 * TPW - 1) / (3 * TPW);
  const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3, yi = qq % NY, split = (qq / NY) * 8 + xcd;
  if (split >= p.S) return;
  const int tap0 = (yi * 3 + wave) * TPW;
  float acc[TPW][CI][CO];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int a = 0; a < CI; ++a)
#pragma unroll
      for (int b = 0; b < CO; ++b) acc[t][a][b] = 0.f;
  for (int tile = split; tile < p.ntiles; tile += p.S) {
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * STH, x0 = tx * STW;
    __syncthreads();
    stage_tile<K, CI>(xs, p.x + (int64_t)n * p.H * p.W * p.ldx, p.H, p.W, p.Cin, p.ldx, y0, x0, p.pt, p.pl, p.pad_mode, p.pad_value, p.vec_in);
    {   // dz tile (zero outside the image)
      const float* zin = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
      constexpr int Q = CO / 4;
      for (int u = tid; u < STH * STW * Q; u += 192) {
        const int q = u % Q, pix = u / Q, r = pix / STW, c = pix - r * STW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y0 + r < p.Ho && x0 + c < p.Wo) {
          const float* src = zin + ((int64_t)(y0 + r) * p.Wo + x0 + c) * p.lddz + 4 * q;
          if (p.vec_dz && 4 * q + 3 < p.Cout) v = *reinterpret_cast<const f32x4*>(src);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 4 * q + j < p.Cout ? src[j] : 0.f;
          }
        }
        *reinterpret_cast<f32x4*>(zs + pix * COS + 4 * q) = v;
      }
    }
    __syncthreads();
    if (tap0 < 9) {
#pragma unroll 1
      for (int b = 0; b < 4; ++b) {                      // 4 batches of 64 pixels: lane = pixel
        const int pix = b * 64 + lane, r = pix >> 5, c = pix & 31;
        float zv[CO];
#pragma unroll
        for (int q = 0; q < CO / 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(zs + pix * COS + 4 * q);
          zv[4 * q] = v[0]; zv[4 * q + 1] = v[1]; zv[4 * q + 2] = v[2]; zv[4 * q + 3] = v[3];
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int tap = tap0 + t;
          if (tap < 9) {
            const int i = tap / 3, j = tap - 3 * i;
            const float* px = xs + ((r + i) * TC + (c + j)) * CIS;
#pragma unroll
            for (int q = 0; q < CI / 4; ++q) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(px + 4 * q);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int o = 0; o < CO; ++o) acc[t][4 * q + jj][o] = fmaf(v[jj], zv[o], acc[t][4 * q + jj][o]);
            }
          }
        }
      }
    }
  }
  // ---- one reduction over the 64 lanes per accumulator, then partial[split][tap][ci][co]
  if (tap0 < 9) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tap = tap0 + t;
#pragma unroll
      for (int a = 0; a < CI; ++a)
#pragma unroll
        for (int b = 0; b < CO; ++b) {
          float v = acc[t][a][b];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if (lane == 0 && tap < 9 && a < p.Cin && b < p.Cout) p.part[(((int64_t)split * 9 + tap) * p.Cin + a) * p.Cout + b] = v;
        }
    }
  }
}

__global__ void small_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nel, int S) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nel) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[(int64_t)k * nel + e];
  dw[e] = s;
}

int wgrad_splits(const pcnn_conv_desc* d) {
  const int64_t ntiles = (int64_t)d->N * pcnn_cdiv(d->Ho, STH) * pcnn_cdiv(d->Wo, STW);
  return (int)std::min<int64_t>(ntiles, 1024);
}

template <int CI, int CO, int TPW>
void launch_small_wgrad(pcnn_handle h, const SmallWgradParams& p) {
  constexpr size_t lds = ((size_t)(STH + 2) * (STW + 2) * lds_stride(CI) + (size_t)STH * STW * lds_stride(CO)) * sizeof(float);
  constexpr int NY = (9 + 3 * TPW - 1) / (3 * TPW);        // workgroups per split: 3 tap groups (waves) each
  hipLaunchKernelGGL((conv_small_wgrad_kernel<CI, CO, TPW>), dim3((unsigned)(8 * pcnn_cdiv(p.S, 8) * NY)), dim3(192), lds, h->stream, p);
}

}  // namespace

static int pad4(int c) { return (c + 3) & ~3; }

bool pcnn_conv_small_fwd_eligible(const pcnn_conv_desc* d) {
  static const int on = getenv("PCNN_SMALL_CONV") ? atoi(getenv("PCNN_SMALL_CONV")) : 1;
  if (!on || d->kh != d->kw || (d->kh != 3 && d->kh != 5) || d->Cin > 16 || d->Cout > 16) return false;
  if (d->kh == 5 && pad4(d->Cin) * pad4(d->Cout) > 16 * 16) return false;
  return true;
}

bool pcnn_conv_small_wgrad_eligible(const pcnn_conv_desc* d) {
  static const int on = getenv("PCNN_SMALL_CONV") ? atoi(getenv("PCNN_SMALL_CONV")) : 1;
  return on && d->kh == 3 && d->kw == 3 && d->Cin <= 16 && d->Cout <= 16 && pad4(d->Cin) * pad4(d->Cout) <= 192;
}

size_t pcnn_conv_small_wgrad_workspace(const pcnn_conv_desc* d) {
  return pcnn_conv_small_wgrad_eligible(d) ? (size_t)wgrad_splits(d) * 9 * d->Cin * d->Cout * sizeof(float) : 0;
}

int pcnn_conv_small_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                        const float* bn_shift, const float* residual, float* y, float* act_out) {
  const int CI = pad4(d->Cin), CO = pad4(d->Cout), taps = d->kh * d->kw;
  const size_t need = (size_t)taps * CI * CO * sizeof(float);
  if (h->scratch_bytes < need) {
    if (h->scratch) { (void)hipStreamSynchronize(h->stream); (void)hipFree(h->scratch); h->scratch = nullptr; h->scratch_bytes = 0; }
    const size_t cap = 4u << 20;
    if (hipMalloc(&h->scratch, cap) != hipSuccess) PCNN_FAIL(h, "pcnn_conv2d_fwd: cannot allocate %zu B of filter scratch", cap);
    h->scratch_bytes = cap;
  }
  float* wp = static_cast<float*>(h->scratch);
  hipLaunchKernelGGL(pack_small_kernel, dim3(pcnn_cdiv(taps * CI * CO, 256)), dim3(256), 0, h->stream, w, wp, taps, d->Cin, d->Cout, CI, CO);
  SmallParams p;
  p.x = x; p.wp = wp; p.bias = bias; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.res = residual; p.y = y; p.act_out = act_out;
  p.absmax = reinterpret_cast<unsigned*>(h->y_absmax);
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.ldy = d->ldy;
  p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value; p.act = d->act; p.alpha = d->act_alpha;
  p.ld_res = d->ld_res; p.ld_act = d->ld_act_out;
  p.tiles_x = pcnn_cdiv(d->Wo, STW); p.tiles_y = pcnn_cdiv(d->Ho, STH);
  p.vec_in = (d->ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.vec_out = conv_epilogue_vec_ok(d->Cout, y, d->ldy, residual, d->ld_res, act_out, d->ld_act_out);
  const int64_t nblk = (int64_t)d->N * p.tiles_x * p.tiles_y;
  PCNN_REQUIRE(h, nblk < (1ll << 31), "pcnn_conv2d_fwd: grid too large");
  const bool ok = d->kh == 3 ? dispatch_ci<3>(h, p, nblk, CI, CO) : dispatch_ci<5>(h, p, nblk, CI, CO);
  PCNN_REQUIRE(h, ok, "pcnn_conv2d_fwd(small): no kernel for %d->%d", d->Cin, d->Cout);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_fwd(small)");
  return 0;
}

int pcnn_conv_small_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_conv_small_wgrad_workspace(d), "pcnn_conv2d_wgrad(small): workspace too small");
  const int CI = pad4(d->Cin), CO = pad4(d->Cout);
  SmallWgradParams p;
  p.x = x; p.dz = dz; p.part = static_cast<float*>(workspace);
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.lddz = d->ldy;
  p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.tiles_x = pcnn_cdiv(d->Wo, STW); p.tiles_y = pcnn_cdiv(d->Ho, STH); p.ntiles = d->N * p.tiles_x * p.tiles_y; p.S = wgrad_splits(d);
  p.vec_in = (d->ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.vec_dz = (d->ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15) == 0);
#define PCNN_SW(A, B, T) if (CI == A && CO == B) launch_small_wgrad<A, B, T>(h, p); else
  PCNN_SW(4, 4, 3) PCNN_SW(4, 8, 3) PCNN_SW(8, 4, 3) PCNN_SW(8, 8, 3) PCNN_SW(4, 12, 3) PCNN_SW(12, 4, 3) PCNN_SW(4, 16, 3) PCNN_SW(16, 4, 3)
  PCNN_SW(8, 12, 2) PCNN_SW(12, 8, 2) PCNN_SW(8, 16, 1) PCNN_SW(16, 8, 1) PCNN_SW(12, 12, 1) PCNN_SW(12, 16, 1) PCNN_SW(16, 12, 1)
  { PCNN_FAIL(h, "pcnn_conv2d_wgrad(small): no kernel for %d->%d", d->Cin, d->Cout); }
#undef PCNN_SW
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(small)");
  const int nel = 9 * d->Cin * d->Cout;
  hipLaunchKernelGGL(small_wgrad_reduce_kernel, dim3(pcnn_cdiv(nel, 256)), dim3(256), 0, h->stream, p.part, dw, nel, p.S);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(small reduce)");
  return 0;
}
