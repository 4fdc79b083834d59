// Narrow convolutions (<= 16 channels on both sides, 3x3 / 5x5) - the tail of the final stack (16->12, 12->12, 12->8, 8->8, 8->4, 4->1) and
// the Scaling convolutions.  Their arithmetic intensity (3.6 ... 27 FLOP/B) sits at or below the fp32 ridge, so the matrix cores have
// nothing to offer: a 32-wide MFMA tile would idle on 50-97 % of its lanes.  These kernels are built for the HBM roofline instead:
//
//   conv_small_fwd_kernel   thread = one output pixel with all its output channels in registers; the halo tile (boundary-condition
//                           padding applied by the loader) is staged once in LDS as whole pixels (conflict-free 16-byte reads), the
//                           zero-padded filter is read through the scalar cache (uniform addresses -> s_load, the FMAs take the weight
//                           as their SGPR operand), and the fused epilogue stores 16 bytes per lane.  Exact fp32 FMA chain in both math
//                           modes.  Also the data gradient (flipped / transposed filter).
//   conv_small_wgrad_kernel the one place where the matrix cores do pay: K = pixels is long, so v_mfma_f32_16x16x4_f32 tiles (M = ci, N = co) run at
//                           full rate whatever the channel counts; wave = one filter row, operands straight from the staged NHWC tile,
//                           partials summed in a fixed order (deterministic).  x and dz are each read once from HBM.
#include "pcnn_internal.h"
#include "conv_epilogue.h"

int pcnn_conv_small_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                        const float* bn_shift, const float* residual, float* y, float* act_out);
int pcnn_conv_small_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes);
bool pcnn_conv_fwd_takes_narrow_route(pcnn_handle h, const pcnn_conv_desc* d);                            // spectral_conv.hip
bool pcnn_conv_small_fwd_eligible(const pcnn_conv_desc* d);
bool pcnn_conv_small_wgrad_eligible(const pcnn_conv_desc* d);
size_t pcnn_conv_small_wgrad_workspace(const pcnn_conv_desc* d);

#ifndef PCNN_SMALL_STUDY
#define PCNN_SMALL_STUDY 0       // diagnostic builds only: 1 no FMA loop, 2 no global loads in the tile staging, 4 one epilogue store instead of all
#endif

namespace {

constexpr int STH = 8, STW = 32;                       // output tile: 8 rows x 32 columns = 256 pixels = 256 threads
__host__ __device__ constexpr int lds_stride(int CI) { return CI == 16 ? 20 : CI; }   // floats per staged pixel (16 would hit 4 banks 4-fold)

struct SmallParams {
  const float* x; const float* wp; const float* bias; const float* bn_scale; const float* bn_shift; const float* res; float* y; float* act_out;
  unsigned* absmax;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, pt, pl, pad_mode; float pad_value; int act; float alpha;
  int ld_res, ld_act, tiles_x, tiles_y, vec_in, vec_out;
  // POST (data-gradient launches, round 6): the activation backward of the layer that PRODUCED this convolution's input, applied to the gradient before it is
  // stored - v = conv (+ residual); y2 (if given) receives v, y receives v * act'(gact) (gact = that layer's saved activation output), and the workgroup's
  // per-channel sums of what it stored go to bpart[block * CO + o] (the bias gradient's partial sums, reduced in a fixed order by small_post_bias_kernel)
  const float* gact = nullptr; float* y2 = nullptr; float* bpart = nullptr;
  int ld_gact = 0, ld_y2 = 0, gmode = 0; float galpha = 1.f;
};

// tf.pad index map without control flow (selects only): a load whose address depends on it can be issued unconditionally, so that the
// loads of an unrolled staging loop are all in flight together (with branches every load sits in its own block behind an s_waitcnt vmcnt(0))
__device__ __forceinline__ int pad_index_sel(int i, int n, int mode) {
  const int refl = mode == PCNN_PAD_SYMMETRIC ? (i < 0 ? -i - 1 : 2 * n - 1 - i) : (i < 0 ? -i : 2 * n - 2 - i);
  const int rc = min(max(refl, 0), n - 1);
  return (unsigned)i < (unsigned)n ? i : (mode == PCNN_PAD_CONSTANT ? -1 : rc);
}

// stages the (STH + K - 1) x (STW + K - 1) halo tile of image n as [pixel][CI (stride CIS)] floats, padding applied
template <int K, int CI, int CIS = lds_stride(CI)>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ xin, int H, int W, int Cin, int ldx, int y0, int x0, int pt, int pl,
                                           int pad_mode, float pad_value, int vec) {
  constexpr int TR = STH + K - 1, TC = STW + K - 1, Q = CI / 4;
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

// The same tile, for whole 16-byte pieces (vec_in, Cin % 4 == 0), with ALL of a thread's loads in flight together: addresses are formed without
// control flow (pad_index_sel; padding pieces read a dummy address and are replaced when written to LDS).  The loop above compiles to one
// load per thread in flight - load, s_waitcnt vmcnt(0), ds_write, next - i.e. the tile's HBM latency is paid ceil(pieces / 256) times.
template <int K, int CI, int CIS = lds_stride(CI)>
__device__ __forceinline__ void stage_tile_batched(float* __restrict__ lds, const float* __restrict__ xin, int H, int W, int ldx, int y0, int x0, int pt, int pl,
                                                   int pad_mode, float pad_value) {
  constexpr int TR = STH + K - 1, TC = STW + K - 1, Q = CI / 4, TOTAL = TR * TC * Q, NX = (TOTAL + 255) / 256;
  f32x4 v[NX];
  unsigned okm = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int u = threadIdx.x + i * 256, q = u % Q, pix = u / Q, r = pix / TC, c = pix - r * TC;
    const int sy = pad_index_sel(y0 + r - pt, H, pad_mode), sx = pad_index_sel(x0 + c - pl, W, pad_mode);
    const bool ok = u < TOTAL && sy >= 0 && sx >= 0;
    const int64_t idx = ok ? ((int64_t)sy * W + sx) * ldx + 4 * q : 0;
    if (PCNN_SMALL_STUDY & 2) v[i] = (f32x4){(float)u, 1.f, 2.f, 3.f};
    else v[i] = *reinterpret_cast<const f32x4*>(xin + idx);
    okm |= ok ? (1u << i) : 0u;
  }
  const f32x4 padv = {pad_value, pad_value, pad_value, pad_value};
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int u = threadIdx.x + i * 256;
    if (u < TOTAL) *reinterpret_cast<f32x4*>(lds + (u / Q) * CIS + 4 * (u % Q)) = ((okm >> i) & 1u) ? v[i] : padv;
  }
}

// launch bound: the HBM-bound shapes (CI * CO <= 64) are compiled for 8 waves per SIMD (<= 64 VGPRs) - what hides the load -> compute ->
// store latency chain of a tile is the number of resident workgroups, not instruction-level tricks (a persistent variant with register
// prefetch of the next tile and a 4-rows-per-thread variant were measured: 2.3x and 1.1x SLOWER)
template <int K, int CI, int CO, bool POST = false>
__global__ __launch_bounds__(256, (CI * CO <= 64 ? 8 : 4)) void conv_small_fwd_kernel(SmallParams p) {
  constexpr int TC = STW + K - 1, CIS = lds_stride(CI);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int tile = blockIdx.x;
  const int tx = tile % p.tiles_x; tile /= p.tiles_x;
  const int ty = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int y0 = ty * STH, x0 = tx * STW;
  if (p.vec_in && (p.Cin & 3) == 0) stage_tile_batched<K, CI>(lds, p.x + (int64_t)n * p.H * p.W * p.ldx, p.H, p.W, p.ldx, y0, x0, p.pt, p.pl, p.pad_mode, p.pad_value);
  else stage_tile<K, CI>(lds, p.x + (int64_t)n * p.H * p.W * p.ldx, p.H, p.W, p.Cin, p.ldx, y0, x0, p.pt, p.pl, p.pad_mode, p.pad_value, p.vec_in);
  __syncthreads();
  const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  const float* wp = p.wp;                                 // [K*K][CI][CO], zero padded: uniform addresses -> scalar loads
#pragma unroll 1
  for (int i = 0; i < ((PCNN_SMALL_STUDY & 1) ? 0 : K); ++i)  // filter rows stay a loop: K*K*CI*CO unrolled FMAs would not fit the register file
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
  const int oy = y0 + r, ox = x0 + c;
  float ymax = 0.f;
  if constexpr (POST) {
    // ---- data gradient + the producer's activation backward (the launcher guarantees whole 16-byte channel quads on every tensor, a linear epilogue, no bias,
    // no BN): residual -> y2 (raw copy for a skip connection) -> times act'(gact) -> y; per-channel sums of y for the producer's bias gradient
    float bs[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) bs[o] = 0.f;
    if (oy < p.Ho && ox < p.Wo) {
      const int64_t pix = ((int64_t)n * p.Ho + oy) * p.Wo + ox;
      f32x4 rr[CO / 4], gg[CO / 4];
#pragma unroll
      for (int q = 0; q < CO / 4; ++q) {                       // all loads first: a load between two stores is waited for with vmcnt(0)
        const int cq = 4 * q < p.Cout ? 4 * q : 0;
        gg[q] = *reinterpret_cast<const f32x4*>(p.gact + pix * p.ld_gact + cq);
        if (p.res) rr[q] = *reinterpret_cast<const f32x4*>(p.res + pix * p.ld_res + cq);
      }
#pragma unroll
      for (int q = 0; q < CO / 4; ++q) {
        if (4 * q < p.Cout) {
          f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          if (p.res) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rr[q][j];
          }
          if (p.y2) *reinterpret_cast<f32x4*>(p.y2 + pix * p.ld_y2 + 4 * q) = v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float gq = gg[q][j];
            v[j] *= p.gmode == PCNN_ACT_TANH ? 1.f - gq * gq : (gq > 0.f ? 1.f : p.galpha);
            bs[4 * q + j] = v[j];
            ymax = fmaxf(ymax, fabsf(v[j]));
          }
          *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + 4 * q) = v;
        }
      }
    }
    // the workgroup's sums, in a fixed order: butterfly over the 64 lanes of a wave, then the four waves one after the other
    if (p.bpart) {
      __syncthreads();                                           // every thread has finished reading the staged tile: its LDS is free
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        float v = bs[o];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if ((threadIdx.x & 63) == 0) lds[(threadIdx.x >> 6) * CO + o] = v;
      }
      __syncthreads();
      if (threadIdx.x < CO) p.bpart[(int64_t)blockIdx.x * CO + threadIdx.x] = ((lds[threadIdx.x] + lds[CO + threadIdx.x]) + lds[2 * CO + threadIdx.x]) + lds[3 * CO + threadIdx.x];
    }
    conv_epilogue_absmax(p.absmax, ymax);
    return;
  }
  // ---- fused epilogue: bias -> activation -> (act_out) -> BN affine -> residual -> y
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
      f32x4 rr[CO / 4];                                    // the residual quads first, as one burst: a load between two stores is waited for with vmcnt(0)
      if (p.res) {
#pragma unroll
        for (int q = 0; q < CO / 4; ++q) rr[q] = *reinterpret_cast<const f32x4*>(p.res + pix * p.ld_res + (4 * q < p.Cout ? 4 * q : 0));
      }
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
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rr[q][j];
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
  static_assert(lds >= 4 * 16 * sizeof(float), "the POST epilogue reduces 4 waves x CO sums through the tile's LDS");
  if (p.gact) hipLaunchKernelGGL((conv_small_fwd_kernel<K, CI, CO, true>), dim3((unsigned)nblk), dim3(256), lds, h->stream, p);
  else hipLaunchKernelGGL((conv_small_fwd_kernel<K, CI, CO, false>), dim3((unsigned)nblk), dim3(256), lds, h->stream, p);
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

// ---------------------------------------------------------------------------------------------------------------- weight gradient (3x3, 5x5)
struct SmallWgradParams {
  const float* x; const float* dz; float* part;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, pt, pl, pad_mode; float pad_value;
  int tiles_x, tiles_y, ntiles, S, vec_in, vec_dz;
};

// dw[tap][ci][co] = sum over pixels of x[pixel + tap][ci] dz[pixel][co] on the matrix cores: v_mfma_f32_16x16x4_f32 with M = ci, N = co and
// K = four consecutive pixels of a tile row (exact fp32 products and accumulation, the arithmetic class of every other convolution kernel
// here).  One workgroup = K waves, wave i owns filter row i (its K tap accumulators are 4 registers each) and walks the staged tile:
// per K step one dz operand and K input operands, each ONE ds_read_b32 (lane = (pixel l / 16, channel l % 16) is exactly the NHWC order).
// Workgroups are persistent over tiles (tile += S); partial sums go to part[split][tap][ci][co] and are reduced in a fixed order.
template <int K, int CI, int CO>
__global__ __launch_bounds__(64 * K) void conv_small_wgrad_kernel(SmallWgradParams p) {
  constexpr int TR = STH + K - 1, TC = STW + K - 1, CIS = CI, COS = CO;      // unpadded: the operand reads are 4-byte, 16 consecutive channels per pixel
  constexpr int NT = 64 * K, QX = CI / 4, QZ = CO / 4, NX = (TR * TC * QX + NT - 1) / NT, NZ = (STH * STW * QZ + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* zs = lds + TR * TC * CIS + 16;                     // + slack: lanes of channels >= CI read (and discard) up to 15 floats past a pixel
  const int tid = threadIdx.x, lane = tid & 63, ch = lane & 15, kk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  // <= 8 (<= 4) input channels: two (four) taps of the filter row share one MFMA - rows 16 / GP * js + ci of the A tile hold tap g GP + js
  constexpr int GP = CI <= 4 ? 4 : (CI <= 8 ? 2 : 1), CPT = 16 / GP, NM = (K + GP - 1) / GP;
  const int js = ch / CPT, cc = ch % CPT;
  f32x4 acc[NM];
#pragma unroll
  for (int g = 0; g < NM; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool b_ok = ch < CO;
  // this thread's staging items: NX 16-byte pieces of the input halo tile, NZ of the dz tile (tile-independent coordinates)
  f32x4 xr[NX], zr[NZ];
  unsigned xok = ~0u, zok = ~0u;                            // fast path: bit i clear = piece i is padding / outside the image (its load read a dummy address)
  const bool fast = p.vec_in && p.vec_dz && (p.Cin & 3) == 0 && (p.Cout & 3) == 0;
  const f32x4 padv = {p.pad_value, p.pad_value, p.pad_value, p.pad_value}, zero4 = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int tile) {                              // global -> registers; the loads stay in flight under the previous tile's MFMAs
    int tt = tile;
    const int tx = tt % p.tiles_x; tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int y0 = ty * STH, x0 = tx * STW;
    const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
    const float* zin = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
    if (fast) {                                             // whole 16-byte pieces: branch-free addresses, every load of the tile in flight at once
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int u = tid + i * NT, q = u % QX, pix = u / QX, r = pix / TC, c = pix - r * TC;
        const int sy = pad_index_sel(y0 + r - p.pt, p.H, p.pad_mode), sx = pad_index_sel(x0 + c - p.pl, p.W, p.pad_mode);
        const bool ok = u < TR * TC * QX && sy >= 0 && sx >= 0;
        xr[i] = *reinterpret_cast<const f32x4*>(xin + (ok ? ((int64_t)sy * p.W + sx) * p.ldx + 4 * q : 0));
        xok = ok ? (xok | (1u << i)) : (xok & ~(1u << i));
      }
#pragma unroll
      for (int i = 0; i < NZ; ++i) {
        const int u = tid + i * NT, q = u % QZ, pix = u / QZ, r = pix / STW, c = pix - r * STW;
        const bool ok = u < STH * STW * QZ && y0 + r < p.Ho && x0 + c < p.Wo;
        zr[i] = *reinterpret_cast<const f32x4*>(zin + (ok ? ((int64_t)(y0 + r) * p.Wo + x0 + c) * p.lddz + 4 * q : 0));
        zok = ok ? (zok | (1u << i)) : (zok & ~(1u << i));
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int u = tid + i * NT, q = u % QX, pix = u / QX, r = pix / TC, c = pix - r * TC;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (u < TR * TC * QX) {
        const int sy = pcnn_pad_index(y0 + r - p.pt, p.H, p.pad_mode), sx = pcnn_pad_index(x0 + c - p.pl, p.W, p.pad_mode);
        if (sy < 0 || sx < 0) v = (f32x4){p.pad_value, p.pad_value, p.pad_value, p.pad_value};
        else {
          const float* src = xin + ((int64_t)sy * p.W + sx) * p.ldx + 4 * q;
          if (p.vec_in && 4 * q + 3 < p.Cin) v = *reinterpret_cast<const f32x4*>(src);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 4 * q + j < p.Cin ? src[j] : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * q + j >= p.Cin) v[j] = 0.f;               // channels beyond Cin (also under constant padding): zero
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const int u = tid + i * NT, q = u % QZ, pix = u / QZ, r = pix / STW, c = pix - r * STW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (u < STH * STW * QZ && y0 + r < p.Ho && x0 + c < p.Wo) {               // zero outside the image
        const float* src = zin + ((int64_t)(y0 + r) * p.Wo + x0 + c) * p.lddz + 4 * q;
        if (p.vec_dz && 4 * q + 3 < p.Cout) v = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = 4 * q + j < p.Cout ? src[j] : 0.f;
        }
      }
      zr[i] = v;
    }
  };
  int tile = split;
  if (tile < p.ntiles) fetch(tile);
  for (; tile < p.ntiles; tile += p.S) {
    __syncthreads();                                        // the previous tile's operands are no longer read
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int u = tid + i * NT;
      if (u < TR * TC * QX) *reinterpret_cast<f32x4*>(xs + (u / QX) * CIS + 4 * (u % QX)) = ((xok >> i) & 1u) ? xr[i] : padv;
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const int u = tid + i * NT;
      if (u < STH * STW * QZ) *reinterpret_cast<f32x4*>(zs + (u / QZ) * COS + 4 * (u % QZ)) = ((zok >> i) & 1u) ? zr[i] : zero4;
    }
    __syncthreads();
    if (tile + p.S < p.ntiles) fetch(tile + p.S);
#pragma unroll 1
    for (int r = 0; r < STH; ++r) {
      const float* zrow = zs + (r * STW + kk) * COS + ch;
      const float* xrow = xs + ((r + wave) * TC + kk + js) * CIS + cc;
#pragma unroll
      for (int c0 = 0; c0 < STW; c0 += 4) {
        float bv = zrow[c0 * COS];
        bv = b_ok ? bv : 0.f;
#pragma unroll
        for (int g = 0; g < NM; ++g) {
          float av = xrow[(c0 + g * GP) * CIS];
          av = (cc < CI && g * GP + js < K) ? av : 0.f;
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[g], 0, 0, 0);
        }
      }
    }
  }
  // accumulator register r of lane l: row (= ci) 4 (l / 16) + r, column (= co) l % 16
#pragma unroll
  for (int g = 0; g < NM; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kk + r, j = g * GP + row / CPT, ci = row % CPT;
      if (j < K && ci < p.Cin && ch < p.Cout) p.part[(((int64_t)split * K * K + wave * K + j) * p.Cin + ci) * p.Cout + ch] = acc[g][r];
    }
}

// dw[e] = sum over splits, sixteen interleaved partial sums per element combined in a fixed order
__global__ __launch_bounds__(256) void small_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nel, int S) {
  __shared__ float red[16][16];
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4, e = blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < nel)
    for (int k = q; k < S; k += 16) s += part[(int64_t)k * nel + e];
  red[q][el] = s;
  __syncthreads();
  if (q == 0 && e < nel) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][el];
    dw[e] = t;
  }
}

size_t wgrad_lds_bytes(int K, int CI, int CO) { return ((size_t)(STH + K - 1) * (STW + K - 1) * CI + 16 + (size_t)STH * STW * CO + 16) * sizeof(float); }

// one persistent workgroup per slot the chip can hold at once (LDS-limited, at most 8 per CU): every workgroup is resident from the start
// and walks the same number of tiles (+-1)
int wgrad_splits(const pcnn_conv_desc* d) {
  const int64_t ntiles = (int64_t)d->N * pcnn_cdiv(d->Ho, STH) * pcnn_cdiv(d->Wo, STW);
  const int per_cu = (int)std::min<size_t>(8, (160u << 10) / wgrad_lds_bytes(d->kh, (d->Cin + 3) & ~3, (d->Cout + 3) & ~3));
  return (int)std::min<int64_t>(ntiles, 256 * per_cu);
}

template <int K, int CI, int CO>
void launch_small_wgrad(pcnn_handle h, const SmallWgradParams& p) {
  hipLaunchKernelGGL((conv_small_wgrad_kernel<K, CI, CO>), dim3((unsigned)p.S), dim3(64 * K), wgrad_lds_bytes(K, CI, CO), h->stream, p);
}
template <int K, int CI>
bool wgrad_dispatch_co(pcnn_handle h, const SmallWgradParams& p, int CO) {
  switch (CO) {
    case 4: launch_small_wgrad<K, CI, 4>(h, p); return true;
    case 8: launch_small_wgrad<K, CI, 8>(h, p); return true;
    case 12: launch_small_wgrad<K, CI, 12>(h, p); return true;
    case 16: launch_small_wgrad<K, CI, 16>(h, p); return true;
  }
  return false;
}
template <int K>
bool wgrad_dispatch_ci(pcnn_handle h, const SmallWgradParams& p, int CI, int CO) {
  switch (CI) {
    case 4: return wgrad_dispatch_co<K, 4>(h, p, CO);
    case 8: return wgrad_dispatch_co<K, 8>(h, p, CO);
    case 12: return wgrad_dispatch_co<K, 12>(h, p, CO);
    case 16: return wgrad_dispatch_co<K, 16>(h, p, CO);
  }
  return false;
}


// ---------------------------------------------------------------------------------------------------------------- fused resnet stage (forward)
// blocks/resnet.py:29-39 on a narrow stage (3 x 3, C = 4 / 8 channels, zero CONSTANT padding, no BatchNormalization):
//   o0 = act(conv0(x) + b0);  a1 = act(conv1(o0) + b1);  o1 = x + a1;  y = act(conv2(o1) + b2)
// as ONE kernel: the workgroup stages x for its TH x 32 output tile with a 3-pixel halo, computes o0 on the tile + 2 pixels into LDS, o1 on the
// tile + 1 pixel in place over the staged x (only the thread that owns a pixel reads its x and writes its o1) and y from there.  The three
// separate launches move 8 tensor passes through HBM (conv1 reads o0 and the skip input and writes a1 and o1); this one reads x once and
// writes y - plus, when the backward pass will need them (training), o0, a1 and o1: 2 passes for inference, 5 for training.  The price is the
// halo recomputed per tile ((TH + 4)(36) + (TH + 2)(34) + 32 TH pixel convolutions instead of 3 x 32 TH) on the vector ALUs that already bound
// the single layers.  An intermediate OUTSIDE the image is the zero padding of the next convolution, not a
// convolution of padded input: it is stored as 0.  Arithmetic per output value: the same fp32 FMA chain, tap by tap, as conv_small_fwd_kernel
// (bit-identical results).  Measured at 8 x 1024^2 against the three launches (tools/probe_stage.py, training / inference): 4 channels 0.267 ->
// 0.167 ms / 0.263 -> 0.149 (0.80 of the HBM peak on the unfused layers' bytes), 8 channels 0.577 -> 0.540 / 0.537 -> 0.491 (0.50), 12 channels
// 1.267 -> 1.262 / 1.017 -> 1.221: at 12 channels the recomputed halo costs what the saved passes bring, so 12-channel stages keep three launches.
struct StageParams {
  const float* x; const float* w0; const float* w1; const float* w2; const float* b0; const float* b1; const float* b2;
  float* o0; float* a1; float* o1; float* y;
  int N, H, W, tiles_x, tiles_y; float slope;            // act(v) = v > 0 ? v : slope v (linear 1, relu 0, leaky relu alpha)
};

// The filters are read through the CONSTANT address space: between the three stages the kernel stores to global memory (o0, a1, o1), and a load
// that a preceding store might alias is not "invariant" for the compiler - it would fetch every weight with a vector load per lane (measured:
// the whole kernel 3.5 x slower) instead of the s_load + SGPR-operand form the single-layer kernel gets for free (its stores all come last).
typedef const float __attribute__((address_space(4))) * const_weights;

template <int C>
__device__ __forceinline__ void stage_conv3(const float* __restrict__ src, int pitch, const float* wg, float (&acc)[C]) {
  const const_weights w = (const_weights)wg;
#pragma unroll 1
  for (int i = 0; i < 3; ++i)                              // filter rows stay a loop (registers; see conv_small_fwd_kernel)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* px = src + (i * pitch + j) * C;
      float xv[C];
#pragma unroll
      for (int q = 0; q < C / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(px + 4 * q);
        xv[4 * q] = v[0]; xv[4 * q + 1] = v[1]; xv[4 * q + 2] = v[2]; xv[4 * q + 3] = v[3];
      }
#pragma unroll
      for (int ci = 0; ci < C; ++ci)
#pragma unroll
        for (int o = 0; o < C; ++o) acc[o] = fmaf(xv[ci], w[((i * 3 + j) * C + ci) * C + o], acc[o]);
    }
}

template <int C, int TH>
__global__ __launch_bounds__(256) void resnet3_stage_kernel(StageParams p) {
  constexpr int XR = TH + 6, XC = STW + 6, R0 = TH + 4, C0 = STW + 4, R1 = TH + 2, C1 = STW + 2, Q = C / 4;
  constexpr int TOTAL = XR * XC * Q, NX = (TOTAL + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const X = lds;                                    // [XR][XC][C]: x, later o1 (tile + 1 pixel) in place
  float* const O0 = lds + XR * XC * C;                     // [R0][C0][C]
  int tile = blockIdx.x;
  const int tx = tile % p.tiles_x; tile /= p.tiles_x;
  const int ty = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * STW;
  const int64_t img = (int64_t)n * p.H * p.W * C;
  {                                                        // x with a 3-pixel halo, zero outside the image; all loads of a thread in flight together
    const float* xin = p.x + img;
    f32x4 v[NX];
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int u = threadIdx.x + i * 256, q = u % Q, pix = u / Q, r = pix / XC, c = pix - r * XC;
      const int sy = y0 + r - 3, sx = x0 + c - 3;
      const bool ok = u < TOTAL && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
      v[i] = *reinterpret_cast<const f32x4*>(xin + (ok ? ((int64_t)sy * p.W + sx) * C + 4 * q : 0));
      okm |= ok ? (1u << i) : 0u;
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int u = threadIdx.x + i * 256;
      if (u < TOTAL) *reinterpret_cast<f32x4*>(X + (u / Q) * C + 4 * (u % Q)) = ((okm >> i) & 1u) ? v[i] : z4;
    }
  }
  __syncthreads();
  auto bias_act = [&](float (&acc)[C], const float* bg) {
    const const_weights b = (const_weights)bg;             // (constant address space: scalar loads, see stage_conv3)
#pragma unroll
    for (int o = 0; o < C; ++o) { const float t = acc[o] + (bg ? b[o] : 0.f); acc[o] = t > 0.f ? t : p.slope * t; }
  };
  auto store_px = [&](float* dst, int gy, int gx, const float (&val)[C]) {
    float* d = dst + img + ((int64_t)gy * p.W + gx) * C;
#pragma unroll
    for (int q = 0; q < Q; ++q) *reinterpret_cast<f32x4*>(d + 4 * q) = (f32x4){val[4 * q], val[4 * q + 1], val[4 * q + 2], val[4 * q + 3]};
  };
  // ---- o0 on the tile + 2 pixels
#pragma unroll 1
  for (int idx = threadIdx.x; idx < R0 * C0; idx += 256) {
    const int r = idx / C0, c = idx - r * C0, gy = y0 - 2 + r, gx = x0 - 2 + c;
    float acc[C];
#pragma unroll
    for (int o = 0; o < C; ++o) acc[o] = 0.f;
    stage_conv3<C>(X + (r * XC + c) * C, XC, p.w0, acc);
    bias_act(acc, p.b0);
    const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
#pragma unroll
    for (int q = 0; q < Q; ++q)
      *reinterpret_cast<f32x4*>(O0 + idx * C + 4 * q) = inside ? (f32x4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]} : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.o0 && inside && r >= 2 && r < TH + 2 && c >= 2 && c < STW + 2) store_px(p.o0, gy, gx, acc);
  }
  __syncthreads();
  // ---- a1 = act(conv1(o0)), o1 = x + a1 on the tile + 1 pixel, in place over x
#pragma unroll 1
  for (int idx = threadIdx.x; idx < R1 * C1; idx += 256) {
    const int r = idx / C1, c = idx - r * C1, gy = y0 - 1 + r, gx = x0 - 1 + c;
    float acc[C];
#pragma unroll
    for (int o = 0; o < C; ++o) acc[o] = 0.f;
    stage_conv3<C>(O0 + (r * C0 + c) * C, C0, p.w1, acc);
    bias_act(acc, p.b1);
    const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const bool own = inside && r >= 1 && r < TH + 1 && c >= 1 && c < STW + 1;
    if (p.a1 && own) store_px(p.a1, gy, gx, acc);
    float* xs = X + ((r + 2) * XC + (c + 2)) * C;
    float s[C];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[4 * q + j] = inside ? xv[j] + acc[4 * q + j] : 0.f;
      *reinterpret_cast<f32x4*>(xs + 4 * q) = (f32x4){s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]};
    }
    if (p.o1 && own) store_px(p.o1, gy, gx, s);
  }
  __syncthreads();
  // ---- y = act(conv2(o1)) on the tile
#pragma unroll 1
  for (int idx = threadIdx.x; idx < TH * STW; idx += 256) {
    const int r = idx >> 5, c = idx & 31, gy = y0 + r, gx = x0 + c;
    float acc[C];
#pragma unroll
    for (int o = 0; o < C; ++o) acc[o] = 0.f;
    stage_conv3<C>(X + ((r + 2) * XC + (c + 2)) * C, XC, p.w2, acc);
    bias_act(acc, p.b2);
    if (gy < p.H && gx < p.W) store_px(p.y, gy, gx, acc);
  }
}

template <int C, int TH>
void launch_stage(pcnn_handle h, StageParams p) {
  constexpr size_t lds = ((size_t)(TH + 6) * (STW + 6) + (size_t)(TH + 4) * (STW + 4)) * C * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resnet3_stage_kernel<C, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  p.tiles_x = pcnn_cdiv(p.W, STW); p.tiles_y = pcnn_cdiv(p.H, TH);
  hipLaunchKernelGGL((resnet3_stage_kernel<C, TH>), dim3((unsigned)(p.N * p.tiles_x * p.tiles_y)), dim3(256), lds, h->stream, p);
}

// POST: dbias[ch] = sum over the workgroups' partial sums, in a fixed order (thread t sums blocks t, t + 256, ...; then a tree).  One workgroup per channel.
__global__ __launch_bounds__(256) void small_post_bias_kernel(const float* __restrict__ bpart, int64_t nblk, int CO, float* __restrict__ dbias) {
  __shared__ float red[256];
  const int ch = blockIdx.x, t = threadIdx.x;
  float a = 0.f;
  for (int64_t b = t; b < nblk; b += 256) a += bpart[b * CO + ch];
  red[t] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (t < st) red[t] += red[t + st]; __syncthreads(); }
  if (t == 0) dbias[ch] = red[0];
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
  return on && d->kh == d->kw && (d->kh == 3 || d->kh == 5) && d->Cin <= 16 && d->Cout <= 16;
}

size_t pcnn_conv_small_wgrad_workspace(const pcnn_conv_desc* d) {
  return pcnn_conv_small_wgrad_eligible(d) ? (size_t)wgrad_splits(d) * d->kh * d->kw * d->Cin * d->Cout * sizeof(float) : 0;
}

static int small_fwd_impl(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                          const float* bn_shift, const float* residual, float* y, float* act_out, const pcnn_post_desc* post);

int pcnn_conv_small_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                        const float* bn_shift, const float* residual, float* y, float* act_out) {
  return small_fwd_impl(h, d, x, w, bias, bn_scale, bn_shift, residual, y, act_out, nullptr);
}

// The data gradient of a narrow layer with the activation backward of the layer that produced its input fused into the epilogue (include/pcnn.h
// pcnn_conv2d_dgrad_post): eligible when the data-gradient convolution takes the narrow route and every tensor can be accessed in whole 16-byte channel quads.
static bool quads_ok(const void* q, int ld) { return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0); }
extern "C" int pcnn_conv2d_dgrad_post_eligible(pcnn_handle h, const pcnn_conv_desc* dg, const float* dz, const float* residual, const float* dx, const pcnn_post_desc* post) {
  static const int on = getenv("PCNN_SMALL_POST") ? atoi(getenv("PCNN_SMALL_POST")) : 1;
  if (!on || !h || !dg || !post || !post->act_out) return 0;
  if (dg->pad_mode != PCNN_PAD_CONSTANT || dg->act != PCNN_ACT_LINEAR || dg->N < 1) return 0;
  if (!pcnn_conv_small_fwd_eligible(dg) || !pcnn_conv_fwd_takes_narrow_route(h, dg)) return 0;
  if (dg->Cout % 4 != 0) return 0;                               // (the gradient coming in may have any channel count: the tile loader handles it)
  if (post->act != PCNN_ACT_LINEAR && post->act != PCNN_ACT_RELU && post->act != PCNN_ACT_LEAKY_RELU && post->act != PCNN_ACT_TANH) return 0;
  return (quads_ok(dx, dg->ldy) && quads_ok(residual, dg->ld_res) && quads_ok(post->act_out, post->ld_act_out) && quads_ok(post->raw_out, post->ld_raw) && dz) ? 1 : 0;
}
extern "C" int pcnn_conv2d_dgrad_post(pcnn_handle h, const pcnn_conv_desc* dg, const float* dz, const float* w_flipped, const float* residual, float* dx,
                                      const pcnn_post_desc* post) {
  PCNN_REQUIRE(h, h && dg && dz && w_flipped && dx && post, "pcnn_conv2d_dgrad_post: null argument");
  PCNN_REQUIRE(h, pcnn_conv2d_dgrad_post_eligible(h, dg, dz, residual, dx, post), "pcnn_conv2d_dgrad_post: not eligible (ask pcnn_conv2d_dgrad_post_eligible first)");
  return small_fwd_impl(h, dg, dz, w_flipped, nullptr, nullptr, nullptr, residual, dx, nullptr, post);
}

static int small_fwd_impl(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale,
                          const float* bn_shift, const float* residual, float* y, float* act_out, const pcnn_post_desc* post) {
  const int CI = pad4(d->Cin), CO = pad4(d->Cout), taps = d->kh * d->kw;
  const size_t need = (size_t)taps * CI * CO * sizeof(float);
  if (h->scratch_bytes < need) {
    if (h->scratch) { pcnn_release(h, h->scratch); h->scratch = nullptr; h->scratch_bytes = 0; }
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
  if (post) {
    p.gact = post->act_out; p.ld_gact = post->ld_act_out; p.y2 = post->raw_out; p.ld_y2 = post->ld_raw; p.gmode = post->act;
    p.galpha = post->act == PCNN_ACT_LINEAR ? 1.f : (post->act == PCNN_ACT_RELU ? 0.f : post->act_alpha);
    if (post->dbias) {                                           // the workgroups' partial sums: nblk x CO floats in the handle's auxiliary scratch
      const size_t need = (size_t)nblk * CO * sizeof(float);
      if (h->aux_ws_bytes < need) {
        if (h->aux_ws) { pcnn_release(h, h->aux_ws); h->aux_ws = nullptr; h->aux_ws_bytes = 0; }
        if (hipMalloc(&h->aux_ws, need) != hipSuccess) PCNN_FAIL(h, "pcnn_conv2d_dgrad_post: cannot allocate %zu B of scratch", need);
        h->aux_ws_bytes = need;
      }
      p.bpart = static_cast<float*>(h->aux_ws);
    }
  }
  const bool ok = d->kh == 3 ? dispatch_ci<3>(h, p, nblk, CI, CO) : dispatch_ci<5>(h, p, nblk, CI, CO);
  PCNN_REQUIRE(h, ok, "pcnn_conv2d_fwd(small): no kernel for %d->%d", d->Cin, d->Cout);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_fwd(small)");
  if (post && post->dbias) {
    hipLaunchKernelGGL(small_post_bias_kernel, dim3((unsigned)d->Cout), dim3(256), 0, h->stream, p.bpart, nblk, CO, post->dbias);
    PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_dgrad_post(bias)");
  }
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
  const bool ok = d->kh == 3 ? wgrad_dispatch_ci<3>(h, p, CI, CO) : wgrad_dispatch_ci<5>(h, p, CI, CO);
  PCNN_REQUIRE(h, ok, "pcnn_conv2d_wgrad(small): no kernel for %d->%d", d->Cin, d->Cout);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(small)");
  const int nel = d->kh * d->kw * d->Cin * d->Cout;
  hipLaunchKernelGGL(small_wgrad_reduce_kernel, dim3(pcnn_cdiv(nel, 16)), dim3(256), 0, h->stream, p.part, dw, nel, p.S);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_wgrad(small reduce)");
  return 0;
}

extern "C" int pcnn_resnet3_fwd_eligible(pcnn_handle h, int C, int act) {
  static const int on = getenv("PCNN_FUSED_STAGE") ? atoi(getenv("PCNN_FUSED_STAGE")) : 1;
  return (h && on && (C == 4 || C == 8) && (act == PCNN_ACT_LINEAR || act == PCNN_ACT_RELU || act == PCNN_ACT_LEAKY_RELU)) ? 1 : 0;
}

extern "C" int pcnn_resnet3_fwd(pcnn_handle h, int N, int H, int W, int C, int act, float act_alpha, const float* x, const float* w0, const float* b0,
                                const float* w1, const float* b1, const float* w2, const float* b2, float* o0, float* a1, float* o1, float* y) {
  PCNN_REQUIRE(h, h && x && w0 && w1 && w2 && y, "pcnn_resnet3_fwd: null argument");
  PCNN_REQUIRE(h, pcnn_resnet3_fwd_eligible(h, C, act), "pcnn_resnet3_fwd: %d channels / activation %d are not eligible (ask pcnn_resnet3_fwd_eligible first)", C, act);
  PCNN_REQUIRE(h, N >= 1 && H >= 1 && W >= 1, "pcnn_resnet3_fwd: bad shape");
  for (const void* q : {(const void*)x, (const void*)o0, (const void*)a1, (const void*)o1, (const void*)y})
    PCNN_REQUIRE(h, (reinterpret_cast<uintptr_t>(q) & 15) == 0, "pcnn_resnet3_fwd: tensors must be 16-byte aligned");
  PCNN_REQUIRE(h, (int64_t)N * pcnn_cdiv(W, STW) * pcnn_cdiv(H, 8) < (1ll << 31), "pcnn_resnet3_fwd: grid too large");
  StageParams p;
  p.x = x; p.w0 = w0; p.w1 = w1; p.w2 = w2; p.b0 = b0; p.b1 = b1; p.b2 = b2; p.o0 = o0; p.a1 = a1; p.o1 = o1; p.y = y;
  p.N = N; p.H = H; p.W = W; p.tiles_x = 0; p.tiles_y = 0;
  p.slope = act == PCNN_ACT_LINEAR ? 1.f : (act == PCNN_ACT_RELU ? 0.f : act_alpha);
  static const int th = getenv("PCNN_STAGE_TH") ? atoi(getenv("PCNN_STAGE_TH")) : 16;
  const bool tall = th == 16 && H > 8;
  if (C == 4) { if (tall) launch_stage<4, 16>(h, p); else launch_stage<4, 8>(h, p); }
  else { if (tall) launch_stage<8, 16>(h, p); else launch_stage<8, 8>(h, p); }
  PCNN_CHECK_LAUNCH(h, "pcnn_resnet3_fwd");
  return 0;
}
