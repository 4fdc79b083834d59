// Shared epilogue of the forward convolution kernels (conv_fwd.hip, conv_fwd_split.hip): bias -> activation -> optional store of
// the pre-BN activation -> BN scale/shift -> residual -> y.  The accumulator layout of a 32x32 MFMA tile (lane = output channel,
// register = pixel) would store 4 bytes per lane into two different pixels per instruction - 64 scattered store instructions per
// thread, which is what bounds the small-channel layers.  The vector path therefore stages 4 output rows at a time (one per wave)
// in LDS as [pixel][channel] and writes them out as 16-byte pieces of contiguous NHWC rows (4 passes per 16-row tile).
#pragma once
#include "pcnn_internal.h"

struct ConvEpilogue {
  const float* bias; const float* bn_scale; const float* bn_shift; const float* res; float* y; float* act_out;
  int Ho, Wo, Cout, ldy, ld_res, ld_act, act; float alpha;
  int vec;                    // Cout, ldy, ld_res, ld_act multiples of 4 and y / res / act_out 16-byte aligned
  unsigned* absmax;           // optional: receives max|y| (float bits) - the weight gradient of the NEXT layer scales its x by it
};

static inline int conv_epilogue_vec_ok(int Cout, const float* y, int ldy, const float* res, int ld_res, const float* act_out, int ld_act) {
  if (Cout % 4 || ldy % 4 || (reinterpret_cast<uintptr_t>(y) & 15)) return 0;
  if (res && (ld_res % 4 || (reinterpret_cast<uintptr_t>(res) & 15))) return 0;
  if (act_out && (ld_act % 4 || (reinterpret_cast<uintptr_t>(act_out) & 15))) return 0;
  return 1;
}
static inline size_t conv_epilogue_lds_bytes(int Cout) { return (size_t)4 * 32 * ((Cout + 3) & ~3) * sizeof(float); }

__device__ __forceinline__ void conv_epilogue_absmax(unsigned* absmax, float v) {
  if (!absmax) return;                 // uniform
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  // tens of thousands of waves would otherwise serialise on one address: the running maximum only grows, so a wave whose value does not
  // exceed what it reads (possibly stale, i.e. lower) can skip the atomic; after the first few workgroups nearly all do
  if ((threadIdx.x & 63) == 0) {
    const unsigned bits = __float_as_uint(v <= 3.0e38f ? v : 3.0e38f);
    if (bits > __atomic_load_n(absmax, __ATOMIC_RELAXED)) atomicMax(absmax, bits);
  }
}

// tot[m][t][i]: wave w owns output rows y0 + w*MT + m; lane (col = lane & 31, half = lane >> 5): channel t*32 + col, pixel
// x0 + 8*(i>>2) + 4*half + (i&3).  `mul` scales the accumulator (1 for the fp32 kernel).  `stage`: >= conv_epilogue_lds_bytes()
// of LDS that no wave still reads (the function starts with a barrier).  All 256 threads must call it.
template <int MT, int NT, typename Acc>
__device__ __forceinline__ void conv_epilogue_store(const Acc (&tot)[MT][NT], float mul, const ConvEpilogue& e, int n, int y0, int x0,
                                                    float* __restrict__ stage) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
  float ymax = 0.f;
  if (!e.vec) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = t * 32 + col;
      if (co >= e.Cout) continue;
      const float bias = e.bias ? e.bias[co] : 0.f;
      const float sc = e.bn_scale ? e.bn_scale[co] : 1.f;
      const float sh = e.bn_scale ? e.bn_shift[co] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = y0 + wave * MT + m;
        if (oy >= e.Ho) continue;
        const int64_t rowpix = ((int64_t)n * e.Ho + oy) * e.Wo;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ox = x0 + 8 * (i >> 2) + 4 * half + (i & 3);
          if (ox >= e.Wo) continue;
          const int64_t pix = rowpix + ox;
          float v = pcnn_act(tot[m][t][i] * mul + bias, e.act, e.alpha);
          if (e.act_out) e.act_out[pix * e.ld_act + co] = v;
          v = v * sc + sh;
          if (e.res) v += e.res[pix * e.ld_res + co];
          e.y[pix * e.ldy + co] = v;
          ymax = fmaxf(ymax, fabsf(v));
        }
      }
    }
    conv_epilogue_absmax(e.absmax, ymax);
    return;
  }
  const int CP = e.Cout, Q = CP >> 2, nvec = 4 * 32 * Q;
  float bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bias[t] = (e.bias && t * 32 + col < e.Cout) ? e.bias[t * 32 + col] : 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    __syncthreads();                                   // the previous pass (or the K loop) no longer reads this LDS
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = t * 32 + col;
      if (co < e.Cout) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int px = 8 * (i >> 2) + 4 * half + (i & 3);
          stage[(wave * 32 + px) * CP + co] = pcnn_act(tot[m][t][i] * mul + bias[t], e.act, e.alpha);
        }
      }
    }
    __syncthreads();
    for (int v = tid; v < nvec; v += 256) {
      const int pp = v / Q, q4 = (v - pp * Q) << 2;    // staged pixel (row slot pp >> 5, column pp & 31), first channel
      const int oy = y0 + (pp >> 5) * MT + m, ox = x0 + (pp & 31);
      if (oy >= e.Ho || ox >= e.Wo) continue;
      float4 a = *reinterpret_cast<const float4*>(stage + pp * CP + q4);
      const int64_t pix = ((int64_t)n * e.Ho + oy) * e.Wo + ox;
      if (e.act_out) *reinterpret_cast<float4*>(e.act_out + pix * e.ld_act + q4) = a;
      if (e.bn_scale) {
        a.x = a.x * e.bn_scale[q4] + e.bn_shift[q4]; a.y = a.y * e.bn_scale[q4 + 1] + e.bn_shift[q4 + 1];
        a.z = a.z * e.bn_scale[q4 + 2] + e.bn_shift[q4 + 2]; a.w = a.w * e.bn_scale[q4 + 3] + e.bn_shift[q4 + 3];
      }
      if (e.res) {
        const float4 r = *reinterpret_cast<const float4*>(e.res + pix * e.ld_res + q4);
        a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
      }
      *reinterpret_cast<float4*>(e.y + pix * e.ldy + q4) = a;
      ymax = fmaxf(ymax, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    }
  }
  conv_epilogue_absmax(e.absmax, ymax);
}
