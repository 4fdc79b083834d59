// HBM-bound helpers of the hot path: epilogue backward + per-channel sums, tf.pad adjoint, axpby, per-sample and
// per-channel scaling, BC ring, input assembly, Jacobi sweep, Adam/SGD.  All are single-pass streaming kernels over
// NHWC data (channel index fastest -> coalesced), reductions are two-stage and deterministic (no float atomics).
#include "pcnn_internal.h"

namespace {

constexpr int CS_BLOCK = 256;
constexpr int CS_MAXBLK = 1024;

static int pow2_ge(int c) { int p = 1; while (p < c) p <<= 1; return p; }

// ---------------------------------------------------------------- epilogue backward + column sums
// thread -> (row r = tid / CP, channel c = tid % CP); K = 3 running sums per thread
__global__ __launch_bounds__(CS_BLOCK) void epilogue_bwd_kernel(int64_t npix, int C, int CP, const float* __restrict__ dy, int lddy,
                                                                const float* __restrict__ a, int lda, const float* __restrict__ bn_scale,
                                                                int act, float alpha, float* __restrict__ dz, int lddz,
                                                                float* __restrict__ partial /*[gridDim][3][C]*/, unsigned* __restrict__ absmax) {
  __shared__ float red[3][CS_BLOCK];
  const int tid = threadIdx.x, c = tid % CP, r = tid / CP, R = CS_BLOCK / CP;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, zmax = 0.f;
  if (c < C) {
    const float sc = bn_scale ? bn_scale[c] : 1.f;
    for (int64_t pix = (int64_t)blockIdx.x * R + r; pix < npix; pix += (int64_t)gridDim.x * R) {
      const float g = dy[pix * lddy + c];
      const float av = a ? a[pix * lda + c] : 0.f;
      const float z = g * sc * pcnn_act_grad_from_out(av, act, alpha);
      if (dz) dz[pix * lddz + c] = z;
      s0 += z; s1 += g * av; s2 += g; zmax = fmaxf(zmax, fabsf(z));
    }
  }
  if (absmax) {                        // max |dz| of the tensor: one atomic per wave (float bits of non-negative values order like integers)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o));
    if ((tid & 63) == 0) atomicMax(absmax, __float_as_uint(zmax <= 3.0e38f ? zmax : 3.0e38f));
  }
  red[0][tid] = s0; red[1][tid] = s1; red[2][tid] = s2;
  __syncthreads();
  if (r == 0 && c < C) {
    for (int k = 0; k < 3; ++k) {
      float s = 0.f;
      for (int q = 0; q < R; ++q) s += red[k][q * CP + c];
      partial[((int64_t)blockIdx.x * 3 + k) * C + c] = s;
    }
  }
}

// float4 variant (C, the channel strides and the base pointers multiples of 4 floats): thread -> (pixel slot r = tid / GQ,
// channel quad q = tid % GQ), no padding of the channel count to a power of two; same partial layout
__global__ __launch_bounds__(CS_BLOCK) void epilogue_bwd_vec4_kernel(int64_t npix, int C, const float* __restrict__ dy, int lddy,
                                                                     const float* __restrict__ a, int lda, const float* __restrict__ bn_scale,
                                                                     int act, float alpha, float* __restrict__ dz, int lddz,
                                                                     float* __restrict__ partial /*[gridDim][3][C]*/, unsigned* __restrict__ absmax) {
  __shared__ float red[3][CS_BLOCK * 4];
  float zmax = 0.f;
  const int GQ = C >> 2, R = CS_BLOCK / GQ;
  const int tid = threadIdx.x, r = tid / GQ, q = tid - r * GQ, c = q << 2;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < R) {
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = bn_scale ? bn_scale[c + j] : 1.f;
    for (int64_t pix = (int64_t)blockIdx.x * R + r; pix < npix; pix += (int64_t)gridDim.x * R) {
      const float4 g4 = *reinterpret_cast<const float4*>(dy + pix * lddy + c);
      const float4 a4 = a ? *reinterpret_cast<const float4*>(a + pix * lda + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w}, av[4] = {a4.x, a4.y, a4.z, a4.w};
      float z[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        z[j] = g[j] * sc[j] * pcnn_act_grad_from_out(av[j], act, alpha);
        s0[j] += z[j]; s1[j] += g[j] * av[j]; s2[j] += g[j]; zmax = fmaxf(zmax, fabsf(z[j]));
      }
      if (dz) *reinterpret_cast<float4*>(dz + pix * lddz + c) = make_float4(z[0], z[1], z[2], z[3]);
    }
  }
  if (absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o));
    if ((tid & 63) == 0) atomicMax(absmax, __float_as_uint(zmax <= 3.0e38f ? zmax : 3.0e38f));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][tid * 4 + j] = s0[j]; red[1][tid * 4 + j] = s1[j]; red[2][tid * 4 + j] = s2[j]; }
  __syncthreads();
  if (tid < C) {                       // channel tid: entries (r * GQ + tid / 4) * 4 + tid % 4 = r * C + tid
    for (int k = 0; k < 3; ++k) {
      float s = 0.f;
      for (int rr = 0; rr < R; ++rr) s += red[k][rr * C + tid];
      partial[((int64_t)blockIdx.x * 3 + k) * C + tid] = s;
    }
  }
}

// one workgroup per channel: the (<= 1024) block partials of each of the 3 sums are tree-reduced in a fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* o0, float* o1, float* o2) {
  __shared__ float red[3][256];
  const int c = blockIdx.x, t = threadIdx.x;
  float s[3] = {0.f, 0.f, 0.f};
  for (int b = t; b < nblk; b += 256)
    for (int k = 0; k < 3; ++k) s[k] += partial[((int64_t)b * 3 + k) * C + c];
  for (int k = 0; k < 3; ++k) red[k][t] = s[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (t < st)
      for (int k = 0; k < 3; ++k) red[k][t] += red[k][t + st];
    __syncthreads();
  }
  if (t == 0) {
    if (o0) o0[c] = red[0][0];
    if (o1) o1[c] = red[1][0];
    if (o2) o2[c] = red[2][0];
  }
}

static int colsum_blocks(int64_t npix, int CP) {
  const int R = CS_BLOCK / CP;
  int64_t nb = pcnn_cdiv64(npix, (int64_t)R * 16);
  if (nb > CS_MAXBLK) nb = CS_MAXBLK;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// ---------------------------------------------------------------- tf.pad adjoint
__global__ void pad_fold_kernel(int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode, const float* __restrict__ gp, int ldgp,
                                float* __restrict__ gx, int ldgx, int accumulate) {
  const int Hp = H + pt + pb, Wp = W + pl + pr;
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; int64_t r = i / C; const int x = r % W; r /= W; const int y = r % H; const int n = r / H;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + pt;
    xs[nx++] = x + pl;
    if (mode == PCNN_PAD_SYMMETRIC) {
      if (y < pt) ys[ny++] = pt - 1 - y;
      if (H - 1 - y < pb) ys[ny++] = pt + H + (H - 1 - y);
      if (x < pl) xs[nx++] = pl - 1 - x;
      if (W - 1 - x < pr) xs[nx++] = pl + W + (W - 1 - x);
    } else if (mode == PCNN_PAD_REFLECT) {
      if (y >= 1 && y <= pt) ys[ny++] = pt - y;
      if (H - 2 - y >= 0 && H - 2 - y < pb) ys[ny++] = pt + H + (H - 2 - y);
      if (x >= 1 && x <= pl) xs[nx++] = pl - x;
      if (W - 2 - x >= 0 && W - 2 - x < pr) xs[nx++] = pl + W + (W - 2 - x);
    }
    float s = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += gp[(((int64_t)n * Hp + ys[a]) * Wp + xs[b]) * ldgp + c];
    float* dst = &gx[(((int64_t)n * H + y) * W + x) * ldgx + c];
    *dst = accumulate ? *dst + s : s;
  }
}

// float4 form (C, the channel strides and the base pointers multiples of 4 floats): a quarter of the index arithmetic and memory instructions
__global__ void pad_fold_vec4_kernel(int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode, const float* __restrict__ gp, int ldgp,
                                     float* __restrict__ gx, int ldgx, int accumulate) {
  const int Hp = H + pt + pb, Wp = W + pl + pr, CV = C >> 2;
  const int64_t total = (int64_t)N * H * W * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) << 2; int64_t r = i / CV; const int x = r % W; r /= W; const int y = r % H; const int n = r / H;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + pt;
    xs[nx++] = x + pl;
    if (mode == PCNN_PAD_SYMMETRIC) {
      if (y < pt) ys[ny++] = pt - 1 - y;
      if (H - 1 - y < pb) ys[ny++] = pt + H + (H - 1 - y);
      if (x < pl) xs[nx++] = pl - 1 - x;
      if (W - 1 - x < pr) xs[nx++] = pl + W + (W - 1 - x);
    } else if (mode == PCNN_PAD_REFLECT) {
      if (y >= 1 && y <= pt) ys[ny++] = pt - y;
      if (H - 2 - y >= 0 && H - 2 - y < pb) ys[ny++] = pt + H + (H - 2 - y);
      if (x >= 1 && x <= pl) xs[nx++] = pl - x;
      if (W - 2 - x >= 0 && W - 2 - x < pr) xs[nx++] = pl + W + (W - 2 - x);
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        const float4 q = *reinterpret_cast<const float4*>(gp + (((int64_t)n * Hp + ys[a]) * Wp + xs[b]) * ldgp + c);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
      }
    float4* dst = reinterpret_cast<float4*>(gx + (((int64_t)n * H + y) * W + x) * ldgx + c);
    if (accumulate) { const float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *dst = s;
  }
}

// tf.pad adjoint + the PRODUCER's activation backward in one pass (pcnn_pad_fold_bwd_post; round 5): the gradient of a SYMMETRIC / REFLECT-padded
// convolution arrives on the padded domain; folding it back and multiplying by act'(a) of the layer that produced the convolution's input used to be two
// passes (pad_fold: read gp, write dx; epilogue_bwd: read dx and a, write dz).  Here: g = fold(gp) [+ add_to], raw_out = g (optional: a skip connection
// branches off), dz = g act'(a), bias-gradient partial sums per block (colsum_final_kernel combines them in a fixed order) - dx never reaches memory.
// float4 form only (the caller checks: C, the channel strides and the base pointers multiples of 4 floats); thread layout of epilogue_bwd_vec4_kernel.
__global__ __launch_bounds__(CS_BLOCK) void pad_fold_post_vec4_kernel(int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode,
                                                                      const float* __restrict__ gp, int ldgp, const float* __restrict__ add_to, int ldadd,
                                                                      const float* __restrict__ a, int lda, const float* __restrict__ bn_scale, int act, float alpha,
                                                                      float* __restrict__ raw, int ldraw, float* __restrict__ dz, int lddz,
                                                                      float* __restrict__ partial /*[gridDim][3][C]*/) {
  __shared__ float red[3][CS_BLOCK * 4];
  const int Hp = H + pt + pb, Wp = W + pl + pr;
  const int64_t npix = (int64_t)N * H * W;
  const int GQ = C >> 2, R = CS_BLOCK / GQ;
  const int tid = threadIdx.x, r = tid / GQ, q = tid - r * GQ, c = q << 2;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < R) {
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = bn_scale ? bn_scale[c + j] : 1.f;
    for (int64_t pix = (int64_t)blockIdx.x * R + r; pix < npix; pix += (int64_t)gridDim.x * R) {
      int64_t t = pix; const int x = t % W; t /= W; const int y = t % H; const int n = t / H;
      int ys[3], xs[3], ny = 0, nx = 0;
      ys[ny++] = y + pt;
      xs[nx++] = x + pl;
      if (mode == PCNN_PAD_SYMMETRIC) {
        if (y < pt) ys[ny++] = pt - 1 - y;
        if (H - 1 - y < pb) ys[ny++] = pt + H + (H - 1 - y);
        if (x < pl) xs[nx++] = pl - 1 - x;
        if (W - 1 - x < pr) xs[nx++] = pl + W + (W - 1 - x);
      } else if (mode == PCNN_PAD_REFLECT) {
        if (y >= 1 && y <= pt) ys[ny++] = pt - y;
        if (H - 2 - y >= 0 && H - 2 - y < pb) ys[ny++] = pt + H + (H - 2 - y);
        if (x >= 1 && x <= pl) xs[nx++] = pl - x;
        if (W - 2 - x >= 0 && W - 2 - x < pr) xs[nx++] = pl + W + (W - 2 - x);
      }
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int ia = 0; ia < ny; ++ia)                                 // the summation order of pad_fold_vec4_kernel: the folded values are the same bits
        for (int ib = 0; ib < nx; ++ib) {
          const float4 v = *reinterpret_cast<const float4*>(gp + (((int64_t)n * Hp + ys[ia]) * Wp + xs[ib]) * ldgp + c);
          g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
        }
      if (add_to) { const float4 o = *reinterpret_cast<const float4*>(add_to + pix * ldadd + c); g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
      if (raw) *reinterpret_cast<float4*>(raw + pix * ldraw + c) = g;
      const float4 a4 = *reinterpret_cast<const float4*>(a + pix * lda + c);
      const float gv[4] = {g.x, g.y, g.z, g.w}, av[4] = {a4.x, a4.y, a4.z, a4.w};
      float z[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {                                   // the arithmetic of epilogue_bwd_vec4_kernel, operation for operation
        z[j] = gv[j] * sc[j] * pcnn_act_grad_from_out(av[j], act, alpha);
        s0[j] += z[j]; s1[j] += gv[j] * av[j]; s2[j] += gv[j];
      }
      *reinterpret_cast<float4*>(dz + pix * lddz + c) = make_float4(z[0], z[1], z[2], z[3]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][tid * 4 + j] = s0[j]; red[1][tid * 4 + j] = s1[j]; red[2][tid * 4 + j] = s2[j]; }
  __syncthreads();
  if (tid < C) {                       // channel tid: entries (r * GQ + tid / 4) * 4 + tid % 4 = r * C + tid
    for (int k = 0; k < 3; ++k) {
      float s = 0.f;
      for (int rr = 0; rr < R; ++rr) s += red[k][rr * C + tid];
      partial[((int64_t)blockIdx.x * 3 + k) * C + tid] = s;
    }
  }
}

// ---------------------------------------------------------------- simple elementwise kernels
__global__ void axpby_kernel(int64_t npix, int C, float alpha, const float* __restrict__ x, int ldx, float beta, float* __restrict__ y, int ldy) {
  const int64_t total = npix * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; const int64_t pix = i / C;
    float* dst = &y[pix * ldy + c];
    const float v = alpha * x[pix * ldx + c];
    *dst = beta == 0.f ? v : v + beta * *dst;
  }
}

__global__ void assemble_input_kernel(int N, int H, int W, const float* __restrict__ rhs, int use_pos, float* __restrict__ out, int ldo) {
  const int64_t total = (int64_t)N * H * W;
  const float pi = 3.14159265358979323846f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = i % W; const int y = (i / W) % H;
    out[i * ldo] = rhs[i];
    if (use_pos) {
      // tf.linspace(0,1,n): start + i*delta in float32, delta = 1/(n-1)
      const float ly = H > 1 ? (float)y * (1.0f / (float)(H - 1)) : 0.f;
      const float lx = W > 1 ? (float)x * (1.0f / (float)(W - 1)) : 0.f;
      out[i * ldo + 1] = cosf(pi * ly);
      out[i * ldo + 2] = cosf(pi * lx);
    }
  }
}

__global__ void channel_scale_fwd_kernel(int N, int64_t hw, int C, const float* __restrict__ x, int ldx, const float* __restrict__ s,
                                         float* __restrict__ y, int ldy) {
  const int64_t total = (int64_t)N * hw * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; const int64_t pix = i / C; const int n = pix / hw;
    y[pix * ldy + c] = x[pix * ldx + c] * s[n * C + c];
  }
}

// dx = dy*s ; per-sample column sums of dy*x -> partial[n][blk][C]
__global__ __launch_bounds__(CS_BLOCK) void channel_scale_bwd_kernel(int64_t hw, int C, int CP, const float* __restrict__ x, int ldx,
                                                                     const float* __restrict__ s, const float* __restrict__ dy, int lddy,
                                                                     float* __restrict__ dx, int lddx, float* __restrict__ partial,
                                                                     int gmode, float galpha, float* __restrict__ bpartial) {
  // bpartial != nullptr (pcnn_channel_scale_bwd_post): x IS the saved activation output of the layer that produced it, so that layer's activation backward
  // rides along - dx = dy s act'(x), and the per-channel sums of dx (its bias gradient) are formed like the sums of ds
  __shared__ float red[CS_BLOCK];
  const int n = blockIdx.y, tid = threadIdx.x, c = tid % CP, r = tid / CP, R = CS_BLOCK / CP;
  float acc = 0.f, bacc = 0.f;
  if (c < C) {
    const float sv = s[n * C + c];
    for (int64_t q = (int64_t)blockIdx.x * R + r; q < hw; q += (int64_t)gridDim.x * R) {
      const int64_t pix = (int64_t)n * hw + q;
      const float g = dy[pix * lddy + c], xv = x[pix * ldx + c];
      acc += g * xv;
      float v = g * sv;
      if (bpartial) {
        v *= gmode == PCNN_ACT_TANH ? 1.f - xv * xv : (xv > 0.f ? 1.f : galpha);
        bacc += v;
      }
      dx[pix * lddx + c] = v;
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (r == 0 && c < C) {
    float t = 0.f;
    for (int q = 0; q < R; ++q) t += red[q * CP + c];
    partial[((int64_t)n * gridDim.x + blockIdx.x) * C + c] = t;
  }
  if (bpartial) {
    __syncthreads();
    red[tid] = bacc;
    __syncthreads();
    if (r == 0 && c < C) {
      float t = 0.f;
      for (int q = 0; q < R; ++q) t += red[q * CP + c];
      bpartial[((int64_t)n * gridDim.x + blockIdx.x) * C + c] = t;
    }
  }
}

// dbias[c] = sum over samples and blocks of the partial sums above, in a fixed order (sample-major); one thread per channel
__global__ void channel_scale_bias_final_kernel(const float* __restrict__ bpartial, int N, int nb, int C, float* __restrict__ dbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t = 0.f;
  for (int i = 0; i < N * nb; ++i) t += bpartial[(int64_t)i * C + c];
  dbias[c] = t;
}

__global__ void channel_scale_final_kernel(const float* __restrict__ partial, int N, int nblk, int C, float* __restrict__ ds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  float t = 0.f;
  for (int b = 0; b < nblk; ++b) t += partial[((int64_t)n * nblk + b) * C + c];
  ds[i] = t;
}

__global__ void sample_scale_fwd_kernel(int N, int64_t per, const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ y) {
  const int64_t total = (int64_t)N * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] * (1.0f + g[i / per]);
}

// one block per sample: dx = dy*(1+g), dg = sum dy*x
// dx = dy (1 + g[n]), dg[n] = sum dy x: SS_SPLIT workgroups per sample (a sample per workgroup left 8 of 256 CUs busy at 8 x 1024^2), each
// over one contiguous piece; partial sums in part[n][split], combined in a fixed order by sample_scale_final_kernel
constexpr int SS_SPLIT = 64;
__global__ __launch_bounds__(256) void sample_scale_bwd_kernel(int64_t per, const float* __restrict__ x, const float* __restrict__ g,
                                                               const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ part) {
  __shared__ float red[256];
  const int n = blockIdx.y, sp = blockIdx.x;
  const float gv = 1.0f + g[n];
  const int64_t piece = (per + SS_SPLIT - 1) / SS_SPLIT, q0 = (int64_t)sp * piece, q1 = q0 + piece < per ? q0 + piece : per;
  float acc = 0.f;
  for (int64_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
    const int64_t i = (int64_t)n * per + q;
    const float d = dy[i];
    acc += d * x[i];
    dx[i] = d * gv;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[n * SS_SPLIT + sp] = red[0];
}
__global__ void sample_scale_final_kernel(int N, const float* __restrict__ part, float* __restrict__ dg) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < SS_SPLIT; ++k) s += part[n * SS_SPLIT + k];
  dg[n] = s;
}

__global__ void bc_ring_fwd_kernel(int N, int H, int W, int neumann, const float* __restrict__ x, float* __restrict__ y) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = i % W; const int yy = (i / W) % H; const int64_t base = i - ((int64_t)yy * W + xx);
    const bool ring = yy == 0 || yy == H - 1 || xx == 0 || xx == W - 1;
    float v;
    if (!ring) v = x[i];
    else if (!neumann) v = 0.f;
    else {
      const int sy = yy == 0 ? 1 : (yy == H - 1 ? H - 2 : yy);
      const int sx = xx == 0 ? 1 : (xx == W - 1 ? W - 2 : xx);
      v = x[base + (int64_t)sy * W + sx];
    }
    y[i] = v;
  }
}

__global__ void bc_ring_bwd_kernel(int N, int H, int W, int neumann, const float* __restrict__ dy, float* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = i % W; const int yy = (i / W) % H; const int64_t base = i - ((int64_t)yy * W + xx);
    const bool ring = yy == 0 || yy == H - 1 || xx == 0 || xx == W - 1;
    float v = 0.f;
    if (!ring) {
      v = dy[i];
      if (neumann) {
        // interior cell (yy,xx) also feeds the ring cells that mirror it
        const int ny = (yy == 1) + (yy == H - 2), nx = (xx == 1) + (xx == W - 2);
        int ys[3], xs[3], cy = 0, cx = 0;
        ys[cy++] = yy; xs[cx++] = xx;
        if (yy == 1) ys[cy++] = 0;
        if (yy == H - 2) ys[cy++] = H - 1;
        if (xx == 1) xs[cx++] = 0;
        if (xx == W - 2) xs[cx++] = W - 1;
        (void)ny; (void)nx;
        v = 0.f;
        for (int a = 0; a < cy; ++a)
          for (int b = 0; b < cx; ++b) v += dy[base + (int64_t)ys[a] * W + xs[b]];
      }
    }
    dx[i] = v;
  }
}

// one weighted-Jacobi sweep of the 3x3 second-order Laplacian (layers/JacobiIterationLayer.py:43-54)
__global__ void jacobi_kernel(int N, int H, int W, const float* __restrict__ u, const float* __restrict__ rhs, const float* __restrict__ dx,
                              float* __restrict__ out) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = i % W; const int yy = (i / W) % H; const int n = i / ((int64_t)H * W);
    if (yy == 0 || yy == H - 1 || xx == 0 || xx == W - 1) { out[i] = u[i]; continue; }
    const float ay = 1.0f / (dx[2 * n] * dx[2 * n]), ax = 1.0f / (dx[2 * n + 1] * dx[2 * n + 1]);
    const float cr = ay * (u[i - W] + u[i + W]) + ax * (u[i - 1] + u[i + 1]);
    const float dinv = 1.0f / (-2.0f * ay - 2.0f * ax);
    out[i] = dinv * (rhs[i] - cr);
  }
}

// adjoint of the sweep w.r.t. u
__global__ void jacobi_bwd_kernel(int N, int H, int W, const float* __restrict__ dout, const float* __restrict__ dx, float* __restrict__ du) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = i % W; const int yy = (i / W) % H; const int n = i / ((int64_t)H * W);
    const float ay = 1.0f / (dx[2 * n] * dx[2 * n]), ax = 1.0f / (dx[2 * n + 1] * dx[2 * n + 1]);
    const float dinv = 1.0f / (-2.0f * ay - 2.0f * ax);
    auto interior = [&](int y, int x) { return y > 0 && y < H - 1 && x > 0 && x < W - 1; };
    float v = interior(yy, xx) ? 0.f : dout[i];
    if (yy - 1 >= 0 && interior(yy - 1, xx)) v -= dinv * ay * dout[i - W];
    if (yy + 1 < H && interior(yy + 1, xx)) v -= dinv * ay * dout[i + W];
    if (xx - 1 >= 0 && interior(yy, xx - 1)) v -= dinv * ax * dout[i - 1];
    if (xx + 1 < W && interior(yy, xx + 1)) v -= dinv * ax * dout[i + 1];
    du[i] = v;
  }
}

// vhat != nullptr: the AMSGrad variant (tf.keras Adam(amsgrad=True)): the denominator uses the running maximum of v
__global__ void adam_kernel(int64_t n, float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vhat, float lr_t, float beta1, float beta2, float eps, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    if (vhat) { vi = fmaxf(vhat[i], vi); vhat[i] = vi; }
    w[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

__global__ void sgd_kernel(int64_t n, float* __restrict__ w, const float* __restrict__ g, float lr, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) w[i] -= lr * gscale * g[i];
}

// tf.keras.optimizers.SGD with momentum (TF 2.4): v = momentum v - lr g;  w += v  (nesterov: w += momentum v - lr g)
__global__ void sgd_momentum_kernel(int64_t n, float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, float lr, float momentum,
                                    int nesterov, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = gscale * g[i];
    const float vi = momentum * v[i] - lr * gi;
    v[i] = vi;
    w[i] += nesterov ? momentum * vi - lr * gi : vi;
  }
}

__global__ void bn_fold_kernel(int n, const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = gamma[i] / sqrtf(var[i] + eps);
  scale[i] = s; shift[i] = beta[i] - mean[i] * s;
}

__global__ void bn_fold_bwd_kernel(int n, const float* s1, const float* s2, const float* mean, const float* var, float eps, float* dgamma, float* dbeta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dgamma[i] = (s1[i] - mean[i] * s2[i]) / sqrtf(var[i] + eps);
  dbeta[i] = s2[i];
}

static dim3 grid1d(int64_t total, int block = 256, int maxb = 8192) {
  int64_t b = pcnn_cdiv64(total, block);
  if (b > maxb) b = maxb;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace

extern "C" size_t pcnn_colsum_workspace(int C) { return (size_t)CS_MAXBLK * 3 * (size_t)(C > 0 ? C : 1) * sizeof(float); }

extern "C" int pcnn_conv2d_epilogue_bwd(pcnn_handle h, int64_t npix, int C, const float* dy, int lddy, const float* a, int lda,
                                        const float* bn_scale, int act, float act_alpha, float* dz, int lddz, float* dbias,
                                        float* dsum_dy_a, float* dsum_dy, void* workspace, size_t workspace_bytes) {
  return pcnn_conv2d_epilogue_bwd_absmax(h, npix, C, dy, lddy, a, lda, bn_scale, act, act_alpha, dz, lddz, dbias, dsum_dy_a, dsum_dy, nullptr, workspace,
                                         workspace_bytes);
}

extern "C" int pcnn_conv2d_epilogue_bwd_absmax(pcnn_handle h, int64_t npix, int C, const float* dy, int lddy, const float* a, int lda,
                                               const float* bn_scale, int act, float act_alpha, float* dz, int lddz, float* dbias,
                                               float* dsum_dy_a, float* dsum_dy, float* dz_absmax, void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, h && dy && workspace, "pcnn_conv2d_epilogue_bwd: null argument");
  PCNN_REQUIRE(h, C >= 1 && C <= CS_BLOCK, "pcnn_conv2d_epilogue_bwd: C=%d unsupported", C);
  PCNN_REQUIRE(h, a || act == PCNN_ACT_LINEAR, "pcnn_conv2d_epilogue_bwd: activation output required for a non-linear activation");
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_colsum_workspace(C), "pcnn_conv2d_epilogue_bwd: workspace too small");
  const int CP = pow2_ge(C);
  const int nb = colsum_blocks(npix, CP);
  float* partial = static_cast<float*>(workspace);
  const bool vec4 = C % 4 == 0 && lddy % 4 == 0 && (!a || lda % 4 == 0) && (!dz || lddz % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0;
  unsigned* amax = reinterpret_cast<unsigned*>(dz_absmax);
  if (amax) (void)hipMemsetAsync(amax, 0, sizeof(unsigned), h->stream);
  if (vec4)
    hipLaunchKernelGGL(epilogue_bwd_vec4_kernel, dim3(nb), dim3(CS_BLOCK), 0, h->stream, npix, C, dy, lddy, a, lda, bn_scale, act, act_alpha,
                       dz, lddz, partial, amax);
  else
    hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(nb), dim3(CS_BLOCK), 0, h->stream, npix, C, CP, dy, lddy, a, lda, bn_scale, act, act_alpha,
                       dz, lddz, partial, amax);
  PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_epilogue_bwd");
  if (dbias || dsum_dy_a || dsum_dy) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(256), 0, h->stream, partial, nb, C, dbias, dsum_dy_a, dsum_dy);
    PCNN_CHECK_LAUNCH(h, "pcnn_conv2d_epilogue_bwd(final)");
  }
  return 0;
}

extern "C" int pcnn_pad_fold_bwd(pcnn_handle h, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int pad_mode, const float* gp,
                                 int ldgp, float* gx, int ldgx, int accumulate) {
  PCNN_REQUIRE(h, h && gp && gx, "pcnn_pad_fold_bwd: null argument");
  PCNN_REQUIRE(h, pad_mode >= 0 && pad_mode <= 2 && pt >= 0 && pb >= 0 && pl >= 0 && pr >= 0, "pcnn_pad_fold_bwd: bad padding");
  if (C % 4 == 0 && ldgp % 4 == 0 && ldgx % 4 == 0 && ((reinterpret_cast<uintptr_t>(gp) | reinterpret_cast<uintptr_t>(gx)) & 15) == 0)
    hipLaunchKernelGGL(pad_fold_vec4_kernel, grid1d((int64_t)N * H * W * (C / 4)), dim3(256), 0, h->stream, N, H, W, C, pt, pb, pl, pr, pad_mode, gp, ldgp, gx, ldgx,
                       accumulate);
  else
    hipLaunchKernelGGL(pad_fold_kernel, grid1d((int64_t)N * H * W * C), dim3(256), 0, h->stream, N, H, W, C, pt, pb, pl, pr, pad_mode, gp, ldgp,
                       gx, ldgx, accumulate);
  PCNN_CHECK_LAUNCH(h, "pcnn_pad_fold_bwd");
  return 0;
}

extern "C" int pcnn_pad_fold_bwd_post_eligible(int C, int ldgp, int ld_add, const void* gp, const void* add_to, const pcnn_post_desc* post, int lddz, const void* dz) {
  if (!post || !post->act_out || !gp || !dz) return 0;
  if (C < 4 || C > CS_BLOCK || (C & 3) || (ldgp & 3) || (lddz & 3) || (post->ld_act_out & 3)) return 0;
  if (add_to && (ld_add & 3)) return 0;
  if (post->raw_out && (post->ld_raw & 3)) return 0;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(gp) | reinterpret_cast<uintptr_t>(add_to) | reinterpret_cast<uintptr_t>(dz) |
                         reinterpret_cast<uintptr_t>(post->act_out) | reinterpret_cast<uintptr_t>(post->raw_out);
  return (bits & 15) == 0 ? 1 : 0;
}

extern "C" int pcnn_pad_fold_bwd_post(pcnn_handle h, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int pad_mode, const float* gp, int ldgp,
                                      const float* add_to, int ld_add, const pcnn_post_desc* post, const float* bn_scale, float* dsum_dy_a, float* dsum_dy,
                                      float* dz, int lddz, void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, h && gp && post && dz && workspace, "pcnn_pad_fold_bwd_post: null argument");
  PCNN_REQUIRE(h, pad_mode >= 0 && pad_mode <= 2 && pt >= 0 && pb >= 0 && pl >= 0 && pr >= 0, "pcnn_pad_fold_bwd_post: bad padding");
  PCNN_REQUIRE(h, pcnn_pad_fold_bwd_post_eligible(C, ldgp, ld_add, gp, add_to, post, lddz, dz),
               "pcnn_pad_fold_bwd_post: shape not eligible (ask pcnn_pad_fold_bwd_post_eligible first: channels and strides multiples of 4, 16-byte aligned tensors)");
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_colsum_workspace(C), "pcnn_pad_fold_bwd_post: workspace too small");
  const int64_t npix = (int64_t)N * H * W;
  const int nb = colsum_blocks(npix, pow2_ge(C));
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(pad_fold_post_vec4_kernel, dim3(nb), dim3(CS_BLOCK), 0, h->stream, N, H, W, C, pt, pb, pl, pr, pad_mode, gp, ldgp, add_to, ld_add,
                     post->act_out, post->ld_act_out, bn_scale, post->act, post->act_alpha, post->raw_out, post->ld_raw, dz, lddz, partial);
  PCNN_CHECK_LAUNCH(h, "pcnn_pad_fold_bwd_post");
  if (post->dbias || dsum_dy_a || dsum_dy) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(256), 0, h->stream, partial, nb, C, post->dbias, dsum_dy_a, dsum_dy);
    PCNN_CHECK_LAUNCH(h, "pcnn_pad_fold_bwd_post(final)");
  }
  return 0;
}

// strided sub-sampling y[n,i,j,c] = x[n, i*s, j*s, c] and its adjoint (zero-fill + scatter): a strided convolution of the reference is the
// stride-1 fused pad+conv followed by this (utils/apply_advanced_padding_and_call_conv_layer.py pads the same whatever the stride)
__global__ void subsample_kernel(int N, int H, int W, int C, int s, int Ho, int Wo, const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                 int adjoint) {
  if (!adjoint) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int c = i % C; int64_t r = i / C; const int ox = r % Wo; r /= Wo; const int oy = r % Ho; const int n = r / Ho;
      y[(((int64_t)n * Ho + oy) * Wo + ox) * ldy + c] = x[(((int64_t)n * H + (int64_t)oy * s) * W + (int64_t)ox * s) * ldx + c];
    }
  } else {   // x: coarse gradient (N,Ho,Wo,C), y: full-resolution gradient (N,H,W,C)
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int c = i % C; int64_t r = i / C; const int xx = r % W; r /= W; const int yy = r % H; const int n = r / H;
      const bool hit = (yy % s == 0) && (xx % s == 0);
      y[(((int64_t)n * H + yy) * W + xx) * ldy + c] = hit ? x[(((int64_t)n * Ho + yy / s) * Wo + xx / s) * ldx + c] : 0.f;
    }
  }
}

extern "C" int pcnn_subsample(pcnn_handle h, int N, int H, int W, int C, int stride, const float* x, int ldx, float* y, int ldy, int adjoint) {
  PCNN_REQUIRE(h, h && x && y && stride >= 1 && N >= 1 && H >= 1 && W >= 1 && C >= 1, "pcnn_subsample: bad argument");
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  hipLaunchKernelGGL(subsample_kernel, grid1d((int64_t)N * (adjoint ? (int64_t)H * W : (int64_t)Ho * Wo) * C), dim3(256), 0, h->stream, N, H, W, C, stride, Ho, Wo,
                     x, ldx, y, ldy, adjoint);
  PCNN_CHECK_LAUNCH(h, "pcnn_subsample");
  return 0;
}

extern "C" int pcnn_axpby(pcnn_handle h, int64_t npix, int C, float alpha, const float* x, int ldx, float beta, float* y, int ldy) {
  PCNN_REQUIRE(h, h && x && y, "pcnn_axpby: null argument");
  hipLaunchKernelGGL(axpby_kernel, grid1d(npix * C), dim3(256), 0, h->stream, npix, C, alpha, x, ldx, beta, y, ldy);
  PCNN_CHECK_LAUNCH(h, "pcnn_axpby");
  return 0;
}

extern "C" int pcnn_assemble_input(pcnn_handle h, int N, int H, int W, const float* rhs, int use_pos, float* out, int ldo) {
  PCNN_REQUIRE(h, h && rhs && out && ldo >= (use_pos ? 3 : 1), "pcnn_assemble_input: bad argument");
  hipLaunchKernelGGL(assemble_input_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, rhs, use_pos, out, ldo);
  PCNN_CHECK_LAUNCH(h, "pcnn_assemble_input");
  return 0;
}

extern "C" int pcnn_channel_scale_fwd(pcnn_handle h, int N, int64_t hw, int C, const float* x, int ldx, const float* s, float* y, int ldy) {
  PCNN_REQUIRE(h, h && x && s && y, "pcnn_channel_scale_fwd: null argument");
  hipLaunchKernelGGL(channel_scale_fwd_kernel, grid1d((int64_t)N * hw * C), dim3(256), 0, h->stream, N, hw, C, x, ldx, s, y, ldy);
  PCNN_CHECK_LAUNCH(h, "pcnn_channel_scale_fwd");
  return 0;
}

static int cs_blocks(int64_t hw, int CP) {
  int64_t nb = pcnn_cdiv64(hw, (int64_t)(CS_BLOCK / CP) * 16);
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" size_t pcnn_channel_scale_workspace(int N, int64_t hw, int C) { return (size_t)N * 256 * (size_t)C * sizeof(float); }

extern "C" int pcnn_channel_scale_bwd(pcnn_handle h, int N, int64_t hw, int C, const float* x, int ldx, const float* s, const float* dy, int lddy,
                                      float* dx, int lddx, float* ds, void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, h && x && s && dy && dx && ds && workspace, "pcnn_channel_scale_bwd: null argument");
  PCNN_REQUIRE(h, C >= 1 && C <= CS_BLOCK, "pcnn_channel_scale_bwd: C=%d unsupported", C);
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_channel_scale_workspace(N, hw, C), "pcnn_channel_scale_bwd: workspace too small");
  const int CP = pow2_ge(C), nb = cs_blocks(hw, CP);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(channel_scale_bwd_kernel, dim3(nb, N), dim3(CS_BLOCK), 0, h->stream, hw, C, CP, x, ldx, s, dy, lddy, dx, lddx, partial, 0, 1.f, (float*)nullptr);
  PCNN_CHECK_LAUNCH(h, "pcnn_channel_scale_bwd");
  hipLaunchKernelGGL(channel_scale_final_kernel, dim3(pcnn_cdiv(N * C, 128)), dim3(128), 0, h->stream, partial, N, nb, C, ds);
  PCNN_CHECK_LAUNCH(h, "pcnn_channel_scale_bwd(final)");
  return 0;
}

// ... with the activation backward of the layer that produced x (x = its saved activation output): dz = dy s act'(x) instead of dx, dbias = sum dz.
// workspace: TWICE pcnn_channel_scale_workspace(N, hw, C).  dz is what pcnn_channel_scale_bwd + pcnn_conv2d_epilogue_bwd give, bit for bit.
extern "C" int pcnn_channel_scale_bwd_post(pcnn_handle h, int N, int64_t hw, int C, const float* x, int ldx, const float* s, const float* dy, int lddy,
                                           float* dz, int lddz, float* ds, int act, float act_alpha, float* dbias, void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, h && x && s && dy && dz && ds && workspace, "pcnn_channel_scale_bwd_post: null argument");
  PCNN_REQUIRE(h, C >= 1 && C <= CS_BLOCK, "pcnn_channel_scale_bwd_post: C=%d unsupported", C);
  PCNN_REQUIRE(h, act == PCNN_ACT_LINEAR || act == PCNN_ACT_RELU || act == PCNN_ACT_LEAKY_RELU || act == PCNN_ACT_TANH, "pcnn_channel_scale_bwd_post: activation %d", act);
  PCNN_REQUIRE(h, workspace_bytes >= 2 * pcnn_channel_scale_workspace(N, hw, C), "pcnn_channel_scale_bwd_post: workspace too small");
  const int CP = pow2_ge(C), nb = cs_blocks(hw, CP);
  float* partial = static_cast<float*>(workspace);
  float* bpartial = partial + (size_t)N * 256 * C;
  const float galpha = act == PCNN_ACT_LINEAR ? 1.f : (act == PCNN_ACT_RELU ? 0.f : act_alpha);
  hipLaunchKernelGGL(channel_scale_bwd_kernel, dim3(nb, N), dim3(CS_BLOCK), 0, h->stream, hw, C, CP, x, ldx, s, dy, lddy, dz, lddz, partial, act, galpha, bpartial);
  PCNN_CHECK_LAUNCH(h, "pcnn_channel_scale_bwd_post");
  hipLaunchKernelGGL(channel_scale_final_kernel, dim3(pcnn_cdiv(N * C, 128)), dim3(128), 0, h->stream, partial, N, nb, C, ds);
  if (dbias) hipLaunchKernelGGL(channel_scale_bias_final_kernel, dim3(pcnn_cdiv(C, 64)), dim3(64), 0, h->stream, bpartial, N, nb, C, dbias);
  PCNN_CHECK_LAUNCH(h, "pcnn_channel_scale_bwd_post(final)");
  return 0;
}

extern "C" int pcnn_sample_scale_fwd(pcnn_handle h, int N, int64_t per, const float* x, const float* g, float* y) {
  PCNN_REQUIRE(h, h && x && g && y, "pcnn_sample_scale_fwd: null argument");
  hipLaunchKernelGGL(sample_scale_fwd_kernel, grid1d((int64_t)N * per), dim3(256), 0, h->stream, N, per, x, g, y);
  PCNN_CHECK_LAUNCH(h, "pcnn_sample_scale_fwd");
  return 0;
}

extern "C" int pcnn_sample_scale_bwd(pcnn_handle h, int N, int64_t per, const float* x, const float* g, const float* dy, float* dx, float* dg) {
  PCNN_REQUIRE(h, h && x && g && dy && dx && dg, "pcnn_sample_scale_bwd: null argument");
  const size_t need = (size_t)N * SS_SPLIT * sizeof(float);
  if (h->aux_ws_bytes < need) {                              // handle-owned scratch (shared with the two-pass resize; one stream per handle)
    if (h->aux_ws) { pcnn_release(h, h->aux_ws); h->aux_ws = nullptr; h->aux_ws_bytes = 0; }
    const size_t cap = need < (1u << 20) ? (1u << 20) : need;
    if (hipMalloc(&h->aux_ws, cap) != hipSuccess) PCNN_FAIL(h, "pcnn_sample_scale_bwd: cannot allocate %zu B of scratch", cap);
    h->aux_ws_bytes = cap;
  }
  float* part = static_cast<float*>(h->aux_ws);
  hipLaunchKernelGGL(sample_scale_bwd_kernel, dim3(SS_SPLIT, N), dim3(256), 0, h->stream, per, x, g, dy, dx, part);
  hipLaunchKernelGGL(sample_scale_final_kernel, dim3((N + 63) / 64), dim3(64), 0, h->stream, N, part, dg);
  PCNN_CHECK_LAUNCH(h, "pcnn_sample_scale_bwd");
  return 0;
}

extern "C" int pcnn_bc_ring_fwd(pcnn_handle h, int N, int H, int W, int neumann, const float* x, float* y) {
  PCNN_REQUIRE(h, h && x && y, "pcnn_bc_ring_fwd: null argument");
  PCNN_REQUIRE(h, H >= 3 && W >= 3, "pcnn_bc_ring_fwd: grid %dx%d too small", H, W);
  hipLaunchKernelGGL(bc_ring_fwd_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, neumann, x, y);
  PCNN_CHECK_LAUNCH(h, "pcnn_bc_ring_fwd");
  return 0;
}

extern "C" int pcnn_bc_ring_bwd(pcnn_handle h, int N, int H, int W, int neumann, const float* dy, float* dx) {
  PCNN_REQUIRE(h, h && dy && dx, "pcnn_bc_ring_bwd: null argument");
  PCNN_REQUIRE(h, H >= 3 && W >= 3, "pcnn_bc_ring_bwd: grid %dx%d too small", H, W);
  hipLaunchKernelGGL(bc_ring_bwd_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, neumann, dy, dx);
  PCNN_CHECK_LAUNCH(h, "pcnn_bc_ring_bwd");
  return 0;
}

extern "C" int pcnn_jacobi_sweep(pcnn_handle h, int N, int H, int W, const float* u, const float* rhs, const float* dx, float* out) {
  PCNN_REQUIRE(h, h && u && rhs && dx && out && u != out, "pcnn_jacobi_sweep: bad argument");
  hipLaunchKernelGGL(jacobi_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, u, rhs, dx, out);
  PCNN_CHECK_LAUNCH(h, "pcnn_jacobi_sweep");
  return 0;
}

extern "C" int pcnn_jacobi_sweep_bwd(pcnn_handle h, int N, int H, int W, const float* dout, const float* dx, float* du) {
  PCNN_REQUIRE(h, h && dout && dx && du && dout != du, "pcnn_jacobi_sweep_bwd: bad argument");
  hipLaunchKernelGGL(jacobi_bwd_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, dout, dx, du);
  PCNN_CHECK_LAUNCH(h, "pcnn_jacobi_sweep_bwd");
  return 0;
}

extern "C" int pcnn_adam_step(pcnn_handle h, int64_t n, float* w, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                              float eps, int step, float grad_scale) {
  return pcnn_adam_amsgrad_step(h, n, w, g, m, v, nullptr, lr, beta1, beta2, eps, step, grad_scale);
}

extern "C" int pcnn_adam_amsgrad_step(pcnn_handle h, int64_t n, float* w, const float* g, float* m, float* v, float* vhat, float lr, float beta1,
                                      float beta2, float eps, int step, float grad_scale) {
  PCNN_REQUIRE(h, h && w && g && m && v && step >= 1, "pcnn_adam_step: bad argument");
  // tf.keras Adam: lr_t = lr * sqrt(1-b2^t)/(1-b1^t); w -= lr_t * m / (sqrt(v) + eps)
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
  hipLaunchKernelGGL(adam_kernel, grid1d(n), dim3(256), 0, h->stream, n, w, g, m, v, vhat, (float)lr_t, beta1, beta2, eps, grad_scale);
  PCNN_CHECK_LAUNCH(h, "pcnn_adam_step");
  return 0;
}

extern "C" int pcnn_sgd_step(pcnn_handle h, int64_t n, float* w, const float* g, float lr, float grad_scale) {
  PCNN_REQUIRE(h, h && w && g, "pcnn_sgd_step: null argument");
  hipLaunchKernelGGL(sgd_kernel, grid1d(n), dim3(256), 0, h->stream, n, w, g, lr, grad_scale);
  PCNN_CHECK_LAUNCH(h, "pcnn_sgd_step");
  return 0;
}

extern "C" int pcnn_sgd_momentum_step(pcnn_handle h, int64_t n, float* w, const float* g, float* velocity, float lr, float momentum, int nesterov,
                                      float grad_scale) {
  PCNN_REQUIRE(h, h && w && g && velocity, "pcnn_sgd_momentum_step: null argument");
  hipLaunchKernelGGL(sgd_momentum_kernel, grid1d(n), dim3(256), 0, h->stream, n, w, g, velocity, lr, momentum, nesterov, grad_scale);
  PCNN_CHECK_LAUNCH(h, "pcnn_sgd_momentum_step");
  return 0;
}

extern "C" int pcnn_bn_fold(pcnn_handle h, int n, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                            float* scale, float* shift) {
  PCNN_REQUIRE(h, h && gamma && beta && mean && var && scale && shift && n >= 0, "pcnn_bn_fold: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(pcnn_cdiv(n, 256)), dim3(256), 0, h->stream, n, gamma, beta, mean, var, eps, scale, shift);
  PCNN_CHECK_LAUNCH(h, "pcnn_bn_fold");
  return 0;
}

extern "C" int pcnn_bn_fold_bwd(pcnn_handle h, int n, const float* s_dy_a, const float* s_dy, const float* mean, const float* var, float eps,
                                float* dgamma, float* dbeta) {
  PCNN_REQUIRE(h, h && s_dy_a && s_dy && mean && var && dgamma && dbeta && n >= 0, "pcnn_bn_fold_bwd: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3(pcnn_cdiv(n, 256)), dim3(256), 0, h->stream, n, s_dy_a, s_dy, mean, var, eps, dgamma, dbeta);
  PCNN_CHECK_LAUNCH(h, "pcnn_bn_fold_bwd");
  return 0;
}

// ---- training-mode BatchNormalization (fused semantics: biased variance normalises, unbiased variance feeds the moving average)
namespace {
__global__ void channel_affine_kernel(int64_t npix, int C, const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                                      const float* __restrict__ shift, const float* __restrict__ res, int ld_res, float* __restrict__ y, int ldy) {
  const int64_t total = npix * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; const int64_t pix = i / C;
    float v = x[pix * ldx + c] * scale[c] + shift[c];
    if (res) v += res[pix * ld_res + c];
    y[pix * ldy + c] = v;
  }
}

__global__ void bn_train_finalize_kernel(int C, float inv_n, float unbias, const float* sum_a, const float* sum_a2, const float* gamma, const float* beta,
                                         float eps, float momentum, float* moving_mean, float* moving_var, float* mean, float* inv_std, float* scale,
                                         float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float m = sum_a[c] * inv_n;
  float v = sum_a2[c] * inv_n - m * m;
  if (v < 0.f) v = 0.f;
  const float is = 1.0f / sqrtf(v + eps);
  mean[c] = m; inv_std[c] = is;
  scale[c] = gamma[c] * is; shift[c] = beta[c] - m * gamma[c] * is;
  moving_mean[c] = moving_mean[c] * momentum + m * (1.0f - momentum);
  moving_var[c] = moving_var[c] * momentum + v * unbias * (1.0f - momentum);
}

__global__ void bn_train_bwd_finalize_kernel(int C, float inv_n, const float* s_dy_a, const float* s_dy, const float* mean, const float* inv_std,
                                             float* dgamma, float* dbeta, float* c1, float* c2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float dg = inv_std[c] * (s_dy_a[c] - mean[c] * s_dy[c]);   // sum dy * xhat
  dgamma[c] = dg; dbeta[c] = s_dy[c];
  c1[c] = s_dy[c] * inv_n; c2[c] = dg * inv_n;
}

__global__ void bn_train_bwd_kernel(int64_t npix, int C, const float* __restrict__ dy, int lddy, const float* __restrict__ a, int lda,
                                    const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ inv_std,
                                    const float* __restrict__ c1, const float* __restrict__ c2, float* __restrict__ da, int ldda) {
  const int64_t total = npix * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; const int64_t pix = i / C;
    const float xh = (a[pix * lda + c] - mean[c]) * inv_std[c];
    da[pix * ldda + c] = scale[c] * (dy[pix * lddy + c] - c1[c] - xh * c2[c]);
  }
}
}  // namespace

extern "C" int pcnn_channel_affine(pcnn_handle h, int64_t npix, int C, const float* x, int ldx, const float* scale, const float* shift,
                                   const float* residual, int ld_res, float* y, int ldy) {
  PCNN_REQUIRE(h, h && x && scale && shift && y, "pcnn_channel_affine: null argument");
  hipLaunchKernelGGL(channel_affine_kernel, grid1d(npix * C), dim3(256), 0, h->stream, npix, C, x, ldx, scale, shift, residual, ld_res, y, ldy);
  PCNN_CHECK_LAUNCH(h, "pcnn_channel_affine");
  return 0;
}

extern "C" int pcnn_bn_train_finalize(pcnn_handle h, int C, int64_t npix, const float* sum_a, const float* sum_a2, const float* gamma, const float* beta,
                                      float eps, float momentum, float* moving_mean, float* moving_var, float* mean, float* inv_std, float* scale,
                                      float* shift) {
  PCNN_REQUIRE(h, h && sum_a && sum_a2 && gamma && beta && moving_mean && moving_var && mean && inv_std && scale && shift && npix > 0,
               "pcnn_bn_train_finalize: bad argument");
  const float unbias = npix > 1 ? (float)((double)npix / (double)(npix - 1)) : 1.0f;
  hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(pcnn_cdiv(C, 64)), dim3(64), 0, h->stream, C, (float)(1.0 / (double)npix), unbias, sum_a, sum_a2, gamma,
                     beta, eps, momentum, moving_mean, moving_var, mean, inv_std, scale, shift);
  PCNN_CHECK_LAUNCH(h, "pcnn_bn_train_finalize");
  return 0;
}

extern "C" int pcnn_bn_train_bwd(pcnn_handle h, int64_t npix, int C, const float* dy, int lddy, const float* a, int lda, const float* scale,
                                 const float* mean, const float* inv_std, const float* s_dy_a, const float* s_dy, float* dgamma, float* dbeta,
                                 float* scratch_2C, float* da, int ldda) {
  PCNN_REQUIRE(h, h && dy && a && scale && mean && inv_std && s_dy_a && s_dy && dgamma && dbeta && scratch_2C && da, "pcnn_bn_train_bwd: null argument");
  float* c1 = scratch_2C; float* c2 = scratch_2C + C;
  hipLaunchKernelGGL(bn_train_bwd_finalize_kernel, dim3(pcnn_cdiv(C, 64)), dim3(64), 0, h->stream, C, (float)(1.0 / (double)npix), s_dy_a, s_dy, mean, inv_std,
                     dgamma, dbeta, c1, c2);
  PCNN_CHECK_LAUNCH(h, "pcnn_bn_train_bwd(finalize)");
  hipLaunchKernelGGL(bn_train_bwd_kernel, grid1d(npix * C), dim3(256), 0, h->stream, npix, C, dy, lddy, a, lda, scale, mean, inv_std, c1, c2, da, ldda);
  PCNN_CHECK_LAUNCH(h, "pcnn_bn_train_bwd");
  return 0;
}
