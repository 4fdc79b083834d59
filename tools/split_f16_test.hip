// Accuracy experiment: 32x32xK fp32 GEMM as (1) a v_mfma_f32_32x32x2_f32 chain, (2) a 3-product fp16 split
//   a = (a_hi + a_lo / S) / sa,  a*b ~ [a_hi b_hi + (a_hi b_lo + a_lo b_hi) / S] / (sa sb)      (S = 2^11, fp32 accumulate)
// on v_mfma_f32_32x32x16_f16, both against an fp64 host reference.  One wave.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k_f32(const float* A, const float* B, float* C, int K) {   // A [32][K], B [K][32]
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[(8 * (i >> 2) + 4 * h + (i & 3)) * 32 + r] = acc[i];
}

__global__ void k_split(const float* A, const float* B, float* C, int K, float sa, float sb, int blocked, float S) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 hh = {0}, cr = {0}, thh = {0}, tcr = {0};
  for (int k = 0; k < K; k += 16) {
    f16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const float a = A[r * K + k + 8 * h + j] * sa, b = B[(k + 8 * h + j) * 32 + r] * sb;
      const _Float16 a1 = (_Float16)a, b1 = (_Float16)b;
      ah[j] = a1; al[j] = (_Float16)((a - (float)a1) * S);
      bh[j] = b1; bl[j] = (_Float16)((b - (float)b1) * S);
    }
    hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hh, 0, 0, 0);
    cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cr, 0, 0, 0);
    cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cr, 0, 0, 0);
    if (blocked && ((k / 16) % 15 == 14)) {   // two-level summation like the fp32 kernel
      for (int i = 0; i < 16; ++i) { thh[i] += hh[i]; tcr[i] += cr[i]; hh[i] = 0; cr[i] = 0; }
    }
  }
  const float inv = 1.0f / (sa * sb);
  for (int i = 0; i < 16; ++i) C[(8 * (i >> 2) + 4 * h + (i & 3)) * 32 + r] = ((thh[i] + hh[i]) + (tcr[i] + cr[i]) * (1.0f / S)) * inv;
}

static double rel(const std::vector<float>& c, const std::vector<double>& ref) {
  double n = 0, d = 0;
  for (size_t i = 0; i < ref.size(); ++i) { n += (c[i] - ref[i]) * (c[i] - ref[i]); d += ref[i] * ref[i]; }
  return sqrt(n / d);
}

int main() {
  const int K = 7200;
  for (int dist = 0; dist < 3; ++dist) {
    std::vector<float> A(32 * K), B(K * 32), C(1024);
    srand(1 + dist);
    auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : A) v = dist == 0 ? rnd() : (dist == 1 ? rnd() * expf(8.f * rnd()) : rnd() * 1e-3f);
    for (auto& v : B) v = dist == 0 ? rnd() : (dist == 1 ? rnd() * expf(8.f * rnd()) : rnd() * 30.f);
    std::vector<double> ref(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 32 + j]; ref[i * 32 + j] = s; }
    float ma = 0, mb = 0;
    for (auto v : A) ma = fmaxf(ma, fabsf(v));
    for (auto v : B) mb = fmaxf(mb, fabsf(v));
    const float sa = exp2f(13.f - ceilf(log2f(ma))), sb = exp2f(13.f - ceilf(log2f(mb)));   // max |scaled| in [2^12, 2^13]
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 4096);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_f32<<<1, 64>>>(dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    const double e32 = rel(C, ref);
    k_split<<<1, 64>>>(dA, dB, dC, K, sa, sb, 0, 2048.f); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    const double es = rel(C, ref);
    k_split<<<1, 64>>>(dA, dB, dC, K, sa, sb, 1, 2048.f); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    const double esb = rel(C, ref);
    k_split<<<1, 64>>>(dA, dB, dC, K, sa, sb, 0, 1.0f); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    const double e1 = rel(C, ref);
    k_split<<<1, 64>>>(dA, dB, dC, K, sa, sb, 1, 1.0f); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    const double e1b = rel(C, ref);
    printf("   unscaled lo (S=1, one accumulator possible): %.3e | + blocked %.3e\n", e1, e1b);
    printf("dist %d (max|a| %.3g max|b| %.3g): fp32 MFMA chain %.3e | f16x3 split %.3e | f16x3 split + blocked sum %.3e\n", dist, ma, mb, e32, es, esb);
  }
  return 0;
}
