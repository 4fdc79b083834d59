// Micro-benchmark: achievable v_mfma_f32_32x32x2_f32 rate on this device (4 independent accumulators per wave, 1 or 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    k<<<blocks, 256>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 32.0 * 32 * 2 * 2;
    printf("%d WG/CU: %.3f ms, %.1f TFLOP/s\n", blocks_per_cu, ms, flop / ms / 1e9);
  }
  return 0;
}
