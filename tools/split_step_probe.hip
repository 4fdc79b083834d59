// Micro-benchmark for the K-step of conv_fwd_split_kernel: 12 x v_mfma_f32_32x32x16_f16 per step, optionally with the step's
// 8 ds_read_b128 (A hi/lo fragments) and 2 global b128 loads (B hi/lo fragments).  Shows which pipe bounds the step.
// hipcc -O3 --offload-arch=gfx950 tools/split_step_probe.hip -o tools/split_step_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSR, int GLB, int MT>
__global__ __launch_bounds__(256, MT == 4 ? 2 : 1) void probe(const f16x8* __restrict__ w, float* out, int iters, int nsteps_w) {
  extern __shared__ f16x8 planes[];            // 2 planes of 2048 pixels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < 4096; i += 256) { f16x8 v; for (int j = 0; j < 8; ++j) v[j] = (_Float16)(0.37f * (float)(((i * 8 + j) * 2654435761u >> 20) & 1023) - 180.f); planes[i] = v; }
  __syncthreads();
  const f16x8* hi = planes; const f16x8* lo = planes + 2048;
  f32x16 acc[MT];
  for (int m = 0; m < MT; ++m) for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  f16x8 ah[MT], al[MT], bh, bl, nah[MT], nal[MT], nbh, nbl;
  for (int m = 0; m < MT; ++m) { ah[m] = hi[col + m * 46]; al[m] = lo[col + m * 46]; }
  const f16x8* wl = w + half * 32 + col;
  bh = wl[0]; bl = wl[64 * nsteps_w];
  int tw = 0;
  for (int t = 0; t < iters; ++t) {
    if (GLB) { tw = tw + 1 == nsteps_w ? 0 : tw + 1; nbh = wl[tw * 64]; nbl = wl[64 * nsteps_w + tw * 64]; } else { nbh = bh; nbl = bl; }
    const int off = wave * MT * 46 + ((t * 5) & 511) + half * 3;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (LDSR) { nah[m] = hi[col + off + m * 46]; nal[m] = lo[col + off + m * 46]; } else { nah[m] = al[m]; nal[m] = ah[m]; }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) { ah[m] = nah[m]; al[m] = nal[m]; }
    bh = nbh; bl = nbl;
  }
  float s = 0.f;
  for (int m = 0; m < MT; ++m) for (int i = 0; i < 16; ++i) s += acc[m][i];
  out[blockIdx.x * 256 + tid] = s;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
// same work per iteration as TWO steps of probe<> (K = 32), issued as v_mfma_f32_16x16x32_f16: 2 x 2 blocks of 16 x 16 per M-tile
template <int LDSR, int GLB, int MT>
__global__ __launch_bounds__(256, 2) void probe16(const f16x8* __restrict__ w, float* out, int iters, int nsteps_w) {
  extern __shared__ f16x8 planes[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kg = lane >> 4;
  for (int i = tid; i < 4096; i += 256) { f16x8 v; for (int j = 0; j < 8; ++j) v[j] = (_Float16)(0.37f * (float)(((i * 8 + j) * 2654435761u >> 20) & 1023) - 180.f); planes[i] = v; }
  __syncthreads();
  const f16x8* hi = planes; const f16x8* lo = planes + 2048;
  f32x4v acc[MT][2][2];
  for (int m = 0; m < MT; ++m) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) acc[m][a][b][i] = 0.f;
  f16x8 ah[MT][2], al[MT][2], bh[2], bl[2], nah[MT][2], nal[MT][2], nbh[2], nbl[2];
  for (int m = 0; m < MT; ++m) for (int a = 0; a < 2; ++a) { ah[m][a] = hi[r16 + 16 * a + m * 46]; al[m][a] = lo[r16 + 16 * a + m * 46]; }
  const f16x8* wl = w + kg * 16 + r16;
  for (int b = 0; b < 2; ++b) { bh[b] = wl[b * 64]; bl[b] = wl[64 * nsteps_w + b * 64]; }
  int tw = 0;
  for (int t = 0; t < iters / 2; ++t) {
    if (GLB) { tw = tw + 2 >= nsteps_w ? 0 : tw + 2; for (int b = 0; b < 2; ++b) { nbh[b] = wl[(tw + b) * 64]; nbl[b] = wl[64 * nsteps_w + (tw + b) * 64]; } }
    else { for (int b = 0; b < 2; ++b) { nbh[b] = bh[b]; nbl[b] = bl[b]; } }
    const int off = wave * MT * 46 + ((t * 5) & 511) + kg * 3;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (LDSR) { nah[m][a] = hi[r16 + 16 * a + off + m * 46]; nal[m][a] = lo[r16 + 16 * a + off + m * 46]; } else { nah[m][a] = al[m][a]; nal[m][a] = ah[m][a]; }
      }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[m][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][a], bh[b], acc[m][a][b], 0, 0, 0);
          acc[m][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][a], bl[b], acc[m][a][b], 0, 0, 0);
          acc[m][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m][a], bh[b], acc[m][a][b], 0, 0, 0);
        }
#pragma unroll
    for (int m = 0; m < MT; ++m) for (int a = 0; a < 2; ++a) { ah[m][a] = nah[m][a]; al[m][a] = nal[m][a]; }
    for (int b = 0; b < 2; ++b) { bh[b] = nbh[b]; bl[b] = nbl[b]; }
  }
  float s = 0.f;
  for (int m = 0; m < MT; ++m) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) s += acc[m][a][b][i];
  out[blockIdx.x * 256 + tid] = s;
}

template <int LDSR, int GLB, int MT>
void run16(const char* name, const f16x8* w, float* out, int nsteps_w) {
  const int iters = 4000, blocks = 512;
  hipFuncSetAttribute((const void*)probe16<LDSR, GLB, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    probe16<LDSR, GLB, MT><<<blocks, 256, 65536>>>(w, out, iters, nsteps_w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * (3.0 * MT) * 32 * 32 * 16 * 2;
    if (rep) printf("%-28s MT=%d: %.3f ms, %.0f TFLOP/s f16 (%.0f fp32-equivalent)  [16x16x32]\n", name, MT, ms, flop / ms / 1e9, flop / ms / 3e9);
  }
}

template <int LDSR, int GLB, int MT>
void run(const char* name, const f16x8* w, float* out, int nsteps_w) {
  const int iters = 4000, blocks = 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    probe<LDSR, GLB, MT><<<blocks, 256, 65536>>>(w, out, iters, nsteps_w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * (3.0 * MT) * 32 * 32 * 16 * 2;
    if (rep) printf("%-28s MT=%d: %.3f ms, %.0f TFLOP/s f16 (%.0f fp32-equivalent)\n", name, MT, ms, flop / ms / 1e9, flop / ms / 3e9);
  }
}

int main() {
  const int nsteps_w = 113;                    // k15: 113 steps per 8-channel chunk
  f16x8* w; float* out;
  hipMalloc(&w, sizeof(f16x8) * 64 * nsteps_w * 2 + 4096); {
    const size_t nh = (sizeof(f16x8) * 64 * nsteps_w * 2 + 4096) / 2;
    _Float16* h = (_Float16*)malloc(nh * 2);
    unsigned r = 12345u;
    for (size_t i = 0; i < nh; ++i) { r = r * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(r >> 16) % 2001 - 1000) * 0.001f * (getenv("ZERO_W") ? 0.f : 1.f)); }
    hipMemcpy(w, h, nh * 2, hipMemcpyHostToDevice); free(h);
  }
  hipMalloc(&out, 512 * 256 * 4);
  hipFuncSetAttribute((const void*)probe<0, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)probe<1, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)probe<0, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)probe<1, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)probe<1, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)probe<0, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run<0, 0, 4>("mfma only", w, out, nsteps_w);
  run<1, 0, 4>("mfma + lds A", w, out, nsteps_w);
  run<0, 1, 4>("mfma + global B", w, out, nsteps_w);
  run<1, 1, 4>("mfma + lds A + global B", w, out, nsteps_w);
  run16<0, 0, 4>("mfma only", w, out, nsteps_w);
  run16<1, 1, 4>("mfma + lds A + global B", w, out, nsteps_w);
  run<0, 0, 8>("mfma only", w, out, nsteps_w);
  run<1, 1, 8>("mfma + lds A + global B", w, out, nsteps_w);
  return 0;
}
