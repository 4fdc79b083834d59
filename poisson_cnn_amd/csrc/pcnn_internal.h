// Internal helpers shared by the libpcnn translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/pcnn.h"

struct pcnn_handle_s {
  int device;
  hipStream_t stream;
  std::string err;
  void* scratch = nullptr;        // packed-filter scratch of the conv kernels (grown on demand, owned by the handle)
  size_t scratch_bytes = 0;
  int math_mode = 0;              // PCNN_MATH_FP32 (exact fp32 MFMA) or PCNN_MATH_SPLIT_F16 (3 x fp16 split, fp32 accumulate)
  float* y_absmax = nullptr;      // set by pcnn_conv2d_fwd_absmax for the duration of one forward launch: receives max|y|
  void* spec_ws = nullptr;        // spectral-convolution workspace (tables, filter spectrum, tile spectra; grown on demand, owned by the handle)
  size_t spec_ws_bytes = 0;
  size_t spec_ws_limit = 0;       // caller's cap on the spectral workspace in bytes (0: none), pcnn_set_workspace_limit
  void* aux_ws = nullptr;         // scratch of the two-pass resize (x-interpolated coarse rows; grown on demand, owned by the handle)
  size_t aux_ws_bytes = 0;
  int spectral_mode = -1;         // PCNN_SPECTRAL_AUTO (cost model) / _OFF / _FORCE, see pcnn_set_spectral_mode
  int spectral_tile = 0;          // 0: per layer (pick_tile), 32 / 64: that tile size wherever the layer allows it, see pcnn_set_spectral_tile
  int spectral_xform = 1;         // transform kernels of the spectral route: 1 = in-register FFT on the vector ALUs (default since round 5), 0 = DFT as a GEMM on the matrix cores (pcnn_set_spectral_transform)
  int retain = 0;                 // pcnn_set_workspace_retain: outgrown handle-owned buffers are kept (a captured hipGraph may still replay into them)
  std::vector<void*> retired;     // ... here, until pcnn_destroy
  unsigned long long filter_version = 0;   // pcnn_set_filter_version: 0 = filter spectra are recomputed by every call; else the caller's weights version
  void* filter_cache = nullptr;   // ... and the spectra kept per (filter pointer, shape, tile size), spectral_conv.hip
  long long fc_hits = 0, fc_fills = 0, fc_refreshes = 0;   // cumulative over the handle's life (pcnn_filter_cache_clear empties the cache, not these)
  void* comm = nullptr;           // RCCL communicator (ncclComm_t) of pcnn_comm_init, see collective.hip
  int comm_rank = 0, comm_size = 0;
};

void pcnn_comm_release(pcnn_handle_s* h);   // collective.hip
void pcnn_filter_cache_free(pcnn_handle_s* h);   // spectral_conv.hip

// A handle-owned buffer is being outgrown (or capped): free it once the stream has drained - unless the caller declared that recorded work
// (a hipGraph captured on this handle's stream) may still use it; then it is parked until pcnn_destroy.
static inline void pcnn_release(pcnn_handle_s* h, void* p) {
  if (!p) return;
  if (h->retain) { h->retired.push_back(p); return; }
  (void)hipStreamSynchronize(h->stream);
  (void)hipFree(p);
}

#define PCNN_FAIL(h, ...)                                   \
  do {                                                      \
    char _b[512];                                           \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                  \
    if (h) (h)->err = _b;                                   \
    return 1;                                               \
  } while (0)

#define PCNN_REQUIRE(h, cond, ...) \
  do {                             \
    if (!(cond)) PCNN_FAIL(h, __VA_ARGS__); \
  } while (0)

#define PCNN_CHECK_LAUNCH(h, name)                                               \
  do {                                                                           \
    hipError_t _e = hipGetLastError();                                           \
    if (_e != hipSuccess) PCNN_FAIL(h, "%s: %s", name, hipGetErrorString(_e));   \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pcnn_act(float v, int act, float alpha) {
  switch (act) {
    case PCNN_ACT_LEAKY_RELU: return v > 0.f ? v : alpha * v;
    case PCNN_ACT_TANH: return tanhf(v);
    case PCNN_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}
// derivative of the activation expressed through its OUTPUT a = act(z)
__device__ __forceinline__ float pcnn_act_grad_from_out(float a, int act, float alpha) {
  switch (act) {
    case PCNN_ACT_LEAKY_RELU: return a > 0.f ? 1.f : alpha;
    case PCNN_ACT_TANH: return 1.f - a * a;
    case PCNN_ACT_RELU: return a > 0.f ? 1.f : 0.f;
    default: return 1.f;
  }
}

// tf.pad index map.  Returns the source index in [0,n) or -1 for "use the constant".
__device__ __forceinline__ int pcnn_pad_index(int i, int n, int mode) {
  if (i >= 0 && i < n) return i;
  if (mode == PCNN_PAD_CONSTANT) return -1;
  int r;
  if (mode == PCNN_PAD_SYMMETRIC) r = i < 0 ? -i - 1 : 2 * n - 1 - i;
  else r = i < 0 ? -i : 2 * n - 2 - i;
  return r < 0 ? 0 : (r >= n ? n - 1 : r);   // clamp: only reached by tile overhang that is never stored
}

static inline int pcnn_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t pcnn_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
