// Handle management for libpcnn.
#include "pcnn_internal.h"
#include <stdlib.h>

extern "C" int pcnn_version(void) { return 100; }

extern "C" int pcnn_create(int device, void* hip_stream, pcnn_handle* out) {
  if (!out) return 1;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return 2;
  if (hipSetDevice(device) != hipSuccess) return 3;
  pcnn_handle h = new pcnn_handle_s();
  h->device = device;
  h->stream = static_cast<hipStream_t>(hip_stream);
  if (const char* e = getenv("PCNN_SPECTRAL")) h->spectral_mode = atoi(e);
  *out = h;
  return 0;
}

extern "C" int pcnn_destroy(pcnn_handle h) {
  if (h && h->scratch) (void)hipFree(h->scratch);
  if (h && h->spec_ws) (void)hipFree(h->spec_ws);
  delete h;
  return 0;
}

extern "C" int pcnn_set_stream(pcnn_handle h, void* hip_stream) {
  if (!h) return 1;
  h->stream = static_cast<hipStream_t>(hip_stream);
  return 0;
}

extern "C" int pcnn_sync(pcnn_handle h) {
  if (!h) return 1;
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) PCNN_FAIL(h, "pcnn_sync: %s", hipGetErrorString(e));
  return 0;
}

extern "C" const char* pcnn_last_error(pcnn_handle h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int pcnn_set_math_mode(pcnn_handle h, int mode) {
  if (!h) return 1;
  PCNN_REQUIRE(h, mode == PCNN_MATH_FP32 || mode == PCNN_MATH_SPLIT_F16, "pcnn_set_math_mode: unknown mode %d", mode);
  h->math_mode = mode;
  return 0;
}

extern "C" int pcnn_get_math_mode(pcnn_handle h) { return h ? h->math_mode : -1; }

extern "C" int pcnn_set_spectral_mode(pcnn_handle h, int mode) {
  if (!h) return 1;
  PCNN_REQUIRE(h, mode >= -1 && mode <= 1, "pcnn_set_spectral_mode: unknown mode %d", mode);
  h->spectral_mode = mode;
  return 0;
}

extern "C" int pcnn_get_spectral_mode(pcnn_handle h) { return h ? h->spectral_mode : -2; }
