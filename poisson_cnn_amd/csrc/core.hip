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
  if (const char* e = getenv("PCNN_SPEC_T")) h->spectral_tile = atoi(e);
  if (const char* e = getenv("PCNN_SPEC_XFORM")) h->spectral_xform = (e[0] == 'f' || e[0] == '1') ? PCNN_XFORM_FFT : PCNN_XFORM_MFMA;
  *out = h;
  return 0;
}

extern "C" int pcnn_destroy(pcnn_handle h) {
  if (h && h->scratch) (void)hipFree(h->scratch);
  if (h && h->spec_ws) (void)hipFree(h->spec_ws);
  if (h && h->aux_ws) (void)hipFree(h->aux_ws);
  if (h) for (void* p : h->retired) (void)hipFree(p);
  if (h && h->comm) pcnn_comm_release(h);
  if (h) pcnn_filter_cache_free(h);
  delete h;
  return 0;
}

extern "C" int pcnn_set_stream(pcnn_handle h, void* hip_stream) {
  if (!h) return 1;
  h->stream = static_cast<hipStream_t>(hip_stream);
  return 0;
}

extern "C" int pcnn_set_workspace_retain(pcnn_handle h, int retain) {
  if (!h) return 1;
  h->retain = retain ? 1 : 0;
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

extern "C" int pcnn_set_spectral_tile(pcnn_handle h, int tile) {
  if (!h) return 1;
  PCNN_REQUIRE(h, tile == 0 || tile == 32 || tile == 64, "pcnn_set_spectral_tile: tile size %d (0 = per layer, 32, 64)", tile);
  h->spectral_tile = tile;
  return 0;
}

extern "C" int pcnn_get_spectral_tile(pcnn_handle h) { return h ? h->spectral_tile : -1; }

extern "C" int pcnn_set_spectral_transform(pcnn_handle h, int xform) {
  if (!h) return 1;
  PCNN_REQUIRE(h, xform == PCNN_XFORM_MFMA || xform == PCNN_XFORM_FFT, "pcnn_set_spectral_transform: unknown transform %d", xform);
  h->spectral_xform = xform;
  return 0;
}

extern "C" int pcnn_get_spectral_transform(pcnn_handle h) { return h ? h->spectral_xform : -1; }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of a HOST buffer: the checksum of TensorFlow's TensorBundle checkpoint files
// (tensorflow/core/lib/hash/crc32c.h) that poisson_cnn_amd/tf_checkpoint.py reads and writes.  Host-only helper, no device work.
extern "C" uint32_t pcnn_crc32c(const void* data, size_t n, uint32_t crc) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  const unsigned char* p = static_cast<const unsigned char*>(data);
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return ~crc;
}
