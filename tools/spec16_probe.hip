// Research spike for the next round (NOT part of the product path): the forward tile transform of csrc/spectral_conv.hip with the channel
// index in a 16-wide N dimension (v_mfma_f32_16x16x4_f32) instead of 32 lanes.  The intermediate U[y][s][c] of a 32 x 32 tile is then 64 KB,
// so TWO workgroups fit a CU and one workgroup's barrier / transition gaps are filled by the other's MFMAs - the structural limit
// DESIGN.md section 4.1 measures (matrix pipes busy 48 % with one 128 KB workgroup per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/spec16_probe.hip -o tools/spec16_probe && tools/spec16_probe [waves_per_wg=4] [wgs_per_cu=2] [pipeline=1] [wide_loads=1]
// Prints the time per 8 192 tiles x 32 channels (the library's spec_fwd_kernel: 0.44 ms) and checks one tile against a host DFT.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int T = 32, ROWS = 1024;
constexpr int TAB_G = 0, TAB_F2 = 1024, TAB_FLOATS = 3072;

struct P { const float* x; float* sp; const float* tab; int H, W, ld, tiles_x, tiles_y, V, ntile, cgroups; };

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
// 4 x 4 transpose inside every quad of lanes: register k of lane 4j + i holds M[k][i] before, M[i][k] after
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3, int lane) {
  const bool i1 = lane & 2, i0 = lane & 1;
  { const float y = dpp_quad<0x4E>(i1 ? v0 : v2); v0 = i1 ? y : v0; v2 = i1 ? v2 : y; }
  { const float y = dpp_quad<0x4E>(i1 ? v1 : v3); v1 = i1 ? y : v1; v3 = i1 ? v3 : y; }
  { const float y = dpp_quad<0xB1>(i0 ? v0 : v1); v0 = i0 ? y : v0; v1 = i0 ? v1 : y; }
  { const float y = dpp_quad<0xB1>(i0 ? v2 : v3); v2 = i0 ? y : v2; v3 = i0 ? v3 : y; }
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NW waves per workgroup; lane = (q = lane / 16: K sub-index, n = lane % 16: A row / B column = channel)
template <int NW, bool PIPE, bool WIDE>
__global__ __launch_bounds__(64 * NW, 2) void fwd16_kernel(P p) {
  extern __shared__ __attribute__((aligned(16))) float U[];                  // U[(y*32 + s)*16 + c]
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int par = wave & 1, qq = wave >> 1;                                     // y axis: this wave's output parity and column set
  constexpr int NQ = NW / 2;
  float greg[2][8], freg[2][8];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      greg[mb][ks] = p.tab[TAB_G + (4 * ks + kq) * 32 + 16 * mb + n];
      freg[mb][ks] = p.tab[TAB_F2 + par * 1024 + (4 * ks + kq) * 32 + 16 * mb + n];
    }
  const int total = p.ntile * p.cgroups;
  constexpr int RPW = 32 / NW;                                                 // window rows per wave
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    const int g = item % p.cgroups;
    int t = item / p.cgroups;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int nimg = t / p.tiles_y;
    const float* img = p.x + (int64_t)nimg * p.H * p.W * p.ld + 16 * g + n;
    const int wy0 = ty * p.V, wx0 = tx * p.V;
    // WIDE: one dwordx4 per lane fetches four channels of ONE pixel; an in-quad 4 x 4 transpose turns four such loads (the quad's lanes
    // take four different pixels = K steps) into the MFMA operand layout (lane = channel, register = K step): 2 loads instead of 8 per row
    unsigned off[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) off[ks] = (unsigned)(min(wx0 + 4 * ks + kq, p.W - 1) * p.ld);
    unsigned offw[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) offw[m] = (unsigned)(min(wx0 + 4 * (4 * m + (n & 3)) + kq, p.W - 1) * p.ld);
    const float* imgw = p.x + (int64_t)nimg * p.H * p.W * p.ld + 16 * g + (n & ~3);
    // ---- x axis: two 16-row blocks of the real -> half-complex matrix, K = 32 pixels in 8 steps of 4
    float v[2][8];
    auto load_row = [&](int y, float (&d)[8]) {
      if (WIDE) {
        const float* row = imgw + (int64_t)min(wy0 + y, p.H - 1) * p.W * p.ld;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(row + offw[m]);
          d[4 * m] = q[0]; d[4 * m + 1] = q[1]; d[4 * m + 2] = q[2]; d[4 * m + 3] = q[3];
        }
      } else {
        const float* row = img + (int64_t)min(wy0 + y, p.H - 1) * p.W * p.ld;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) d[ks] = row[off[ks]];
      }
    };
    auto fix_row = [&](float (&d)[8]) {                                        // after the loads have landed: the two quad transposes
      if (WIDE) {
        quad_transpose(d[0], d[1], d[2], d[3], lane);
        quad_transpose(d[4], d[5], d[6], d[7], lane);
      }
    };
    load_row(wave, v[0]);
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int y = wave + NW * j;
      if (j + 1 < RPW) load_row(y + NW, v[(j + 1) & 1]);
      fix_row(v[j & 1]);
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        a0 = mfma16(greg[0][ks], v[j & 1][ks], a0);
        a1 = mfma16(greg[1][ks], v[j & 1][ks], a1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        U[(y * 32 + 4 * kq + r) * 16 + n] = a0[r];
        U[(y * 32 + 16 + 4 * kq + r) * 16 + n] = a1[r];
      }
    }
    lds_barrier();
    // ---- y axis: wave (qq, par) takes complex columns fx = 1 + qq, 1 + qq + NQ, ... for its parity; the last column set also takes one
    // real column (0 for par = 0, 16 for par = 1)
    float* out = p.sp + (int64_t)item * ROWS * 16;
    auto load_col = [&](int fx, float (&b)[8]) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int k = 4 * ks + kq, y = k & 15, sc = k < 16 ? fx : 16 + fx;
        const float lo = U[(y * 32 + sc) * 16 + n], hi = U[((y + 16) * 32 + sc) * 16 + n];
        b[ks] = par ? lo - hi : lo + hi;
      }
    };
    // two accumulator sets: the chain of column c + 1 is queued BEFORE the stores of column c are issued, so the matrix pipe keeps running
    // while this wave waits for column c's results and writes them (PIPE); without PIPE: read, multiply, store, one column at a time
    auto chain = [&](const float (&b)[8], f32x4& a0, f32x4& a1) {
      a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        a0 = mfma16(freg[0][ks], b[ks], a0);
        a1 = mfma16(freg[1][ks], b[ks], a1);
      }
    };
    auto store_col = [&](int fx, const f32x4& a0, const f32x4& a1) {
#ifdef ST16
      // 16-byte stores: an in-quad transpose gives lane (kq, n = 4 j + i) accumulator row 4 kq + i with channels 4 j .. 4 j + 3 - two stores per
      // column and lane instead of eight.  Together with the wide loads a wave then has ~34 vector-memory operations per item in flight:
      // below the 63 that the 6-bit vmcnt counter of gfx9 can tell apart
      float t0[4] = {a0[0], a0[1], a0[2], a0[3]}, t1[4] = {a1[0], a1[1], a1[2], a1[3]};
      quad_transpose(t0[0], t0[1], t0[2], t0[3], lane);
      quad_transpose(t1[0], t1[1], t1[2], t1[3], lane);
      float* o4 = out + (int64_t)(64 + 64 * (fx - 1) + par) * 16 + (n & ~3);
      *reinterpret_cast<f32x4*>(o4 + (2 * (4 * kq + (n & 3))) * 16) = (f32x4){t0[0], t0[1], t0[2], t0[3]};
      *reinterpret_cast<f32x4*>(o4 + (32 + 2 * (4 * kq + (n & 3))) * 16) = (f32x4){t1[0], t1[1], t1[2], t1[3]};
      return;
#endif
      float* o = out + (int64_t)(64 + 64 * (fx - 1) + par) * 16 + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                            // accumulator row 16 part + m -> spectrum row 32 part + 2 m + par
        o[(2 * (4 * kq + r)) * 16] = a0[r];
        o[(32 + 2 * (4 * kq + r)) * 16] = a1[r];
      }
    };
    float b[8];
    f32x4 p0, p1, c0, c1;
    load_col(1 + qq, b);
    chain(b, p0, p1);
#pragma unroll 1
    for (int fx = 1 + qq; fx <= 15; fx += NQ) {
      const bool more = fx + NQ <= 15;
      if (PIPE) {
        if (more) { load_col(fx + NQ, b); chain(b, c0, c1); }
        store_col(fx, p0, p1);
        p0 = c0; p1 = c1;
      } else {
        store_col(fx, p0, p1);
        if (more) { load_col(fx + NQ, b); chain(b, p0, p1); }
      }
    }
    if (qq == NQ - 1) {
      const int col = par ? 16 : 0;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) b[ks] = U[((4 * ks + kq) * 32 + col) * 16 + n];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        a0 = mfma16(greg[0][ks], b[ks], a0);
        a1 = mfma16(greg[1][ks], b[ks], a1);
      }
      float* o = out + (int64_t)(col ? 32 : 0) * 16 + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[(4 * kq + r) * 16] = a0[r];
        o[(16 + 4 * kq + r) * 16] = a1[r];
      }
    }
    lds_barrier();
  }
}

static void build_tables(std::vector<float>& tab) {
  tab.assign(TAB_FLOATS, 0.f);
  const double tp = 2.0 * M_PI / T;
  auto G = [&](int s, int x) { return s <= 16 ? cos(tp * s * x) : -sin(tp * (s - 16) * x); };
  auto F2 = [&](int par, int row, int k) {
    const int po = row >> 4, m = row & 15, pi = k >> 4, y = k & 15;
    const double th = tp * (((2 * m + par) * y) & 31);
    return po == pi ? cos(th) : (po == 0 ? sin(th) : -sin(th));
  };
  for (int k = 0; k < 32; ++k)
    for (int m = 0; m < 32; ++m) {
      tab[TAB_G + k * 32 + m] = (float)G(m, k);
      for (int par = 0; par < 2; ++par) tab[TAB_F2 + par * 1024 + k * 32 + m] = (float)F2(par, m, k);
    }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int NW = argc > 1 ? atoi(argv[1]) : 4, per_cu = argc > 2 ? atoi(argv[2]) : 2;
  const int N = 8, H = 1024, W = 1024, C = 32, V = 18;
  const int tiles = (H + V - 1) / V;
  const int ntile_all = N * tiles * tiles, ntile = 8192;
  std::vector<float> hx((size_t)N * H * W * C), tab;
  srand(1);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  build_tables(tab);
  float *dx, *dsp, *dtab;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dtab, tab.size() * 4));
  CK(hipMalloc(&dsp, (size_t)ntile * 2 * ROWS * 16 * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  P p{dx, dsp, dtab, H, W, C, tiles, tiles, V, ntile, 2};
  (void)ntile_all;
  const size_t lds = (size_t)T * T * 16 * 4;
  const int grid = 256 * per_cu;
  const bool pipe = argc > 3 ? atoi(argv[3]) != 0 : true, wide = argc > 4 ? atoi(argv[4]) != 0 : true;
#define VARIANT(NWv, Pv, Wv) if (NW == NWv && pipe == Pv && wide == Wv) { \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fwd16_kernel<NWv, Pv, Wv>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((fwd16_kernel<NWv, Pv, Wv>), dim3(grid), dim3(64 * NWv), lds, 0, p); }
  auto launch = [&]() {
    VARIANT(4, true, true) VARIANT(4, true, false) VARIANT(4, false, true) VARIANT(4, false, false)
    VARIANT(8, true, true) VARIANT(8, true, false) VARIANT(8, false, true) VARIANT(8, false, false)
  };
  launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("fwd16 (%d waves / workgroup, %d workgroups / CU, 64 KB LDS each, y-axis pipelining %s, wide loads %s): %.3f ms per 8192 tiles x 32 channels (library spec_fwd_kernel: 0.44 ms)\n", NW, per_cu, pipe ? "on" : "off", wide ? "on" : "off", ms / 10);
  // ---- check item 5 (tile 2, channel group 1) against a host DFT
  const int item = 5, g = item % 2, t = item / 2, tx = t % tiles, ty = (t / tiles) % tiles;
  std::vector<float> hs((size_t)ROWS * 16);
  CK(hipMemcpy(hs.data(), dsp + (size_t)item * ROWS * 16, hs.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (int c = 0; c < 16; c += 5) {
    for (int fx = 0; fx <= 16; fx += 3)
      for (int fy = 0; fy < 32; fy += 5) {
        double re = 0.0, im = 0.0;
        for (int y = 0; y < 32; ++y)
          for (int x = 0; x < 32; ++x) {
            const int sy = std::min(ty * V + y, H - 1), sx = std::min(tx * V + x, W - 1);
            const double v = hx[((size_t)sy * W + sx) * C + 16 * g + c];            // image 0
            const double th = -2.0 * M_PI * (fx * x + fy * y) / 32.0;
            re += v * cos(th); im += v * sin(th);
          }
        double gre, gim;
        if (fx == 0 || fx == 16) {                                                   // real columns: half-complex in y
          const int base = fx ? 32 : 0;
          if (fy > 16) continue;
          gre = hs[(size_t)(base + fy) * 16 + c];
          gim = (fy == 0 || fy == 16) ? 0.0 : hs[(size_t)(base + 16 + fy) * 16 + c];
        } else {
          gre = hs[(size_t)(64 + 64 * (fx - 1) + fy) * 16 + c];
          gim = hs[(size_t)(64 + 64 * (fx - 1) + 32 + fy) * 16 + c];
        }
        worst = std::max(worst, std::max(fabs(gre - re), fabs(gim - im)));
        scale = std::max(scale, std::max(fabs(re), fabs(im)));
      }
  }
  printf("check vs host DFT (tile %d, channel group %d): max abs err %.3e at scale %.3e\n", t, g, worst, scale);
  return worst < 1e-3 * scale ? 0 : 2;
}
