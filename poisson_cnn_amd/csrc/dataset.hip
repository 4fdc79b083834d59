// On-device reference-solution generators (poisson_CNN/dataset):
//   * Dirichlet 5-point FD Poisson solve by DST-I diagonalisation, carried out in fp64 on the f64 matrix cores
//     (v_mfma_f64_16x16x4_f64) as four dense products with the orthonormal sine matrix - this replaces pyamg's
//     Ruge-Stuben V-cycles / AMGX (dataset/solvers/multigrid.py:98-150) and poisson_RHS (dataset/solvers/cholesky.py:45-119)
//     with a direct solve of the SAME linear system (fp64 solve, fp32 I/O like the reference).
//   * separable series synthesis sum_{A,B} c[A,B] f(A x) g(B y) (dataset/utils/generate_smooth_function.py:45-62)
//   * rank-R separable sums (the Taylor component of dataset/generators/reverse.py:231-256)
//   * per-sample max-magnitude normalisation (dataset/utils/set_max_magnitude.py)
#include <math.h>
#include "pcnn_internal.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {

static dim3 grid1d(int64_t total, int block = 256, int maxb = 16384) {
  int64_t b = pcnn_cdiv64(total, block);
  if (b > maxb) b = maxb;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

// ---------------------------------------------------------------- fp64 batched GEMM, C = A * B (row-major)
constexpr int GT = 64, GK = 16;

__global__ __launch_bounds__(256) void gemm_f64_kernel(int M, int Nn, int K, const double* __restrict__ A, int64_t sA, int lda,
                                                       const double* __restrict__ B, int64_t sB, int ldb, double* __restrict__ C, int64_t sC, int ldc) {
  __shared__ double As[GT][GK + 1];
  __shared__ double Bs[GK][GT + 1];
  const int b = blockIdx.z, m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  A += (int64_t)b * sA; B += (int64_t)b * sB; C += (int64_t)b * sC;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wr = wave >> 1, wc = wave & 1, l16 = lane & 15, lq = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += GK) {
    __syncthreads();
    for (int e = tid; e < GT * GK; e += 256) {
      const int r = e / GK, c = e % GK;
      As[r][c] = (m0 + r < M && k0 + c < K) ? A[(int64_t)(m0 + r) * lda + k0 + c] : 0.0;
      const int r2 = e / GT, c2 = e % GT;
      Bs[r2][c2] = (k0 + r2 < K && n0 + c2 < Nn) ? B[(int64_t)(k0 + r2) * ldb + n0 + c2] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      double a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[wr * 32 + i * 16 + l16][kk * 4 + lq];
#pragma unroll
      for (int j = 0; j < 2; ++j) bb[j] = Bs[kk * 4 + lq][wc * 32 + j * 16 + l16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + i * 16 + lq + 4 * r, colc = n0 + wc * 32 + j * 16 + l16;   // f64 C/D map: row = (lane>>4) + 4*reg
        if (row < M && colc < Nn) C[(int64_t)row * ldc + colc] = acc[i][j][r];
      }
}

// ---------------------------------------------------------------- DST Poisson solve pieces
// B[n,i,j] = -dx^2 f[i+1,j+1] + boundary values folded onto the first interior ring (dataset/solvers/cholesky.py:85,114-117)
__global__ void dst_build_rhs_kernel(int N, int H, int W, const float* __restrict__ rhs, const float* __restrict__ left, const float* __restrict__ right,
                                     const float* __restrict__ bottom, const float* __restrict__ top, const float* __restrict__ dx, double* __restrict__ Bm) {
  const int nh = H - 2, nw = W - 2;
  const int64_t total = (int64_t)N * nh * nw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % nw; const int i = (idx / nw) % nh; const int n = idx / ((int64_t)nh * nw);
    const double h = (double)dx[n];
    double v = -h * h * (double)rhs[((int64_t)n * H + i + 1) * W + j + 1];
    if (j == 0) v += (double)bottom[(int64_t)n * H + i + 1];       // F[..., 1:-1, 1]  += bottom
    if (j == nw - 1) v += (double)top[(int64_t)n * H + i + 1];     // F[..., 1:-1, -2] += top
    if (i == 0) v += (double)left[(int64_t)n * W + j + 1];         // F[..., 1, 1:-1]  += left
    if (i == nh - 1) v += (double)right[(int64_t)n * W + j + 1];   // F[..., -2, 1:-1] += right
    Bm[idx] = v;
  }
}

__global__ void dst_divide_kernel(int N, int nh, int nw, const double* __restrict__ lam_h, const double* __restrict__ lam_w, double* __restrict__ T) {
  const int64_t total = (int64_t)N * nh * nw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % nw; const int i = (idx / nw) % nh;
    T[idx] /= (lam_h[i] + lam_w[j]);
  }
}

// interior from U, then [:, -1]=top, [:, 0]=bottom, [0, :]=left, [-1, :]=right in that order (dataset/solvers/multigrid.py:145-148)
__global__ void dst_write_soln_kernel(int N, int H, int W, const double* __restrict__ U, const float* __restrict__ left, const float* __restrict__ right,
                                      const float* __restrict__ bottom, const float* __restrict__ top, float* __restrict__ soln) {
  const int nh = H - 2, nw = W - 2;
  const int64_t total = (int64_t)N * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % W; const int i = (idx / W) % H; const int n = idx / ((int64_t)H * W);
    float v;
    if (i == 0) v = left[(int64_t)n * W + j];
    else if (i == H - 1) v = right[(int64_t)n * W + j];
    else if (j == 0) v = bottom[(int64_t)n * H + i];
    else if (j == W - 1) v = top[(int64_t)n * H + i];
    else v = (float)U[((int64_t)n * nh + i - 1) * nw + j - 1];
    soln[idx] = v;
  }
}

// ---------------------------------------------------------------- mixed Dirichlet / Neumann 5-point solve
// Vertex-centred grid as above; an edge is Dirichlet (its nodes are given) or Neumann (du/dn = g along the outward normal; its nodes are
// unknowns, closed by the second-order ghost node u_ghost = u_inner + 2 dx g).  The unknowns are rows i0..i1 x columns j0..j1:
//   4 u_ij - [neighbours] = -dx^2 f_ij + (Dirichlet neighbours' values) + 2 dx g on Neumann edges, the mirrored neighbour counted twice.
// types bit 0/1/2/3 = left / right / bottom / top is Neumann.
__global__ void mixed_build_rhs_kernel(int N, int H, int W, int types, const float* __restrict__ rhs, const float* __restrict__ left,
                                       const float* __restrict__ right, const float* __restrict__ bottom, const float* __restrict__ top,
                                       const float* __restrict__ dx, double* __restrict__ Bm) {
  const int i0 = (types & 1) ? 0 : 1, i1 = (types & 2) ? H - 1 : H - 2, j0 = (types & 4) ? 0 : 1, j1 = (types & 8) ? W - 1 : W - 2;
  const int mh = i1 - i0 + 1, mw = j1 - j0 + 1;
  const int64_t total = (int64_t)N * mh * mw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int jj = idx % mw; const int ii = (idx / mw) % mh; const int n = idx / ((int64_t)mh * mw);
    const int i = i0 + ii, j = j0 + jj;
    const double h = (double)dx[n];
    double v = -h * h * (double)rhs[((int64_t)n * H + i) * W + j];
    // axis -2 (index i): left edge i = 0, right edge i = H-1; their values are indexed by j
    if (i == 0) v += 2.0 * h * (double)left[(int64_t)n * W + j];                       // Neumann node on the left edge
    else if (i == 1 && !(types & 1)) v += (double)left[(int64_t)n * W + j];            // Dirichlet neighbour
    if (i == H - 1) v += 2.0 * h * (double)right[(int64_t)n * W + j];
    else if (i == H - 2 && !(types & 2)) v += (double)right[(int64_t)n * W + j];
    // axis -1 (index j): bottom edge j = 0, top edge j = W-1; values indexed by i.  A Dirichlet corner takes the left / right value
    // (the write order of dataset/solvers/multigrid.py:145-148 lets left / right overwrite bottom / top there)
    if (j == 0) v += 2.0 * h * (double)bottom[(int64_t)n * H + i];
    else if (j == 1 && !(types & 4)) v += (double)bottom[(int64_t)n * H + i];
    if (j == W - 1) v += 2.0 * h * (double)top[(int64_t)n * H + i];
    else if (j == W - 2 && !(types & 8)) v += (double)top[(int64_t)n * H + i];
    Bm[idx] = v;
  }
}

// coefficient / (lam_h + lam_w); the null mode of the all-Neumann problem (lam = 0) is dropped, which is the solution whose
// trapezoidal integral vanishes - the zero-integral constraint Navier_Stokes_2D/solvers.py:258-259 imposes with a Lagrange multiplier
__global__ void mixed_divide_kernel(int N, int mh, int mw, const double* __restrict__ lam_h, const double* __restrict__ lam_w, double* __restrict__ T) {
  const int64_t total = (int64_t)N * mh * mw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % mw; const int i = (idx / mw) % mh;
    const double l = lam_h[i] + lam_w[j];
    T[idx] = fabs(l) > 1e-13 ? T[idx] / l : 0.0;
  }
}

__global__ void mixed_write_soln_kernel(int N, int H, int W, int types, const double* __restrict__ U, const float* __restrict__ left,
                                        const float* __restrict__ right, const float* __restrict__ bottom, const float* __restrict__ top,
                                        float* __restrict__ soln) {
  const int i0 = (types & 1) ? 0 : 1, i1 = (types & 2) ? H - 1 : H - 2, j0 = (types & 4) ? 0 : 1, j1 = (types & 8) ? W - 1 : W - 2;
  const int mh = i1 - i0 + 1, mw = j1 - j0 + 1;
  const int64_t total = (int64_t)N * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % W; const int i = (idx / W) % H; const int n = idx / ((int64_t)H * W);
    float v;
    if (i < i0) v = left[(int64_t)n * W + j];
    else if (i > i1) v = right[(int64_t)n * W + j];
    else if (j < j0) v = bottom[(int64_t)n * H + i];
    else if (j > j1) v = top[(int64_t)n * H + i];
    else v = (float)U[((int64_t)n * mh + (i - i0)) * mw + (j - j0)];
    soln[idx] = v;
  }
}

// ---------------------------------------------------------------- series synthesis
// out[n,a,b] (+)= sum_{A<ka,B<kb} c[n,A,B] f((A+1) x_a) g((B+1) y_b), x = linspace(0,pi,H), y = linspace(0,pi,W)
constexpr int SYN_ROWS = 16, SYN_MAXK = 16;
__global__ __launch_bounds__(256) void series_kernel(int H, int W, int ka, int kb, const float* __restrict__ coef, int trig, int accumulate,
                                                     float* __restrict__ out) {
  extern __shared__ float sm[];
  float* T = sm;                 // [ka][W]   : sum_B c[A,B] g((B+1) y_b)
  float* F = sm + ka * W;        // [ka][SYN_ROWS]
  const int n = blockIdx.y, a0 = blockIdx.x * SYN_ROWS;
  const float pi = 3.14159265358979323846f;
  const float* c = coef + (int64_t)n * ka * kb;
  const float dy = W > 1 ? pi / (float)(W - 1) : 0.f, dxs = H > 1 ? pi / (float)(H - 1) : 0.f;
  for (int b = threadIdx.x; b < W; b += blockDim.x) {
    float g[SYN_MAXK];
    const float yb = (float)b * dy;
    for (int B = 0; B < kb; ++B) g[B] = trig ? cosf((float)(B + 1) * yb) : sinf((float)(B + 1) * yb);
    for (int A = 0; A < ka; ++A) {
      float s = 0.f;
      for (int B = 0; B < kb; ++B) s = fmaf(c[A * kb + B], g[B], s);
      T[A * W + b] = s;
    }
  }
  for (int e = threadIdx.x; e < ka * SYN_ROWS; e += blockDim.x) {
    const int A = e / SYN_ROWS, r = e % SYN_ROWS;
    const float xa = (float)(a0 + r) * dxs;
    F[e] = trig ? cosf((float)(A + 1) * xa) : sinf((float)(A + 1) * xa);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < SYN_ROWS * W; e += blockDim.x) {
    const int r = e / W, b = e % W;
    if (a0 + r >= H) break;
    float s = 0.f;
    for (int A = 0; A < ka; ++A) s = fmaf(F[A * SYN_ROWS + r], T[A * W + b], s);
    float* dst = &out[((int64_t)n * H + a0 + r) * W + b];
    *dst = accumulate ? *dst + s : s;
  }
}

// out[n,a,b] (+)= sum_r U[n,r,a] V[n,r,b]
__global__ void separable_sum_kernel(int N, int H, int W, int R, const float* __restrict__ U, const float* __restrict__ V, int accumulate,
                                     float* __restrict__ out) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int b = idx % W; const int a = (idx / W) % H; const int n = idx / ((int64_t)H * W);
    float s = 0.f;
    for (int r = 0; r < R; ++r) s = fmaf(U[((int64_t)n * R + r) * H + a], V[((int64_t)n * R + r) * W + b], s);
    out[idx] = accumulate ? out[idx] + s : s;
  }
}

// one workgroup per sample: x *= target / max|x|
__global__ __launch_bounds__(1024) void set_max_magnitude_kernel(int64_t per, const float* __restrict__ target, float* __restrict__ x,
                                                                 float* __restrict__ factors) {
  __shared__ float red[1024];
  const int n = blockIdx.x;
  float* xs = x + (int64_t)n * per;
  float mx = 0.f;
  for (int64_t q = threadIdx.x; q < per; q += blockDim.x) mx = fmaxf(mx, fabsf(xs[q]));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const float f = target[n] / red[0];
  for (int64_t q = threadIdx.x; q < per; q += blockDim.x) xs[q] *= f;
  if (threadIdx.x == 0 && factors) factors[n] = f;
}

__global__ void scale_samples_kernel(int N, int64_t per, const float* __restrict__ s, float* __restrict__ x) {
  const int64_t total = (int64_t)N * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) x[i] *= s[i / per];
}

}  // namespace

extern "C" int pcnn_batched_gemm_f64(pcnn_handle h, int batch, int M, int Nn, int K, const double* A, int64_t strideA, int lda, const double* B,
                                     int64_t strideB, int ldb, double* C, int64_t strideC, int ldc) {
  PCNN_REQUIRE(h, h && A && B && C && batch >= 1 && M >= 1 && Nn >= 1 && K >= 1, "pcnn_batched_gemm_f64: bad argument");
  PCNN_REQUIRE(h, batch <= 65535, "pcnn_batched_gemm_f64: batch %d too large", batch);
  hipLaunchKernelGGL(gemm_f64_kernel, dim3(pcnn_cdiv(Nn, GT), pcnn_cdiv(M, GT), batch), dim3(256), 0, h->stream, M, Nn, K, A, strideA, lda, B, strideB, ldb,
                     C, strideC, ldc);
  PCNN_CHECK_LAUNCH(h, "pcnn_batched_gemm_f64");
  return 0;
}

// host helper: orthonormal DST-I matrix S[j,k] = sqrt(2/(m+1)) sin(pi (j+1)(k+1)/(m+1)) and eigenvalues 2-2cos(pi (j+1)/(m+1)),
// m = n-2 interior points of an n-point axis.
extern "C" int pcnn_dst_setup(int n, double* S, double* lam) {
  if (n < 3 || !S || !lam) return 1;
  const int m = n - 2;
  const double c = sqrt(2.0 / (m + 1.0)), pi = 3.14159265358979323846264338327950288;
  for (int j = 0; j < m; ++j) {
    lam[j] = 2.0 - 2.0 * cos(pi * (j + 1.0) / (m + 1.0));
    for (int k = 0; k < m; ++k) S[(size_t)j * m + k] = c * sin(pi * (double)(((int64_t)(j + 1) * (k + 1)) % (2 * (m + 1))) / (m + 1.0));
  }
  return 0;
}

extern "C" int pcnn_fd_poisson_dst(pcnn_handle h, int N, int H, int W, const float* rhs, const float* left, const float* right, const float* bottom,
                                   const float* top, const float* dx, const double* S_h, const double* lam_h, const double* S_w, const double* lam_w,
                                   double* tmp, float* soln) {
  PCNN_REQUIRE(h, h && rhs && left && right && bottom && top && dx && S_h && lam_h && S_w && lam_w && tmp && soln, "pcnn_fd_poisson_dst: null argument");
  PCNN_REQUIRE(h, H >= 3 && W >= 3 && N >= 1, "pcnn_fd_poisson_dst: grid %dx%d too small", H, W);
  const int nh = H - 2, nw = W - 2;
  const int64_t per = (int64_t)nh * nw;
  double* Bm = tmp; double* T = tmp + (int64_t)N * per;
  hipLaunchKernelGGL(dst_build_rhs_kernel, grid1d(N * per), dim3(256), 0, h->stream, N, H, W, rhs, left, right, bottom, top, dx, Bm);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_dst(build)");
  int rc;
  if ((rc = pcnn_batched_gemm_f64(h, N, nh, nw, nh, S_h, 0, nh, Bm, per, nw, T, per, nw))) return rc;     // T  = S_h B
  if ((rc = pcnn_batched_gemm_f64(h, N, nh, nw, nw, T, per, nw, S_w, 0, nw, Bm, per, nw))) return rc;     // Bm = T S_w   (spectral coefficients)
  hipLaunchKernelGGL(dst_divide_kernel, grid1d(N * per), dim3(256), 0, h->stream, N, nh, nw, lam_h, lam_w, Bm);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_dst(divide)");
  if ((rc = pcnn_batched_gemm_f64(h, N, nh, nw, nh, S_h, 0, nh, Bm, per, nw, T, per, nw))) return rc;     // T  = S_h (.)
  if ((rc = pcnn_batched_gemm_f64(h, N, nh, nw, nw, T, per, nw, S_w, 0, nw, Bm, per, nw))) return rc;     // Bm = T S_w = u
  hipLaunchKernelGGL(dst_write_soln_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, Bm, left, right, bottom, top, soln);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_dst(write)");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------ rocFFT route
// The same Dirichlet solve with the two DST-I passes done by rocFFT instead of the 8 n^3 GEMMs (BASELINE.json north star: "HIP stencil +
// rocFFT kernel"): the odd extension of an n-vector to length L = 2 (n + 1) turns the DST-I into a real FFT, Y[k] = -2i (D x)[k-1] with the
// unnormalised D[k][j] = sin(pi (j+1)(k+1)/(n+1)); in two dimensions FFT2(odd-odd extension)[a+1][b+1] = -4 (D_h B D_w)[a][b], real.  So
//   u = c_h c_w D_h ((D_h B D_w) ./ (lam_h + lam_w)) D_w,  c = 2 / (n + 1),
// is: build B -> odd-extend -> rocFFT real 2-D forward -> (take -Re/4, divide, odd-extend again: one kernel) -> the same FFT -> -Re/4 c_h c_w.
// O(n^2 log n) instead of O(n^3) - but the extension quadruples the data and 2 (n + 1) is an awkward length for the configs' grids (2046 =
// 2 3 11 31), so the fp64 matrix cores win up to ~1536^2 (tools/bench_fd_solver.py: 0.25 vs 0.35 ms per 1024^2 sample, 1.86 vs 1.66 ms at 2048^2):
// the Python side takes this route from 2048 points per axis; below that the GEMM route is the default and the CHECKER of this one in the tests.
// rocFFT is bound at run time (dlopen of librocfft.so on the first call), like RCCL in collective.hip: libpcnn.so keeps no link-time dependency.
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <tuple>
namespace {

typedef int (*RfSetupFn)();
typedef int (*RfPlanCreateFn)(void** plan, int placement, int transform_type, int precision, size_t dims, const size_t* lengths, size_t batch, void* desc);
typedef int (*RfPlanWorkSizeFn)(void* plan, size_t* bytes);
typedef int (*RfInfoCreateFn)(void** info);
typedef int (*RfInfoSetStreamFn)(void* info, void* stream);
typedef int (*RfInfoSetWorkFn)(void* info, void* buf, size_t bytes);
typedef int (*RfExecuteFn)(void* plan, void** in, void** out, void* info);

struct RocFFT {
  void* lib = nullptr;
  RfSetupFn setup = nullptr; RfPlanCreateFn plan_create = nullptr; RfPlanWorkSizeFn work_size = nullptr; RfInfoCreateFn info_create = nullptr;
  RfInfoSetStreamFn set_stream = nullptr; RfInfoSetWorkFn set_work = nullptr; RfExecuteFn execute = nullptr;
  std::string why;
};
struct FftPlan { void* plan = nullptr; void* info = nullptr; void* work = nullptr; size_t work_bytes = 0; };

RocFFT& rocfft() {
  static RocFFT r;
  if (r.lib || !r.why.empty()) return r;
  const char* forced = getenv("PCNN_ROCFFT_LIBRARY");
  for (const char* name : {forced ? forced : "librocfft.so.0", "librocfft.so", "/opt/rocm/lib/librocfft.so"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.lib || forced) break;
  }
  if (!r.lib) { const char* e = dlerror(); r.why = std::string("cannot load librocfft.so: ") + (e ? e : "unknown dlopen error"); return r; }
  r.setup = reinterpret_cast<RfSetupFn>(dlsym(r.lib, "rocfft_setup"));
  r.plan_create = reinterpret_cast<RfPlanCreateFn>(dlsym(r.lib, "rocfft_plan_create"));
  r.work_size = reinterpret_cast<RfPlanWorkSizeFn>(dlsym(r.lib, "rocfft_plan_get_work_buffer_size"));
  r.info_create = reinterpret_cast<RfInfoCreateFn>(dlsym(r.lib, "rocfft_execution_info_create"));
  r.set_stream = reinterpret_cast<RfInfoSetStreamFn>(dlsym(r.lib, "rocfft_execution_info_set_stream"));
  r.set_work = reinterpret_cast<RfInfoSetWorkFn>(dlsym(r.lib, "rocfft_execution_info_set_work_buffer"));
  r.execute = reinterpret_cast<RfExecuteFn>(dlsym(r.lib, "rocfft_execute"));
  if (!r.setup || !r.plan_create || !r.work_size || !r.info_create || !r.set_stream || !r.set_work || !r.execute) {
    r.why = "librocfft.so lacks an expected rocfft_* symbol"; dlclose(r.lib); r.lib = nullptr; return r;
  }
  if (r.setup() != 0) { r.why = "rocfft_setup failed"; dlclose(r.lib); r.lib = nullptr; }
  return r;
}

// batched real 2-D forward transform of (N, Lh, Lw) doubles -> (N, Lh, Lw/2+1) complex.  One plan + execution info + work buffer per (device, STREAM,
// Lh, Lw, N), kept for the process: the execution info carries the stream and the work buffer, so two handles on different streams that solve the same
// shape concurrently must not share one.  A plan enters the cache only when every step of its construction has succeeded.
int fft_plan(pcnn_handle h, int Lh, int Lw, int N, FftPlan** out) {
  static std::map<std::tuple<int, void*, int, int, int>, FftPlan> plans;
  static std::mutex mu;
  RocFFT& r = rocfft();
  if (!r.lib) PCNN_FAIL(h, "pcnn_fd_poisson_fft: %s", r.why.c_str());
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(h->device, (void*)h->stream, Lh, Lw, N);
  auto it = plans.find(key);
  if (it == plans.end()) {
    FftPlan p;
    typedef int (*RfDestroyFn)(void*);
    auto drop = [&]() {                                             // a half-built plan never reaches the cache
      RfDestroyFn pd = reinterpret_cast<RfDestroyFn>(dlsym(r.lib, "rocfft_plan_destroy")), id = reinterpret_cast<RfDestroyFn>(dlsym(r.lib, "rocfft_execution_info_destroy"));
      if (p.work) (void)hipFree(p.work);
      if (p.info && id) id(p.info);
      if (p.plan && pd) pd(p.plan);
    };
    const size_t lengths[2] = {(size_t)Lw, (size_t)Lh};            // rocFFT: fastest dimension first
    // placement 1 = not in place, transform type 2 = real forward, precision 1 = double (rocfft.h enums)
    if (r.plan_create(&p.plan, 1, 2, 1, 2, lengths, (size_t)N, nullptr) != 0) PCNN_FAIL(h, "pcnn_fd_poisson_fft: rocfft_plan_create(%d x %d x %d) failed", N, Lh, Lw);
    if (r.info_create(&p.info) != 0) { drop(); PCNN_FAIL(h, "pcnn_fd_poisson_fft: rocfft_execution_info_create failed"); }
    if (r.work_size(p.plan, &p.work_bytes) != 0) { drop(); PCNN_FAIL(h, "pcnn_fd_poisson_fft: rocfft_plan_get_work_buffer_size failed"); }
    if (p.work_bytes) {
      if (hipMalloc(&p.work, p.work_bytes) != hipSuccess) { p.work = nullptr; drop(); PCNN_FAIL(h, "pcnn_fd_poisson_fft: cannot allocate %zu B of rocFFT work buffer", p.work_bytes); }
      if (r.set_work(p.info, p.work, p.work_bytes) != 0) { drop(); PCNN_FAIL(h, "pcnn_fd_poisson_fft: rocfft_execution_info_set_work_buffer failed"); }
    }
    if (r.set_stream(p.info, h->stream) != 0) { drop(); PCNN_FAIL(h, "pcnn_fd_poisson_fft: rocfft_execution_info_set_stream failed"); }
    it = plans.emplace(key, p).first;
  }
  *out = &it->second;
  return 0;
}

// E[n][i][j] = odd-odd extension of src(n, ii, jj) to (2 (nh + 1)) x (2 (nw + 1)): zero on rows / columns 0 and n + 1, mirrored with a sign change beyond
template <typename Src>
__device__ __forceinline__ void odd_extend(int64_t total, int nh, int nw, double* __restrict__ E, Src src) {
  const int Lh = 2 * (nh + 1), Lw = 2 * (nw + 1);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % Lw; const int i = (idx / Lw) % Lh; const int n = idx / ((int64_t)Lh * Lw);
    double v = 0.0;
    if (i != 0 && i != nh + 1 && j != 0 && j != nw + 1) {
      const int ii = i <= nh ? i - 1 : 2 * nh + 1 - i, jj = j <= nw ? j - 1 : 2 * nw + 1 - j;
      v = src(n, ii, jj);
      if ((i > nh) != (j > nw)) v = -v;
    }
    E[idx] = v;
  }
}
__global__ void dst_extend_rhs_kernel(int N, int nh, int nw, const double* __restrict__ B, double* __restrict__ E) {
  odd_extend((int64_t)N * 4 * (nh + 1) * (nw + 1), nh, nw, E, [&](int n, int ii, int jj) { return B[((int64_t)n * nh + ii) * nw + jj]; });
}
// coefficients (D B D)[a][b] = -Re Ehat[a+1][b+1] / 4, divided by the eigenvalues, extended again
__global__ void dst_extend_coeff_kernel(int N, int nh, int nw, const double2* __restrict__ Eh, const double* __restrict__ lam_h, const double* __restrict__ lam_w,
                                        double* __restrict__ E) {
  const int Lh = 2 * (nh + 1), Cw = nw + 2;                         // Lw / 2 + 1 complex columns
  odd_extend((int64_t)N * 4 * (nh + 1) * (nw + 1), nh, nw, E, [&](int n, int a, int b) {
    return -0.25 * Eh[((int64_t)n * Lh + a + 1) * Cw + b + 1].x / (lam_h[a] + lam_w[b]);
  });
}
__global__ void dst_fft_write_soln_kernel(int N, int H, int W, const double2* __restrict__ Eh, const float* __restrict__ left, const float* __restrict__ right,
                                          const float* __restrict__ bottom, const float* __restrict__ top, float* __restrict__ soln) {
  const int nh = H - 2, nw = W - 2, Lh = 2 * (nh + 1), Cw = nw + 2;
  const double scale = -0.25 * (2.0 / (nh + 1)) * (2.0 / (nw + 1));
  const int64_t total = (int64_t)N * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = idx % W; const int i = (idx / W) % H; const int n = idx / ((int64_t)H * W);
    float v;
    if (i == 0) v = left[(int64_t)n * W + j];
    else if (i == H - 1) v = right[(int64_t)n * W + j];
    else if (j == 0) v = bottom[(int64_t)n * H + i];
    else if (j == W - 1) v = top[(int64_t)n * H + i];
    else v = (float)(scale * Eh[((int64_t)n * Lh + i) * Cw + j].x);     // u[i-1][j-1] sits at Ehat[i][j]
    soln[idx] = v;
  }
}

}  // namespace

extern "C" size_t pcnn_fd_poisson_fft_workspace(int N, int H, int W) {
  if (N < 1 || H < 3 || W < 3) return 0;
  const size_t nh = H - 2, nw = W - 2;
  return ((size_t)N * nh * nw + (size_t)N * 4 * (nh + 1) * (nw + 1) + 2 * (size_t)N * 2 * (nh + 1) * (nw + 2)) * sizeof(double);
}

extern "C" int pcnn_fd_poisson_fft(pcnn_handle h, int N, int H, int W, const float* rhs, const float* left, const float* right, const float* bottom,
                                   const float* top, const float* dx, const double* lam_h, const double* lam_w, void* workspace, float* soln) {
  PCNN_REQUIRE(h, h && rhs && left && right && bottom && top && dx && lam_h && lam_w && workspace && soln, "pcnn_fd_poisson_fft: null argument");
  PCNN_REQUIRE(h, H >= 3 && W >= 3 && N >= 1, "pcnn_fd_poisson_fft: grid %dx%d too small", H, W);
  const int nh = H - 2, nw = W - 2, Lh = 2 * (nh + 1), Lw = 2 * (nw + 1);
  FftPlan* pl;
  if (int rc = fft_plan(h, Lh, Lw, N, &pl)) return rc;
  RocFFT& r = rocfft();
  double* Bm = static_cast<double*>(workspace);
  double* E = Bm + (int64_t)N * nh * nw;
  double2* Eh = reinterpret_cast<double2*>(E + (int64_t)N * Lh * Lw);
  hipLaunchKernelGGL(dst_build_rhs_kernel, grid1d((int64_t)N * nh * nw), dim3(256), 0, h->stream, N, H, W, rhs, left, right, bottom, top, dx, Bm);
  hipLaunchKernelGGL(dst_extend_rhs_kernel, grid1d((int64_t)N * Lh * Lw), dim3(256), 0, h->stream, N, nh, nw, Bm, E);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_fft(extend)");
  void* in[1] = {E}; void* out[1] = {Eh};
  if (r.execute(pl->plan, in, out, pl->info) != 0) PCNN_FAIL(h, "pcnn_fd_poisson_fft: rocfft_execute failed");
  hipLaunchKernelGGL(dst_extend_coeff_kernel, grid1d((int64_t)N * Lh * Lw), dim3(256), 0, h->stream, N, nh, nw, Eh, lam_h, lam_w, E);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_fft(coefficients)");
  if (r.execute(pl->plan, in, out, pl->info) != 0) PCNN_FAIL(h, "pcnn_fd_poisson_fft: rocfft_execute failed");
  hipLaunchKernelGGL(dst_fft_write_soln_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, Eh, left, right, bottom, top, soln);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_fft(write)");
  return 0;
}

extern "C" int pcnn_series_synthesis(pcnn_handle h, int N, int H, int W, int ka, int kb, const float* coef, int trig, int accumulate, float* out) {
  PCNN_REQUIRE(h, h && coef && out && N >= 1 && H >= 1 && W >= 1, "pcnn_series_synthesis: bad argument");
  PCNN_REQUIRE(h, ka >= 1 && kb >= 1 && ka <= SYN_MAXK && kb <= SYN_MAXK, "pcnn_series_synthesis: %dx%d coefficients unsupported (<=%d)", ka, kb, SYN_MAXK);
  const size_t lds = ((size_t)ka * W + (size_t)ka * SYN_ROWS) * sizeof(float);
  PCNN_REQUIRE(h, lds <= 64 * 1024, "pcnn_series_synthesis: W=%d too wide for %d modes", W, ka);
  hipLaunchKernelGGL(series_kernel, dim3(pcnn_cdiv(H, SYN_ROWS), N), dim3(256), lds, h->stream, H, W, ka, kb, coef, trig, accumulate, out);
  PCNN_CHECK_LAUNCH(h, "pcnn_series_synthesis");
  return 0;
}

extern "C" int pcnn_separable_sum(pcnn_handle h, int N, int H, int W, int R, const float* U, const float* V, int accumulate, float* out) {
  PCNN_REQUIRE(h, h && U && V && out && R >= 1, "pcnn_separable_sum: bad argument");
  hipLaunchKernelGGL(separable_sum_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, R, U, V, accumulate, out);
  PCNN_CHECK_LAUNCH(h, "pcnn_separable_sum");
  return 0;
}

extern "C" int pcnn_set_max_magnitude(pcnn_handle h, int N, int64_t per, const float* target, float* x, float* factors) {
  PCNN_REQUIRE(h, h && target && x && N >= 1, "pcnn_set_max_magnitude: bad argument");
  hipLaunchKernelGGL(set_max_magnitude_kernel, dim3(N), dim3(1024), 0, h->stream, per, target, x, factors);
  PCNN_CHECK_LAUNCH(h, "pcnn_set_max_magnitude");
  return 0;
}

extern "C" int pcnn_scale_samples(pcnn_handle h, int N, int64_t per, const float* s, float* x) {
  PCNN_REQUIRE(h, h && s && x, "pcnn_scale_samples: null argument");
  hipLaunchKernelGGL(scale_samples_kernel, grid1d((int64_t)N * per), dim3(256), 0, h->stream, N, per, s, x);
  PCNN_CHECK_LAUNCH(h, "pcnn_scale_samples");
  return 0;
}

// Mixed Dirichlet / Neumann solve.  Per axis the host supplies the eigen-decomposition of the 1-D operator on that axis' unknowns
// (M = V diag(lam) V^-1: dataset/__init__.py axis_decomposition): Vinv_h (mh x mh), V_h, lam_h, and for the second axis the TRANSPOSED
// matrices VinvT_w, VT_w (applied from the right).  u = V_h ((Vinv_h B VinvT_w) ./ (lam_h + lam_w)) VT_w.
extern "C" int pcnn_fd_poisson_mixed(pcnn_handle h, int N, int H, int W, int neumann_mask, const float* rhs, const float* left, const float* right,
                                     const float* bottom, const float* top, const float* dx, const double* Vinv_h, const double* V_h,
                                     const double* lam_h, const double* VinvT_w, const double* VT_w, const double* lam_w, double* tmp, float* soln) {
  PCNN_REQUIRE(h, h && rhs && left && right && bottom && top && dx && Vinv_h && V_h && lam_h && VinvT_w && VT_w && lam_w && tmp && soln,
               "pcnn_fd_poisson_mixed: null argument");
  PCNN_REQUIRE(h, H >= 3 && W >= 3 && N >= 1 && neumann_mask >= 0 && neumann_mask < 16, "pcnn_fd_poisson_mixed: bad shape or mask");
  const int mh = H - 2 + (neumann_mask & 1) + ((neumann_mask >> 1) & 1), mw = W - 2 + ((neumann_mask >> 2) & 1) + ((neumann_mask >> 3) & 1);
  const int64_t per = (int64_t)mh * mw;
  double* Bm = tmp; double* T = tmp + (int64_t)N * per;
  hipLaunchKernelGGL(mixed_build_rhs_kernel, grid1d(N * per), dim3(256), 0, h->stream, N, H, W, neumann_mask, rhs, left, right, bottom, top, dx, Bm);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_mixed(build)");
  int rc;
  if ((rc = pcnn_batched_gemm_f64(h, N, mh, mw, mh, Vinv_h, 0, mh, Bm, per, mw, T, per, mw))) return rc;
  if ((rc = pcnn_batched_gemm_f64(h, N, mh, mw, mw, T, per, mw, VinvT_w, 0, mw, Bm, per, mw))) return rc;
  hipLaunchKernelGGL(mixed_divide_kernel, grid1d(N * per), dim3(256), 0, h->stream, N, mh, mw, lam_h, lam_w, Bm);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_mixed(divide)");
  if ((rc = pcnn_batched_gemm_f64(h, N, mh, mw, mh, V_h, 0, mh, Bm, per, mw, T, per, mw))) return rc;
  if ((rc = pcnn_batched_gemm_f64(h, N, mh, mw, mw, T, per, mw, VT_w, 0, mw, Bm, per, mw))) return rc;
  hipLaunchKernelGGL(mixed_write_soln_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, neumann_mask, Bm, left, right, bottom, top, soln);
  PCNN_CHECK_LAUNCH(h, "pcnn_fd_poisson_mixed(write)");
  return 0;
}
