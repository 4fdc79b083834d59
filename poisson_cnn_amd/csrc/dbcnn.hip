// Kernels specific to Dirichlet_BC_NN_Legacy_2 (models/Dirichlet_BC_NN_Legacy.py:14-187) and Poisson_CNN_Legacy
// (models/Poisson_CNN_Legacy.py:5-71): boundary-input assembly, average spatial-pyramid pooling, the
// einsum('bmy,mx,bm->bmxy') expansion of the boundary features into the domain (+ positional embeddings), the adjoint of
// set_max_magnitude_in_batch, the boundary-row overwrite, and flip_and_rotate_tensor for 2-D single-channel fields.
// All are streaming / small-reduction kernels; 1-D tensors are NHWC with H = 1.
#include "pcnn_internal.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;

static dim3 grid1d(int64_t total, int block = 256, int maxb = 16384) {
  int64_t b = pcnn_cdiv64(total, block);
  if (b > maxb) b = maxb;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

// out[n, y, :] = {bc[n, y], cos(pi * 0) = 1, cos(pi * y / (L - 1))}: tf.concat([bc, pos_embeddings_nd[..., 0, :]], 1)
// (Dirichlet_BC_NN_Legacy.py:113-124,136-139).  tf.linspace(0, 1, n)[i] = i / (n - 1) (0 for n = 1).
__global__ void assemble_bc_input_kernel(int N, int L, const float* __restrict__ bc, float* __restrict__ out, int ldo) {
  const int64_t total = (int64_t)N * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i % L);
    const float t = L > 1 ? (float)y / (float)(L - 1) : 0.f;
    float* o = out + i * ldo;
    o[0] = bc[i]; o[1] = 1.0f; o[2] = cosf(PI_F * t);
  }
}

// ---------------------------------------------------------------- SPP average over (bin, channels)
__global__ __launch_bounds__(256) void spp_avg_fwd_kernel(int H, int W, int C, int ldx, int nb, const int32_t* __restrict__ bins,
                                                          const float* __restrict__ x, float* __restrict__ out) {
  __shared__ float red[256];
  const int n = blockIdx.y, b = blockIdx.x;
  const int y0 = bins[4 * b], y1 = bins[4 * b + 1], x0 = bins[4 * b + 2], x1 = bins[4 * b + 3];
  const int ww = x1 - x0, cnt = (y1 - y0) * ww * C;
  float s = 0.f;
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    const int c = e % C, q = e / C, yy = y0 + q / ww, xx = x0 + q % ww;
    s += x[((int64_t)n * H * W + (int64_t)yy * W + xx) * ldx + c];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n * nb + b] = red[0] / (float)cnt;
}

// dx[n, y, x, c] = sum over the bins b containing (y, x) of dout[n, b] / count_b  (every pyramid level covers each pixel once)
__global__ void spp_avg_bwd_kernel(int N, int H, int W, int C, int lddx, int nb, const int32_t* __restrict__ bins, const float* __restrict__ dout,
                                   float* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); const int64_t r = i / W; const int yy = (int)(r % H); const int n = (int)(r / H);
    float g = 0.f;
    for (int b = 0; b < nb; ++b) {
      const int y0 = bins[4 * b], y1 = bins[4 * b + 1], x0 = bins[4 * b + 2], x1 = bins[4 * b + 3];
      if (yy >= y0 && yy < y1 && xx >= x0 && xx < x1) g += dout[n * nb + b] / (float)((y1 - y0) * (x1 - x0) * C);
    }
    for (int c = 0; c < C; ++c) dx[i * lddx + c] = g;
  }
}

// ---------------------------------------------------------------- einsum('bmy,mx,bm->bmxy') + positional embeddings
// out[n, x, y, m] = f[n, y, m] * sh[m, x] * d[n, m] for m < M; out[n, x, y, M] = cos(pi x/(X-1)); out[n, x, y, M+1] = cos(pi y/(L-1))
// (Dirichlet_BC_NN_Legacy.py:150-156; generate_position_embeddings :113-124)
__global__ void dbc_expand_fwd_kernel(int N, int X, int L, int M, const float* __restrict__ f, int ldf, const float* __restrict__ sh,
                                      const float* __restrict__ d, float* __restrict__ out, int ldo) {
  const int MC = M + 2;
  const int64_t total = (int64_t)N * X * L * MC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i % MC); int64_t r = i / MC; const int y = (int)(r % L); r /= L; const int x = (int)(r % X); const int n = (int)(r / X);
    float v;
    if (m < M) v = f[((int64_t)n * L + y) * ldf + m] * sh[m * X + x] * d[n * M + m];
    else if (m == M) v = cosf(PI_F * (X > 1 ? (float)x / (float)(X - 1) : 0.f));
    else v = cosf(PI_F * (L > 1 ? (float)y / (float)(L - 1) : 0.f));
    out[(((int64_t)n * X + x) * L + y) * ldo + m] = v;
  }
}

// t[n, y, m] = sum_x dout[n, x, y, m] * sh[m, x]; df[n, y, m] = t * d[n, m]; tf_[n, y, m] = t * f[n, y, m] (summed over y by the next kernel)
__global__ void dbc_expand_bwd_kernel(int N, int X, int L, int M, const float* __restrict__ dout, int lddo, const float* __restrict__ f, int ldf,
                                      const float* __restrict__ sh, const float* __restrict__ d, float* __restrict__ df, int lddf,
                                      float* __restrict__ tf_ /*[N][L][M]*/) {
  const int64_t total = (int64_t)N * L * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i % M); const int64_t r = i / M; const int y = (int)(r % L); const int n = (int)(r / L);
    float t = 0.f;
    for (int x = 0; x < X; ++x) t += dout[(((int64_t)n * X + x) * L + y) * lddo + m] * sh[m * X + x];
    df[((int64_t)n * L + y) * lddf + m] = t * d[n * M + m];
    tf_[i] = t * f[((int64_t)n * L + y) * ldf + m];
  }
}

// dd[n, m] = sum_y tf_[n, y, m]
__global__ void dbc_expand_bwd_dense_kernel(int N, int L, int M, const float* __restrict__ tf_, float* __restrict__ dd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * M) return;
  const int n = i / M, m = i - n * M;
  float s = 0.f;
  for (int y = 0; y < L; ++y) s += tf_[((int64_t)n * L + y) * M + m];
  dd[i] = s;
}

// ---------------------------------------------------------------- adjoint of x -> x * target / max|x| (per sample)
// y = x s, s = T / A, A = max|x| attained at K positions: dx_j = s g_j - [|x_j| = A] sign(x_j) (sum_i g_i x_i) T / (A^2 K)
// (tf.reduce_max splits its gradient evenly over ties)
__global__ __launch_bounds__(1024) void set_max_magnitude_bwd_kernel(int64_t per, const float* __restrict__ target, const float* __restrict__ x,
                                                                     const float* __restrict__ dy, float* __restrict__ dx) {
  __shared__ float red[1024];
  __shared__ float red2[1024];
  const int n = blockIdx.x;
  const float* xs = x + (int64_t)n * per; const float* gs = dy + (int64_t)n * per; float* o = dx + (int64_t)n * per;
  float mx = 0.f, dot = 0.f;
  for (int64_t q = threadIdx.x; q < per; q += blockDim.x) { mx = fmaxf(mx, fabsf(xs[q])); dot += gs[q] * xs[q]; }
  red[threadIdx.x] = mx; red2[threadIdx.x] = dot;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); red2[threadIdx.x] += red2[threadIdx.x + s]; }
    __syncthreads();
  }
  const float A = red[0]; dot = red2[0];
  __syncthreads();
  float cnt = 0.f;
  for (int64_t q = threadIdx.x; q < per; q += blockDim.x) cnt += fabsf(xs[q]) == A ? 1.f : 0.f;
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float K = red[0], T = target[n];
  const float s = T / A, corr = dot * T / (A * A * K);
  for (int64_t q = threadIdx.x; q < per; q += blockDim.x) {
    const float v = xs[q];
    o[q] = s * gs[q] - (fabsf(v) == A ? (v > 0.f ? corr : -corr) : 0.f);
  }
}

// y[n, 0, :] = bc[n, :]: tf.concat([expand_dims(bc), out[..., 1:, :]], 2) (Dirichlet_BC_NN_Legacy.py:164); zero = the adjoint
__global__ void set_first_row_kernel(int N, int X, int L, const float* __restrict__ bc, float* __restrict__ y) {
  const int64_t total = (int64_t)N * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / L; const int c = (int)(i - n * L);
    y[n * X * L + c] = bc ? bc[i] : 0.f;
  }
}

// ---------------------------------------------------------------- flip_and_rotate_tensor, 2-D fields (N, H, W) -> (N, Ho, Wo)
// out[n, a, b] (+)= alpha[n] * in[n, src(a, b)] with src = (transpose ? (b, a) : (a, b)), then index reversal per flag
__global__ void flip_rotate_kernel(int N, int Ho, int Wo, int transpose, int flip_y, int flip_x, const float* __restrict__ in,
                                   const float* __restrict__ alpha, int accumulate, float* __restrict__ out) {
  const int Hi = transpose ? Wo : Ho, Wi = transpose ? Ho : Wo;
  (void)Hi;
  const int64_t total = (int64_t)N * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % Wo); const int64_t r = i / Wo; const int a = (int)(r % Ho); const int n = (int)(r / Ho);
    const int a2 = flip_y ? Ho - 1 - a : a, b2 = flip_x ? Wo - 1 - b : b;     // undo the reversal (applied after the transpose)
    const int sy = transpose ? b2 : a2, sx = transpose ? a2 : b2;
    const float v = (alpha ? alpha[n] : 1.f) * in[((int64_t)n * (transpose ? Wo : Ho) + sy) * Wi + sx];
    out[i] = accumulate ? out[i] + v : v;
  }
}

}  // namespace

extern "C" int pcnn_dbc_assemble_input(pcnn_handle h, int N, int L, const float* bc, float* out, int ldo) {
  PCNN_REQUIRE(h, h && bc && out && N >= 1 && L >= 1 && ldo >= 3, "pcnn_dbc_assemble_input: bad argument");
  hipLaunchKernelGGL(assemble_bc_input_kernel, grid1d((int64_t)N * L), dim3(256), 0, h->stream, N, L, bc, out, ldo);
  PCNN_CHECK_LAUNCH(h, "pcnn_dbc_assemble_input");
  return 0;
}

extern "C" int pcnn_spp_avg_fwd(pcnn_handle h, int N, int H, int W, int C, int ldx, int nb, const int32_t* bins, const float* x, float* out) {
  PCNN_REQUIRE(h, h && bins && x && out && nb >= 1 && ldx >= C, "pcnn_spp_avg_fwd: bad argument");
  hipLaunchKernelGGL(spp_avg_fwd_kernel, dim3(nb, N), dim3(256), 0, h->stream, H, W, C, ldx, nb, bins, x, out);
  PCNN_CHECK_LAUNCH(h, "pcnn_spp_avg_fwd");
  return 0;
}

extern "C" int pcnn_spp_avg_bwd(pcnn_handle h, int N, int H, int W, int C, int lddx, int nb, const int32_t* bins, const float* dout, float* dx) {
  PCNN_REQUIRE(h, h && bins && dout && dx && nb >= 1 && lddx >= C, "pcnn_spp_avg_bwd: bad argument");
  hipLaunchKernelGGL(spp_avg_bwd_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, C, lddx, nb, bins, dout, dx);
  PCNN_CHECK_LAUNCH(h, "pcnn_spp_avg_bwd");
  return 0;
}

extern "C" int pcnn_dbc_expand_fwd(pcnn_handle h, int N, int X, int L, int M, const float* f, int ldf, const float* sinh_table, const float* d,
                                   float* out, int ldo) {
  PCNN_REQUIRE(h, h && f && sinh_table && d && out && M >= 1 && ldf >= M && ldo >= M + 2, "pcnn_dbc_expand_fwd: bad argument");
  hipLaunchKernelGGL(dbc_expand_fwd_kernel, grid1d((int64_t)N * X * L * (M + 2)), dim3(256), 0, h->stream, N, X, L, M, f, ldf, sinh_table, d, out, ldo);
  PCNN_CHECK_LAUNCH(h, "pcnn_dbc_expand_fwd");
  return 0;
}

extern "C" size_t pcnn_dbc_expand_bwd_workspace(int N, int L, int M) { return (size_t)N * L * M * sizeof(float); }

extern "C" int pcnn_dbc_expand_bwd(pcnn_handle h, int N, int X, int L, int M, const float* dout, int lddo, const float* f, int ldf,
                                   const float* sinh_table, const float* d, float* df, int lddf, float* dd, void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, h && dout && f && sinh_table && d && df && dd && workspace, "pcnn_dbc_expand_bwd: null argument");
  PCNN_REQUIRE(h, lddo >= M && ldf >= M && lddf >= M, "pcnn_dbc_expand_bwd: channel stride smaller than the mode count");
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_dbc_expand_bwd_workspace(N, L, M), "pcnn_dbc_expand_bwd: workspace too small");
  float* tf_ = static_cast<float*>(workspace);
  hipLaunchKernelGGL(dbc_expand_bwd_kernel, grid1d((int64_t)N * L * M), dim3(256), 0, h->stream, N, X, L, M, dout, lddo, f, ldf, sinh_table, d, df, lddf, tf_);
  PCNN_CHECK_LAUNCH(h, "pcnn_dbc_expand_bwd");
  hipLaunchKernelGGL(dbc_expand_bwd_dense_kernel, dim3(pcnn_cdiv(N * M, 64)), dim3(64), 0, h->stream, N, L, M, tf_, dd);
  PCNN_CHECK_LAUNCH(h, "pcnn_dbc_expand_bwd(dense)");
  return 0;
}

extern "C" int pcnn_set_max_magnitude_bwd(pcnn_handle h, int N, int64_t per, const float* target, const float* x, const float* dy, float* dx) {
  PCNN_REQUIRE(h, h && target && x && dy && dx && N >= 1 && per >= 1, "pcnn_set_max_magnitude_bwd: bad argument");
  hipLaunchKernelGGL(set_max_magnitude_bwd_kernel, dim3(N), dim3(1024), 0, h->stream, per, target, x, dy, dx);
  PCNN_CHECK_LAUNCH(h, "pcnn_set_max_magnitude_bwd");
  return 0;
}

extern "C" int pcnn_set_first_row(pcnn_handle h, int N, int X, int L, const float* bc, float* y) {
  PCNN_REQUIRE(h, h && y && N >= 1 && X >= 1 && L >= 1, "pcnn_set_first_row: bad argument");
  hipLaunchKernelGGL(set_first_row_kernel, grid1d((int64_t)N * L), dim3(256), 0, h->stream, N, X, L, bc, y);
  PCNN_CHECK_LAUNCH(h, "pcnn_set_first_row");
  return 0;
}

extern "C" int pcnn_flip_rotate(pcnn_handle h, int N, int Ho, int Wo, int transpose, int flip_y, int flip_x, const float* in, const float* alpha,
                                int accumulate, float* out) {
  PCNN_REQUIRE(h, h && in && out && N >= 1 && Ho >= 1 && Wo >= 1, "pcnn_flip_rotate: bad argument");
  hipLaunchKernelGGL(flip_rotate_kernel, grid1d((int64_t)N * Ho * Wo), dim3(256), 0, h->stream, N, Ho, Wo, transpose, flip_y, flip_x, in, alpha, accumulate, out);
  PCNN_CHECK_LAUNCH(h, "pcnn_flip_rotate");
  return 0;
}
