// Transposed convolution with kernel == stride (pixel-shuffle GEMM), small dense layers, spatial-pyramid max pool,
// and the training loss (MAE / MSE / integral-Lp / FD-Laplacian residual) forward + backward.
#include <math.h>
#include "pcnn_internal.h"

// MFMA paths (deconv_mfma.hip): return -1 when the shape is outside their coverage
int pcnn_deconv_fwd_mfma(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx, const float* k,
                         const float* bias, float alpha, float beta, float* y, int ldy);
int pcnn_deconv_bwd_data_mfma(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* dy, int lddy, const float* k,
                              float alpha, float* dx, int lddx);
int pcnn_deconv_bwd_filter_mfma(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx, const float* dy,
                                int lddy, float* partial, int* S_out);

namespace {

static dim3 grid1d(int64_t total, int block = 256, int maxb = 16384) {
  int64_t b = pcnn_cdiv64(total, block);
  if (b > maxb) b = maxb;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

// ---------------------------------------------------------------- deconv (k == stride)
// y[n,Y,X,co] = beta*y + alpha*(bias[co] + sum_ci x[n,(Y+py)/f,(X+px)/f,ci] * k[(Y+py)%f,(X+px)%f,co,ci])
__global__ void deconv_fwd_kernel(int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, int py, int px, const float* __restrict__ x, int ldx,
                                  const float* __restrict__ k, const float* __restrict__ bias, float alpha, float beta, float* __restrict__ y, int ldy) {
  const int64_t total = (int64_t)N * H * W * Cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = i % Cout; int64_t r = i / Cout; const int X = r % W; r /= W; const int Y = r % H; const int n = r / H;
    const int yc = (Y + py) / f, ty = (Y + py) % f, xc = (X + px) / f, tx = (X + px) % f;
    const float* xv = x + (((int64_t)n * hc + yc) * wc + xc) * ldx;
    const float* kv = k + (((int64_t)ty * f + tx) * Cout + co) * Cin;
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xv[ci], kv[ci], acc);
    float* dst = &y[(((int64_t)n * H + Y) * W + X) * ldy + co];
    *dst = beta == 0.f ? alpha * acc : beta * *dst + alpha * acc;
  }
}

// dx[n,yc,xc,ci] = alpha * sum_{ty,tx,co} dy[n, yc*f+ty-py, xc*f+tx-px, co] * k[ty,tx,co,ci]
__global__ void deconv_bwd_data_kernel(int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, int py, int px, const float* __restrict__ dy,
                                       int lddy, const float* __restrict__ k, float alpha, float* __restrict__ dx, int lddx) {
  const int64_t total = (int64_t)N * hc * wc * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = i % Cin; int64_t r = i / Cin; const int xc = r % wc; r /= wc; const int yc = r % hc; const int n = r / hc;
    float acc = 0.f;
    for (int ty = 0; ty < f; ++ty) {
      const int Y = yc * f + ty - py;
      if (Y < 0 || Y >= H) continue;
      for (int tx = 0; tx < f; ++tx) {
        const int X = xc * f + tx - px;
        if (X < 0 || X >= W) continue;
        const float* g = dy + (((int64_t)n * H + Y) * W + X) * lddy;
        const float* kv = k + ((int64_t)ty * f + tx) * Cout * Cin + ci;
        for (int co = 0; co < Cout; ++co) acc = fmaf(g[co], kv[(int64_t)co * Cin], acc);
      }
    }
    dx[(((int64_t)n * hc + yc) * wc + xc) * lddx + ci] = alpha * acc;
  }
}

// partial[split][tap][co][ci] = sum over the split's coarse pixels of dy[.., co] * x[.., ci]; grid (S, f*f)
__global__ __launch_bounds__(256) void deconv_bwd_filter_kernel(int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, int py, int px,
                                                                const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                                float* __restrict__ partial) {
  const int split = blockIdx.x, S = gridDim.x, tap = blockIdx.y, ty = tap / f, tx = tap % f;
  const int64_t npix = (int64_t)N * hc * wc;
  const int nout = Cout * Cin;
  // each thread owns outputs e = tid, tid+256, ... (co = e / Cin, ci = e % Cin); <= 16 per thread for 64x64
  float acc[16];
  int cnt = 0;
  for (int e = threadIdx.x; e < nout && cnt < 16; e += 256) acc[cnt++] = 0.f;
  for (int64_t p = split; p < npix; p += S) {
    int64_t r = p; const int xc = r % wc; r /= wc; const int yc = r % hc; const int n = r / hc;
    const int Y = yc * f + ty - py, X = xc * f + tx - px;
    if (Y < 0 || Y >= H || X < 0 || X >= W) continue;
    const float* g = dy + (((int64_t)n * H + Y) * W + X) * lddy;
    const float* xv = x + p * ldx;
    int q = 0;
    for (int e = threadIdx.x; e < nout && q < 16; e += 256, ++q) acc[q] = fmaf(g[e / Cin], xv[e % Cin], acc[q]);
  }
  float* dst = partial + ((int64_t)split * f * f + tap) * nout;
  int q = 0;
  for (int e = threadIdx.x; e < nout && q < 16; e += 256, ++q) dst[e] = acc[q];
}

__global__ void reduce_splits_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t nel, int S, float alpha) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nel; e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += ws[(int64_t)k * nel + e];
    out[e] = alpha * s;
  }
}

constexpr int DECONV_SPLITS = 256;

// ---------------------------------------------------------------- dense
__global__ void dense_fwd_kernel(int N, int In, int Out, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                 int act, float alpha, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * Out) return;
  const int n = i / Out, o = i % Out;
  float acc = b ? b[o] : 0.f;
  for (int k = 0; k < In; ++k) acc = fmaf(x[n * In + k], w[k * Out + o], acc);
  y[i] = pcnn_act(acc, act, alpha);
}

// grid-stride over output elements (every dx / dw / db element is independent); dz computed on the fly
__global__ __launch_bounds__(256) void dense_bwd_kernel(int N, int In, int Out, const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ y, const float* __restrict__ dy, int act, float alpha,
                                                        float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db) {
  auto dz = [&](int n, int o) { return dy[n * Out + o] * pcnn_act_grad_from_out(y[n * Out + o], act, alpha); };
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x, ts = gridDim.x * blockDim.x;
  if (dx) {
    // one WAVE per dx element: the lanes stride over the outputs (coalesced rows of dy, y and w), then a fixed-order butterfly sum - the
    // hyper-networks of the metalearning layers have Out = k k Cin Cout (10 816 for 13 x 13 x 8 x 8) against N In = a few hundred, and one
    // thread per element walked each row serially at a stride of Out floats (7.6 ms per call, 60 % of a metalearning training step)
    const int lane = threadIdx.x & 63, wv = t0 >> 6, nw = ts >> 6;
    for (int i = wv; i < N * In; i += nw) {
      const int n = i / In, k = i % In;
      float acc = 0.f;
      for (int o = lane; o < Out; o += 64) acc = fmaf(dz(n, o), w[k * Out + o], acc);
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
      if (lane == 0) dx[i] = acc;
    }
  }
  for (int i = t0; i < In * Out; i += ts) {
    const int k = i / Out, o = i % Out;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(x[n * In + k], dz(n, o), acc);
    dw[i] += acc;
  }
  if (db)                                                 // a Dense layer built with use_bias=False has no bias gradient
    for (int o = t0; o < Out; o += ts) {
      float acc = 0.f;
      for (int n = 0; n < N; ++n) acc += dz(n, o);
      db[o] += acc;
    }
}

// ---------------------------------------------------------------- SPP max over (bin, channels)
__global__ __launch_bounds__(256) void spp_max_fwd_kernel(int H, int W, int C, int nb, const int32_t* __restrict__ bins, const float* __restrict__ x,
                                                          float* __restrict__ out, int32_t* __restrict__ argmax) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const int n = blockIdx.y, b = blockIdx.x;
  const int y0 = bins[4 * b], y1 = bins[4 * b + 1], x0 = bins[4 * b + 2], x1 = bins[4 * b + 3];
  const int ww = x1 - x0, cnt = (y1 - y0) * ww * C;
  float best = -INFINITY; int besti = 0x7fffffff;
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    const int c = e % C, q = e / C, yy = y0 + q / ww, xx = x0 + q % ww;
    const int flat = (yy * W + xx) * C + c;
    const float v = x[(int64_t)n * H * W * C + flat];
    if (v > best || (v == best && flat < besti)) { best = v; besti = flat; }
  }
  bv[threadIdx.x] = best; bi[threadIdx.x] = besti;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float ov = bv[threadIdx.x + s]; const int oi = bi[threadIdx.x + s];
      if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  // a bin that holds nothing but NaN never passes `v > best`: its maximum is NaN (tf.reduce_max propagates it) and its argmax must still be an
  // index INSIDE the tensor - the backward kernel writes through it
  if (threadIdx.x == 0) {
    const bool none = bi[0] == 0x7fffffff;
    out[n * nb + b] = none ? __builtin_nanf("") : bv[0];
    argmax[n * nb + b] = none ? (y0 * W + x0) * C : bi[0];
  }
}

__global__ void spp_max_bwd_kernel(int N, int64_t per, int nb, const int32_t* __restrict__ argmax, const float* __restrict__ dout, float* __restrict__ dx) {
  // one thread per sample: bins of different pyramid levels overlap, so accumulate serially (nb <= a few dozen)
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int b = 0; b < nb; ++b) {
    const int a = argmax[n * nb + b];
    if ((unsigned)a < (uint64_t)per) dx[(int64_t)n * per + a] += dout[n * nb + b];          // (an index outside the sample is never written through)
  }
}

// ---------------------------------------------------------------- loss
// one workgroup per sample: out[n] = {sum|p-t|, sum (p-t)^2, sum G (p-t)^2, max|t|}
// per-sample sums of |d|, d^2, G (t - p)^lp and max|t|: LOSS_SPLITS workgroups per sample, combined in a fixed order by a second launch
constexpr int LOSS_SPLITS = 64;
__global__ __launch_bounds__(1024) void loss_partials_kernel(int64_t hw, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                             const float* __restrict__ G, float lp, float* __restrict__ part) {
  __shared__ float red[4][1024];
  const int n = blockIdx.y, sp = blockIdx.x;
  const int64_t per = (hw + LOSS_SPLITS - 1) / LOSS_SPLITS, q0 = sp * per, q1 = q0 + per < hw ? q0 + per : hw;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, mx = 0.f;
  for (int64_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
    const float t = tgt[(int64_t)n * hw + q], d = pred[(int64_t)n * hw + q] - t;
    s0 += fabsf(d); s1 += d * d;
    if (G) s2 += G[q] * (lp == 2.0f ? d * d : powf(-d, lp));          // integral_loss.py:153: (y_true - y_pred) ** Lp_norm_power
    mx = fmaxf(mx, fabsf(t));
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = mx;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s];
      red[2][threadIdx.x] += red[2][threadIdx.x + s]; red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* o = part + ((int64_t)n * LOSS_SPLITS + sp) * 4;
    o[0] = red[0][0]; o[1] = red[1][0]; o[2] = red[2][0]; o[3] = red[3][0];
  }
}

__global__ void loss_partials_final_kernel(const float* __restrict__ part, float* __restrict__ out) {
  const int n = blockIdx.x;
  if (threadIdx.x != 0) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, mx = 0.f;
  for (int sp = 0; sp < LOSS_SPLITS; ++sp) {
    const float* p = part + ((int64_t)n * LOSS_SPLITS + sp) * 4;
    s0 += p[0]; s1 += p[1]; s2 += p[2]; mx = fmaxf(mx, p[3]);
  }
  out[4 * n] = s0; out[4 * n + 1] = s1; out[4 * n + 2] = s2; out[4 * n + 3] = mx;
}

__global__ void loss_bwd_kernel(int N, int64_t hw, const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ G,
                                const float* __restrict__ c_mae, const float* __restrict__ c_mse, const float* __restrict__ c_int, float lp,
                                float* __restrict__ dpred) {
  const int64_t total = (int64_t)N * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = i / hw; const int64_t q = i % hw;
    const float d = pred[i] - tgt[i];
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    float g = c_mae[n] * sg + 2.0f * d * c_mse[n];
    if (G) g += c_int[n] * G[q] * (lp == 2.0f ? 2.0f * d : -lp * powf(-d, lp - 1.0f));        // d/dpred of (target - pred)^lp
    dpred[i] = g;
  }
}

// FD-Laplacian residual: out[n] = sum over interior of (rhs - sum_{ij} kern[n,i,j] pred[y+i-s/2, x+j-s/2])^2
__global__ __launch_bounds__(1024) void pi_partials_kernel(int H, int W, int s, const float* __restrict__ pred, const float* __restrict__ rhs,
                                                           const float* __restrict__ kern, float* __restrict__ out) {
  __shared__ float red[1024];
  const int n = blockIdx.x, hs = s / 2, Hi = H - 2 * hs, Wi = W - 2 * hs;
  const float* kn = kern + (int64_t)n * s * s;
  const float* pn = pred + (int64_t)n * H * W;
  float acc = 0.f;
  for (int64_t q = threadIdx.x; q < (int64_t)Hi * Wi; q += blockDim.x) {
    const int yy = q / Wi + hs, xx = q % Wi + hs;
    float lap = 0.f;
    for (int i = 0; i < s; ++i)
      for (int j = 0; j < s; ++j) {
        const float kv = kn[i * s + j];
        if (kv != 0.f) lap = fmaf(kv, pn[(int64_t)(yy + i - hs) * W + xx + j - hs], lap);
      }
    const float e = rhs[(int64_t)n * H * W + (int64_t)yy * W + xx] - lap;
    acc += e * e;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = red[0];
}

// dpred[n,y,x] += coef[n] * sum over interior cells (Y,X) touching (y,x): -2 e(Y,X) kern[n, y-Y+hs, x-X+hs]
__global__ void pi_bwd_kernel(int N, int H, int W, int s, const float* __restrict__ pred, const float* __restrict__ rhs, const float* __restrict__ kern,
                              const float* __restrict__ coef, float* __restrict__ dpred) {
  const int hs = s / 2;
  const int64_t total = (int64_t)N * H * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int xx = idx % W; const int yy = (idx / W) % H; const int n = idx / ((int64_t)H * W);
    const float* kn = kern + (int64_t)n * s * s;
    const float* pn = pred + (int64_t)n * H * W;
    float acc = 0.f;
    for (int i = 0; i < s; ++i) {
      const int Y = yy - (i - hs);
      if (Y < hs || Y >= H - hs) continue;
      for (int j = 0; j < s; ++j) {
        const int X = xx - (j - hs);
        if (X < hs || X >= W - hs) continue;
        const float kv = kn[i * s + j];
        if (kv == 0.f) continue;
        float lap = 0.f;
        for (int a = 0; a < s; ++a)
          for (int b = 0; b < s; ++b) {
            const float k2 = kn[a * s + b];
            if (k2 != 0.f) lap = fmaf(k2, pn[(int64_t)(Y + a - hs) * W + X + b - hs], lap);
          }
        const float e = rhs[(int64_t)n * H * W + (int64_t)Y * W + X] - lap;
        acc += -2.0f * e * kv;
      }
    }
    dpred[idx] += coef[n] * acc;
  }
}

}  // namespace

extern "C" int pcnn_deconv_fwd(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx, const float* k,
                               const float* bias, float alpha, float beta, float* y, int ldy) {
  PCNN_REQUIRE(h, h && x && k && y, "pcnn_deconv_fwd: null argument");
  PCNN_REQUIRE(h, f >= 1 && hc == pcnn_cdiv(H, f) && wc == pcnn_cdiv(W, f), "pcnn_deconv_fwd: input must be ceil(output/stride) (SAME transpose)");
  {
    const int rc = pcnn_deconv_fwd_mfma(h, N, hc, wc, Cin, H, W, Cout, f, x, ldx, k, bias, alpha, beta, y, ldy);
    if (rc >= 0) return rc;
  }
  const int py = (hc * f - H) / 2, px = (wc * f - W) / 2;
  hipLaunchKernelGGL(deconv_fwd_kernel, grid1d((int64_t)N * H * W * Cout), dim3(256), 0, h->stream, N, hc, wc, Cin, H, W, Cout, f, py, px, x, ldx, k, bias,
                     alpha, beta, y, ldy);
  PCNN_CHECK_LAUNCH(h, "pcnn_deconv_fwd");
  return 0;
}

extern "C" int pcnn_deconv_bwd_data(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* dy, int lddy,
                                    const float* k, float alpha, float* dx, int lddx) {
  PCNN_REQUIRE(h, h && dy && k && dx, "pcnn_deconv_bwd_data: null argument");
  PCNN_REQUIRE(h, f >= 1 && hc == pcnn_cdiv(H, f) && wc == pcnn_cdiv(W, f), "pcnn_deconv_bwd_data: shape mismatch");
  {
    const int rc = pcnn_deconv_bwd_data_mfma(h, N, hc, wc, Cin, H, W, Cout, f, dy, lddy, k, alpha, dx, lddx);
    if (rc >= 0) return rc;
  }
  const int py = (hc * f - H) / 2, px = (wc * f - W) / 2;
  hipLaunchKernelGGL(deconv_bwd_data_kernel, grid1d((int64_t)N * hc * wc * Cin), dim3(256), 0, h->stream, N, hc, wc, Cin, H, W, Cout, f, py, px, dy, lddy, k,
                     alpha, dx, lddx);
  PCNN_CHECK_LAUNCH(h, "pcnn_deconv_bwd_data");
  return 0;
}

extern "C" size_t pcnn_deconv_wgrad_workspace(int N, int hc, int wc, int Cin, int Cout, int f) {
  (void)N; (void)hc; (void)wc;
  const size_t a = (size_t)DECONV_SPLITS * f * f * Cout * Cin * sizeof(float);
  const size_t b = pcnn_colsum_workspace(Cout);
  return a > b ? a : b;
}

extern "C" int pcnn_deconv_bwd_filter(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx,
                                      const float* dy, int lddy, float alpha, float* dk, float* dbias, void* workspace, size_t workspace_bytes) {
  PCNN_REQUIRE(h, h && x && dy && dk && workspace, "pcnn_deconv_bwd_filter: null argument");
  PCNN_REQUIRE(h, Cin * Cout <= 4096, "pcnn_deconv_bwd_filter: Cin*Cout=%d unsupported (<=4096)", Cin * Cout);
  PCNN_REQUIRE(h, workspace_bytes >= pcnn_deconv_wgrad_workspace(N, hc, wc, Cin, Cout, f), "pcnn_deconv_bwd_filter: workspace too small");
  const int py = (hc * f - H) / 2, px = (wc * f - W) / 2;
  int64_t npix = (int64_t)N * hc * wc;
  int S = (int)(npix < DECONV_SPLITS ? npix : DECONV_SPLITS);
  float* partial = static_cast<float*>(workspace);
  {
    int Sm = 0;
    const int rc = pcnn_deconv_bwd_filter_mfma(h, N, hc, wc, Cin, H, W, Cout, f, x, ldx, dy, lddy, partial, &Sm);
    if (rc > 0) return rc;
    if (rc == 0) S = Sm;
    else {
      hipLaunchKernelGGL(deconv_bwd_filter_kernel, dim3(S, f * f), dim3(256), 0, h->stream, N, hc, wc, Cin, H, W, Cout, f, py, px, x, ldx, dy, lddy, partial);
      PCNN_CHECK_LAUNCH(h, "pcnn_deconv_bwd_filter");
    }
  }
  const int64_t nel = (int64_t)f * f * Cout * Cin;
  hipLaunchKernelGGL(reduce_splits_kernel, grid1d(nel), dim3(256), 0, h->stream, partial, dk, nel, S, alpha);
  PCNN_CHECK_LAUNCH(h, "pcnn_deconv_bwd_filter(reduce)");
  if (dbias) {
    // dbias[co] = alpha * sum over all output pixels of dy: reuse the column-sum machinery (scale folded in afterwards)
    int rc = pcnn_conv2d_epilogue_bwd(h, (int64_t)N * H * W, Cout, dy, lddy, nullptr, 0, nullptr, PCNN_ACT_LINEAR, 0.f, nullptr, 0, dbias, nullptr, nullptr,
                                      workspace, workspace_bytes);
    if (rc) return rc;
    if (alpha != 1.0f) {
      rc = pcnn_axpby(h, 1, Cout, alpha, dbias, Cout, 0.f, dbias, Cout);
      if (rc) return rc;
    }
  }
  return 0;
}

extern "C" int pcnn_dense_fwd(pcnn_handle h, int N, int In, int Out, const float* x, const float* w, const float* b, int act, float alpha, float* y) {
  PCNN_REQUIRE(h, h && x && w && y, "pcnn_dense_fwd: null argument");
  hipLaunchKernelGGL(dense_fwd_kernel, dim3(pcnn_cdiv(N * Out, 128)), dim3(128), 0, h->stream, N, In, Out, x, w, b, act, alpha, y);
  PCNN_CHECK_LAUNCH(h, "pcnn_dense_fwd");
  return 0;
}

extern "C" int pcnn_dense_bwd(pcnn_handle h, int N, int In, int Out, const float* x, const float* w, const float* y, const float* dy, int act,
                              float alpha, float* dx, float* dw, float* db) {
  PCNN_REQUIRE(h, h && x && w && y && dy && dw, "pcnn_dense_bwd: null argument");   // db may be null (no bias)
  const int work = std::max(N * In * (dx ? 64 : 1), In * Out);
  hipLaunchKernelGGL(dense_bwd_kernel, dim3(std::min(pcnn_cdiv(work, 256), 1024)), dim3(256), 0, h->stream, N, In, Out, x, w, y, dy, act, alpha, dx, dw, db);
  PCNN_CHECK_LAUNCH(h, "pcnn_dense_bwd");
  return 0;
}

extern "C" int pcnn_spp_max_fwd(pcnn_handle h, int N, int H, int W, int C, int nb, const int32_t* bins, const float* x, float* out, int32_t* argmax) {
  PCNN_REQUIRE(h, h && bins && x && out && argmax && nb >= 1, "pcnn_spp_max_fwd: bad argument");
  hipLaunchKernelGGL(spp_max_fwd_kernel, dim3(nb, N), dim3(256), 0, h->stream, H, W, C, nb, bins, x, out, argmax);
  PCNN_CHECK_LAUNCH(h, "pcnn_spp_max_fwd");
  return 0;
}

extern "C" int pcnn_spp_max_bwd(pcnn_handle h, int N, int H, int W, int C, int nb, const int32_t* argmax, const float* dout, float* dx) {
  PCNN_REQUIRE(h, h && argmax && dout && dx, "pcnn_spp_max_bwd: null argument");
  hipError_t e = hipMemsetAsync(dx, 0, (size_t)N * H * W * C * sizeof(float), h->stream);
  if (e != hipSuccess) PCNN_FAIL(h, "pcnn_spp_max_bwd: memset: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(spp_max_bwd_kernel, dim3(pcnn_cdiv(N, 64)), dim3(64), 0, h->stream, N, (int64_t)H * W * C, nb, argmax, dout, dx);
  PCNN_CHECK_LAUNCH(h, "pcnn_spp_max_bwd");
  return 0;
}

extern "C" int pcnn_loss_partials_p(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G, float lp_power, float* out) {
  PCNN_REQUIRE(h, h && pred && target && out, "pcnn_loss_partials: null argument");
  // scratch for the split sums: N * 64 * 4 floats from the handle's filter scratch (stream-ordered with every other user of it)
  const size_t need = (size_t)N * LOSS_SPLITS * 4 * sizeof(float);
  if (h->scratch_bytes < need) {
    if (h->scratch) { pcnn_release(h, h->scratch); h->scratch = nullptr; h->scratch_bytes = 0; }
    const size_t cap = need < (4u << 20) ? (4u << 20) : need;
    if (hipMalloc(&h->scratch, cap) != hipSuccess) PCNN_FAIL(h, "pcnn_loss_partials: cannot allocate %zu B of scratch", cap);
    h->scratch_bytes = cap;
  }
  float* part = static_cast<float*>(h->scratch);
  hipLaunchKernelGGL(loss_partials_kernel, dim3(LOSS_SPLITS, N), dim3(1024), 0, h->stream, hw, pred, target, G, lp_power, part);
  hipLaunchKernelGGL(loss_partials_final_kernel, dim3(N), dim3(64), 0, h->stream, part, out);
  PCNN_CHECK_LAUNCH(h, "pcnn_loss_partials");
  return 0;
}

extern "C" int pcnn_loss_partials(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G, float* out) {
  return pcnn_loss_partials_p(h, N, hw, pred, target, G, 2.0f, out);
}

extern "C" int pcnn_loss_bwd_p(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G, const float* c_mae,
                               const float* c_mse, const float* c_int, float lp_power, float* dpred) {
  PCNN_REQUIRE(h, h && pred && target && c_mae && c_mse && c_int && dpred, "pcnn_loss_bwd: null argument");
  hipLaunchKernelGGL(loss_bwd_kernel, grid1d((int64_t)N * hw), dim3(256), 0, h->stream, N, hw, pred, target, G, c_mae, c_mse, c_int, lp_power, dpred);
  PCNN_CHECK_LAUNCH(h, "pcnn_loss_bwd");
  return 0;
}

extern "C" int pcnn_loss_bwd(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G, const float* c_mae,
                             const float* c_mse, const float* c_int, float* dpred) {
  return pcnn_loss_bwd_p(h, N, hw, pred, target, G, c_mae, c_mse, c_int, 2.0f, dpred);
}

extern "C" int pcnn_pi_loss_partials(pcnn_handle h, int N, int H, int W, int s, const float* pred, const float* rhs, const float* kern, float* out) {
  PCNN_REQUIRE(h, h && pred && rhs && kern && out && s % 2 == 1 && H > s && W > s, "pcnn_pi_loss_partials: bad argument");
  hipLaunchKernelGGL(pi_partials_kernel, dim3(N), dim3(1024), 0, h->stream, H, W, s, pred, rhs, kern, out);
  PCNN_CHECK_LAUNCH(h, "pcnn_pi_loss_partials");
  return 0;
}

extern "C" int pcnn_pi_loss_bwd(pcnn_handle h, int N, int H, int W, int s, const float* pred, const float* rhs, const float* kern, const float* coef,
                                float* dpred) {
  PCNN_REQUIRE(h, h && pred && rhs && kern && coef && dpred && s % 2 == 1 && H > s && W > s, "pcnn_pi_loss_bwd: bad argument");
  hipLaunchKernelGGL(pi_bwd_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, h->stream, N, H, W, s, pred, rhs, kern, coef, dpred);
  PCNN_CHECK_LAUNCH(h, "pcnn_pi_loss_bwd");
  return 0;
}

// ---- loss_wrapper bookkeeping (losses/loss_wrapper.py:45-71) on the N per-sample partial sums
namespace {
__global__ void loss_coefficients_kernel(int N, float inv_hw, const float* __restrict__ part, float w_mae, float w_mse, float w_int, float lp, int scale_by_peak,
                                         float inv_gbs, const float* __restrict__ extra /*optional scalar added to the loss*/, float* __restrict__ loss,
                                         float* __restrict__ c_mae, float* __restrict__ c_mse, float* __restrict__ c_int, float* __restrict__ mse) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float L = 0.f, sq = 0.f;
  for (int n = 0; n < N; ++n) {
    const float pk = scale_by_peak ? part[4 * n + 3] : 1.f;
    const float i1 = 1.f / pk, i2 = 1.f / (pk * pk), ip = lp == 2.0f ? i2 : 1.f / powf(pk, lp);      // loss_wrapper.py:68: peak ** Lp_norm_power
    L += (w_mae * part[4 * n] * inv_hw * i1 + w_mse * part[4 * n + 1] * inv_hw * i2 + w_int * part[4 * n + 2] * ip) * inv_gbs;
    c_mae[n] = w_mae * inv_hw * i1 * inv_gbs;
    c_mse[n] = w_mse * inv_hw * i2 * inv_gbs;
    c_int[n] = w_int * ip * inv_gbs;
    sq += part[4 * n + 1];
  }
  if (extra) L += extra[0];
  loss[0] = L;
  mse[0] = sq * inv_hw / (float)N;
}
}  // namespace

extern "C" int pcnn_loss_coefficients_p(pcnn_handle h, int N, int64_t hw, const float* partials, float w_mae, float w_mse, float w_int, float lp_power,
                                        int scale_by_peak, int global_batch_size, const float* extra, float* loss, float* c_mae, float* c_mse, float* c_int,
                                        float* mse) {
  PCNN_REQUIRE(h, h && partials && loss && c_mae && c_mse && c_int && mse && N >= 1 && global_batch_size >= 1, "pcnn_loss_coefficients: bad argument");
  hipLaunchKernelGGL(loss_coefficients_kernel, dim3(1), dim3(64), 0, h->stream, N, 1.0f / (float)hw, partials, w_mae, w_mse, w_int, lp_power, scale_by_peak,
                     1.0f / (float)global_batch_size, extra, loss, c_mae, c_mse, c_int, mse);
  PCNN_CHECK_LAUNCH(h, "pcnn_loss_coefficients");
  return 0;
}

extern "C" int pcnn_loss_coefficients(pcnn_handle h, int N, int64_t hw, const float* partials, float w_mae, float w_mse, float w_int, int scale_by_peak,
                                      int global_batch_size, const float* extra, float* loss, float* c_mae, float* c_mse, float* c_int, float* mse) {
  return pcnn_loss_coefficients_p(h, N, hw, partials, w_mae, w_mse, w_int, 2.0f, scale_by_peak, global_batch_size, extra, loss, c_mae, c_mse, c_int, mse);
}

// ---------------------------------------------------------------- LayerNormalization over the feature axis of an (N, F) matrix
// tf.keras.layers.LayerNormalization() (axis -1, epsilon 1e-3, gamma / beta of length F): the optional last layer of the metalearning
// hyper-networks (layers/metalearning_conv.py:128-129).  One workgroup per row.
namespace {
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int F, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ double red[2][256];
  const int n = blockIdx.x;
  const float* xr = x + (int64_t)n * F;
  double s = 0.0, s2 = 0.0;
  for (int f = threadIdx.x; f < F; f += 256) { const double v = xr[f]; s += v; s2 += v * v; }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  const double m = red[0][0] / F;
  double var = red[1][0] / F - m * m;
  if (var < 0.0) var = 0.0;
  const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
  for (int f = threadIdx.x; f < F; f += 256) y[(int64_t)n * F + f] = (xr[f] - mu) * rs * gamma[f] + beta[f];
  if (threadIdx.x == 0) { mean[n] = mu; rstd[n] = rs; }
}

// dx[n,f] = rstd (g - mean_f(g) - xhat mean_f(g xhat)), g = dy gamma
__global__ __launch_bounds__(256) void layernorm_bwd_x_kernel(int F, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ dy, float* __restrict__ dx) {
  __shared__ double red[2][256];
  const int n = blockIdx.x;
  const float mu = mean[n], rs = rstd[n];
  double s = 0.0, s2 = 0.0;
  for (int f = threadIdx.x; f < F; f += 256) {
    const double g = (double)dy[(int64_t)n * F + f] * gamma[f], xh = ((double)x[(int64_t)n * F + f] - mu) * rs;
    s += g; s2 += g * xh;
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  const float mg = (float)(red[0][0] / F), mgx = (float)(red[1][0] / F);
  for (int f = threadIdx.x; f < F; f += 256) {
    const float g = dy[(int64_t)n * F + f] * gamma[f], xh = (x[(int64_t)n * F + f] - mu) * rs;
    dx[(int64_t)n * F + f] = rs * (g - mg - xh * mgx);
  }
}

__global__ void layernorm_bwd_params_kernel(int N, int F, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                            const float* __restrict__ dy, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float sg = 0.f, sb = 0.f;
  for (int n = 0; n < N; ++n) {
    const float d = dy[(int64_t)n * F + f];
    sg += d * (x[(int64_t)n * F + f] - mean[n]) * rstd[n];
    sb += d;
  }
  dgamma[f] = sg; dbeta[f] = sb;
}
}  // namespace

// ---------------------------------------------------------------- softmax over the feature axis of an (N, F) matrix
// tf.keras activation 'softmax' of a Dense layer (the last domain-info layer of models/Dirichlet_BC_NN_Metalearning.py:69-76,236-239).
// One wavefront per row; bwd: dx = y (dy - sum_f dy y).
namespace {
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__global__ __launch_bounds__(64) void softmax_fwd_kernel(int F, const float* __restrict__ x, float* __restrict__ y) {
  const float* xr = x + (int64_t)blockIdx.x * F;
  float* yr = y + (int64_t)blockIdx.x * F;
  float m = -INFINITY;
  for (int f = threadIdx.x; f < F; f += 64) m = fmaxf(m, xr[f]);
  m = wave_max(m);
  float s = 0.f;
  for (int f = threadIdx.x; f < F; f += 64) s += expf(xr[f] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int f = threadIdx.x; f < F; f += 64) yr[f] = expf(xr[f] - m) * inv;
}
__global__ __launch_bounds__(64) void softmax_bwd_kernel(int F, const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx) {
  const int64_t base = (int64_t)blockIdx.x * F;
  float s = 0.f;
  for (int f = threadIdx.x; f < F; f += 64) s += dy[base + f] * y[base + f];
  s = wave_sum(s);
  for (int f = threadIdx.x; f < F; f += 64) dx[base + f] = y[base + f] * (dy[base + f] - s);
}
}  // namespace

extern "C" int pcnn_softmax_fwd(pcnn_handle h, int N, int F, const float* x, float* y) {
  PCNN_REQUIRE(h, h && x && y && N >= 1 && F >= 1, "pcnn_softmax_fwd: bad argument");
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(N), dim3(64), 0, h->stream, F, x, y);
  PCNN_CHECK_LAUNCH(h, "pcnn_softmax_fwd");
  return 0;
}

extern "C" int pcnn_softmax_bwd(pcnn_handle h, int N, int F, const float* y, const float* dy, float* dx) {
  PCNN_REQUIRE(h, h && y && dy && dx && N >= 1 && F >= 1, "pcnn_softmax_bwd: bad argument");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(N), dim3(64), 0, h->stream, F, y, dy, dx);
  PCNN_CHECK_LAUNCH(h, "pcnn_softmax_bwd");
  return 0;
}

extern "C" int pcnn_layernorm_fwd(pcnn_handle h, int N, int F, const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean,
                                  float* rstd) {
  PCNN_REQUIRE(h, h && x && gamma && beta && y && mean && rstd && N >= 1 && F >= 1, "pcnn_layernorm_fwd: bad argument");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(N), dim3(256), 0, h->stream, F, x, gamma, beta, eps, y, mean, rstd);
  PCNN_CHECK_LAUNCH(h, "pcnn_layernorm_fwd");
  return 0;
}

extern "C" int pcnn_layernorm_bwd(pcnn_handle h, int N, int F, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy,
                                  float* dx, float* dgamma, float* dbeta) {
  PCNN_REQUIRE(h, h && x && gamma && mean && rstd && dy && dx && dgamma && dbeta && N >= 1 && F >= 1, "pcnn_layernorm_bwd: bad argument");
  hipLaunchKernelGGL(layernorm_bwd_x_kernel, dim3(N), dim3(256), 0, h->stream, F, x, gamma, mean, rstd, dy, dx);
  hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3((F + 255) / 256), dim3(256), 0, h->stream, N, F, x, mean, rstd, dy, dgamma, dbeta);
  PCNN_CHECK_LAUNCH(h, "pcnn_layernorm_bwd");
  return 0;
}
