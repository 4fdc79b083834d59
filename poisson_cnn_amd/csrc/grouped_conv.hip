// Per-sample-filter ("metalearning") convolutions: every sample of the batch is convolved with ITS OWN filter, emitted by a hyper-network
// (layers/metalearning_conv.py:148-169: tf.map_fn over the batch of one tf.nn.conv2d per sample; layers/metalearning_deconvupscale.py:104-137
// the same for conv2d_transpose).  The reference serialises the samples (map_fn); here ONE launch covers the batch: the sample index is a
// grid dimension and the filter / bias pointers advance by a per-sample stride, so the launch count of a metalearning layer does not depend
// on N.  The layers have few channels (<= 32) and filters of any size up to 31 taps (the reference example uses 19 x 19 x 3 x 4), 1-D layers are
// kh = 1: exact fp32 FMAs on the vector ALUs over LDS-staged tiles (the filters of a workgroup's sample are wave-uniform: scalar loads).
//   grouped_fwd_kernel      tf.pad + conv2d(VALID) + bias + activation; with flip = 1 the filter is read flipped and transposed, which
//                           makes the same kernel the data gradient
//   grouped_wgrad_kernel    per-sample filter gradient, deterministic (row strips reduced in a fixed order)
//   grouped_deconv_*        conv2d_transpose with kernel = stride: forward, data gradient, filter + bias gradient
#include "pcnn_internal.h"
#include <algorithm>

namespace {

constexpr int GTH = 8, GTW = 32;          // output tile of the forward kernel: 8 rows x 32 columns, one thread per pixel

struct GroupedParams {
  const float* x; const float* w; const float* bias; float* y;
  long long w_stride, b_stride;           // floats between the filters / biases of consecutive samples
  int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, kh, kw, pt, pl, pad_mode; float pad_value; int act; float alpha;
  int flip;                               // 1: w'[i][j][ci][co] = w[kh-1-i][kw-1-j][co][ci] (the stored filter is (kh, kw, Cout, Cin): data gradient)
  int tiles_x, tiles_y;
};

template <int CO>
__global__ __launch_bounds__(256) void grouped_fwd_kernel(GroupedParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];                  // [(GTH + kh - 1)][(GTW + kw - 1)][4 input channels]
  const int TR = GTH + p.kh - 1, TC = GTW + p.kw - 1;
  const int n = blockIdx.y;
  const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x % p.tiles_x;
  const int y0 = ty * GTH, x0 = tx * GTW;
  const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
  const float* w = p.w + (int64_t)n * p.w_stride;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  for (int ci0 = 0; ci0 < p.Cin; ci0 += 4) {
    for (int u = threadIdx.x; u < TR * TC; u += 256) {
      const int rr = u / TC, cc = u - rr * TC;
      const int sy = pcnn_pad_index(y0 + rr - p.pt, p.H, p.pad_mode), sx = pcnn_pad_index(x0 + cc - p.pl, p.W, p.pad_mode);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = 0.f;
        if (ci0 + q < p.Cin) v[q] = (sy < 0 || sx < 0) ? p.pad_value : xin[((int64_t)sy * p.W + sx) * p.ldx + ci0 + q];
      }
      *reinterpret_cast<f32x4*>(lds + 4 * u) = (f32x4){v[0], v[1], v[2], v[3]};
    }
    __syncthreads();
    for (int i = 0; i < p.kh; ++i)
      for (int j = 0; j < p.kw; ++j) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(lds + 4 * ((r + i) * TC + c + j));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ci = ci0 + q;
          if (ci < p.Cin) {                                                   // uniform
            // filter row of (i, j, ci): wave-uniform addresses (scalar loads)
            const float* wr = p.flip ? w + ((int64_t)((p.kh - 1 - i) * p.kw + (p.kw - 1 - j)) * p.Cout) * p.Cin + ci
                                     : w + ((int64_t)(i * p.kw + j) * p.Cin + ci) * p.Cout;
            const int ws = p.flip ? p.Cin : 1;
#pragma unroll
            for (int o = 0; o < CO; ++o)
              if (o < p.Cout) acc[o] = fmaf(xv[q], wr[o * ws], acc[o]);
          }
        }
      }
    __syncthreads();
  }
  const int oy = y0 + r, ox = x0 + c;
  if (oy < p.Ho && ox < p.Wo) {
    float* yo = p.y + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.ldy;
    const float* b = p.bias ? p.bias + (int64_t)n * p.b_stride : nullptr;
#pragma unroll
    for (int o = 0; o < CO; ++o)
      if (o < p.Cout) yo[o] = pcnn_act(acc[o] + (b ? b[o] : 0.f), p.act, p.alpha);
  }
}

// ---- per-sample filter gradient: dw[n][i][j][ci][co] = sum over output pixels of xpad[n][y + i - pt][x + j - pl][ci] dz[n][y][x][co]
struct GroupedWgradParams {
  const float* x; const float* dz; float* part; float* dw;
  long long dw_stride;
  int N, H, W, Cin, ldx, Ho, Wo, Cout, lddz, kh, kw, pt, pl, pad_mode; float pad_value;
  int S, rows_per_strip, nout;            // row strips per (sample, filter row); nout = kw Cin Cout outputs of a filter row
  int jw, jchunks;                        // filter columns per workgroup (jw Cin Cout <= 4096) and chunks per filter row
};
constexpr int GXC = 128;                  // columns staged at a time
constexpr int GNE = 16;                   // outputs per thread at most (kw Cin Cout <= 4096)

__global__ __launch_bounds__(256) void grouped_wgrad_kernel(GroupedWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];                  // x chunk [(GXC + kw - 1)][Cin] then dz chunk [GXC][Cout]
  float* xs = lds;
  float* zs = lds + (GXC + p.kw - 1) * p.Cin;
  const int s = blockIdx.x, i = blockIdx.y / p.jchunks, jc = blockIdx.y % p.jchunks, n = blockIdx.z;
  const int j0 = jc * p.jw, nloc = min(p.jw, p.kw - j0) * p.Cin * p.Cout;          // this workgroup's filter columns and outputs
  const float* xin = p.x + (int64_t)n * p.H * p.W * p.ldx;
  const float* dzn = p.dz + (int64_t)n * p.Ho * p.Wo * p.lddz;
  float acc[GNE];
  int xoff[GNE], zoff[GNE];
#pragma unroll
  for (int e = 0; e < GNE; ++e) {
    acc[e] = 0.f;
    const int o = threadIdx.x + 256 * e;                                        // (j, ci, co)
    const int co = o % p.Cout, ci = (o / p.Cout) % p.Cin, j = j0 + o / (p.Cout * p.Cin);
    xoff[e] = j * p.Cin + ci; zoff[e] = co;
  }
  const int ya = s * p.rows_per_strip, yb = min(p.Ho, ya + p.rows_per_strip);
  for (int y = ya; y < yb; ++y) {
    const int sy = pcnn_pad_index(y + i - p.pt, p.H, p.pad_mode);
    for (int xa = 0; xa < p.Wo; xa += GXC) {
      const int nx = min(GXC, p.Wo - xa);
      for (int u = threadIdx.x; u < (nx + p.kw - 1) * p.Cin; u += 256) {
        const int px = u / p.Cin, ci = u - px * p.Cin;
        const int sx = pcnn_pad_index(xa + px - p.pl, p.W, p.pad_mode);
        xs[u] = (sy < 0 || sx < 0) ? p.pad_value : xin[((int64_t)sy * p.W + sx) * p.ldx + ci];
      }
      for (int u = threadIdx.x; u < nx * p.Cout; u += 256) {
        const int px = u / p.Cout, co = u - px * p.Cout;
        zs[u] = dzn[((int64_t)y * p.Wo + xa + px) * p.lddz + co];
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < GNE; ++e) {
        if (threadIdx.x + 256 * e < nloc) {
          float a = acc[e];
          for (int px = 0; px < nx; ++px) a = fmaf(xs[px * p.Cin + xoff[e]], zs[px * p.Cout + zoff[e]], a);
          acc[e] = a;
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int e = 0; e < GNE; ++e) {
    const int o = threadIdx.x + 256 * e;
    if (o < nloc) p.part[(((int64_t)n * p.S + s) * p.kh + i) * p.nout + (int64_t)j0 * p.Cin * p.Cout + o] = acc[e];
  }
}

__global__ void grouped_wgrad_reduce_kernel(GroupedWgradParams p) {
  const int64_t per = (int64_t)p.kh * p.nout, total = (int64_t)p.N * per;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / per, e = t - n * per;
    float a = 0.f;
    for (int s = 0; s < p.S; ++s) a += p.part[((n * p.S + s) * p.kh) * p.nout + e];       // fixed order: deterministic
    p.dw[n * p.dw_stride + e] = a;
  }
}

// ---- transposed convolution with kernel = stride f, TensorFlow's padding='SAME' (layers/metalearning_deconvupscale.py:13-16,28-30):
//   y[n][Y][X][co] = b[n][co] + sum_ci x[n][(Y + py) / f][(X + px) / f][ci] K[n][(Y + py) % f][(X + px) % f][co][ci],
//   py = (H f - Ho) / 2, px = (W f - Wo) / 2 - the crop offset of conv2d_transpose when the coarse grid overhangs the output (H = ceil(Ho / f)),
//   the same convention as deconv_mfma.hip and oracle/np_ops.py:conv2d_transpose_same.
struct GroupedDeconvParams {
  const float* x; const float* k; const float* bias; const float* dy; float* y; float* dx; float* dk; float* dbias; float* part;
  long long k_stride, b_stride;
  int N, H, W, Cin, Ho, Wo, Cout, f, py, px, S, rows_per_strip;
};

__global__ __launch_bounds__(256) void grouped_deconv_fwd_kernel(GroupedDeconvParams p) {
  const int n = blockIdx.y;
  const int64_t per = (int64_t)p.Ho * p.Wo * p.Cout;
  const float* K = p.k + (int64_t)n * p.k_stride;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < per; t += (int64_t)gridDim.x * blockDim.x) {
    const int co = t % p.Cout; const int64_t pix = t / p.Cout; const int X = pix % p.Wo, Y = pix / p.Wo;
    const int Yp = Y + p.py, Xp = X + p.px;
    const int yy = Yp / p.f, xx = Xp / p.f;
    float a = p.bias ? p.bias[(int64_t)n * p.b_stride + co] : 0.f;
    if (yy < p.H && xx < p.W) {
      const float* xv = p.x + (((int64_t)n * p.H + yy) * p.W + xx) * p.Cin;
      const float* kv = K + (((int64_t)(Yp - yy * p.f) * p.f + (Xp - xx * p.f)) * p.Cout + co) * p.Cin;
      for (int ci = 0; ci < p.Cin; ++ci) a = fmaf(xv[ci], kv[ci], a);
    }
    p.y[(int64_t)n * per + t] = a;
  }
}

// dx[n][y][x][ci] = sum over the f x f taps and co of dy[n][f y + u - py][f x + v - px][co] K[n][u][v][co][ci] (fine pixels outside the output: none)
__global__ __launch_bounds__(256) void grouped_deconv_bwd_data_kernel(GroupedDeconvParams p) {
  const int n = blockIdx.y;
  const int64_t per = (int64_t)p.H * p.W * p.Cin;
  const float* K = p.k + (int64_t)n * p.k_stride;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < per; t += (int64_t)gridDim.x * blockDim.x) {
    const int ci = t % p.Cin; const int64_t pix = t / p.Cin; const int x = pix % p.W, y = pix / p.W;
    float a = 0.f;
    for (int u = 0; u < p.f; ++u)
      for (int v = 0; v < p.f; ++v) {
        const int Y = p.f * y + u - p.py, X = p.f * x + v - p.px;
        if (Y >= 0 && Y < p.Ho && X >= 0 && X < p.Wo) {
          const float* g = p.dy + (((int64_t)n * p.Ho + Y) * p.Wo + X) * p.Cout;
          const float* kv = K + ((int64_t)(u * p.f + v) * p.Cout) * p.Cin + ci;
          for (int co = 0; co < p.Cout; ++co) a = fmaf(g[co], kv[(int64_t)co * p.Cin], a);
        }
      }
    p.dx[(int64_t)n * per + t] = a;
  }
}

// dK[n][u][v][co][ci] = sum over (y, x) of dy[n][f y + u - py][f x + v - px][co] x[n][y][x][ci]; dbias[n][co] = sum of dy[n][.][.][co].
// Workgroup = (tap, strip of coarse rows, sample), threads over the (co, ci) pairs, a strip's pixels walked in a fixed order; the S strip
// partials (and the strips' bias partials, written by the tap-0 workgroups) are summed in a fixed order by the reduce kernel: deterministic.
// part layout: [n][s][f f Cout Cin | Cout].
__global__ __launch_bounds__(256) void grouped_deconv_bwd_filter_kernel(GroupedDeconvParams p) {
  const int n = blockIdx.z, s = blockIdx.y, tap = blockIdx.x, u = tap / p.f, v = tap % p.f;
  const int npair = p.Cout * p.Cin;
  const int64_t rec = (int64_t)p.f * p.f * npair + p.Cout;
  float* out = p.part + ((int64_t)n * p.S + s) * rec;
  const int y0 = s * p.rows_per_strip, y1 = min(p.H, y0 + p.rows_per_strip);
  for (int e = threadIdx.x; e < npair; e += 256) {
    const int co = e / p.Cin, ci = e - co * p.Cin;
    float a = 0.f;
    for (int y = y0; y < y1; ++y) {
      const int Y = p.f * y + u - p.py;
      if (Y < 0 || Y >= p.Ho) continue;
      for (int x = 0; x < p.W; ++x) {
        const int X = p.f * x + v - p.px;
        if (X < 0 || X >= p.Wo) continue;
        a = fmaf(p.dy[(((int64_t)n * p.Ho + Y) * p.Wo + X) * p.Cout + co], p.x[(((int64_t)n * p.H + y) * p.W + x) * p.Cin + ci], a);
      }
    }
    out[(int64_t)tap * npair + e] = a;
  }
  if (tap == 0) {                                  // bias partial of this strip: its fine rows are f y0 - py .. f y1 - py (a partition of 0..Ho)
    __shared__ float red[256];
    const int Y0 = max(0, p.f * y0 - p.py), Y1 = min(p.Ho, p.f * y1 - p.py);
    const int64_t cnt = (int64_t)max(0, Y1 - Y0) * p.Wo;
    const float* g = p.dy + ((int64_t)n * p.Ho + Y0) * p.Wo * p.Cout;
    for (int co = 0; co < p.Cout; ++co) {
      float a = 0.f;
      for (int64_t t = threadIdx.x; t < cnt; t += 256) a += g[t * p.Cout + co];
      red[threadIdx.x] = a;
      __syncthreads();
      for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
      if (threadIdx.x == 0) out[(int64_t)p.f * p.f * npair + co] = red[0];
      __syncthreads();
    }
  }
}

__global__ void grouped_deconv_filter_reduce_kernel(GroupedDeconvParams p) {
  const int64_t nk = (int64_t)p.f * p.f * p.Cout * p.Cin, rec = nk + p.Cout, total = (int64_t)p.N * rec;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / rec, e = t - n * rec;
    float a = 0.f;
    for (int s = 0; s < p.S; ++s) a += p.part[(n * p.S + s) * rec + e];
    if (e < nk) p.dk[n * p.k_stride + e] = a;
    else if (p.dbias) p.dbias[n * p.b_stride + (e - nk)] = a;
  }
}

}  // namespace

extern "C" size_t pcnn_grouped_conv2d_wgrad_workspace(const pcnn_conv_desc* d) {
  if (!d) return 0;
  const int S = std::max(1, std::min(16, d->Ho / 8));
  return (size_t)d->N * S * d->kh * d->kw * d->Cin * d->Cout * sizeof(float);
}

extern "C" int pcnn_grouped_conv2d_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, long long w_sample_stride, const float* bias,
                                       long long bias_sample_stride, int flip_transpose, float* y) {
  PCNN_REQUIRE(h, h && d && x && w && y, "pcnn_grouped_conv2d_fwd: null argument");
  PCNN_REQUIRE(h, d->Cout >= 1 && d->Cout <= 32 && d->Cin >= 1 && d->kh >= 1 && d->kw >= 1 && d->kh <= 31 && d->kw <= 31,
               "pcnn_grouped_conv2d_fwd: %d -> %d channels, %d x %d taps unsupported (<= 32 output channels, <= 31 taps)", d->Cin, d->Cout, d->kh, d->kw);
  GroupedParams p;
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.w_stride = w_sample_stride; p.b_stride = bias_sample_stride;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.ldy = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value; p.act = d->act; p.alpha = d->act_alpha;
  p.flip = flip_transpose ? 1 : 0;
  p.tiles_x = pcnn_cdiv(d->Wo, GTW); p.tiles_y = pcnn_cdiv(d->Ho, GTH);
  const size_t lds = (size_t)(GTH + d->kh - 1) * (GTW + d->kw - 1) * 4 * sizeof(float);
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)d->N);
  if (d->Cout <= 4) hipLaunchKernelGGL(grouped_fwd_kernel<4>, grid, dim3(256), lds, h->stream, p);
  else if (d->Cout <= 8) hipLaunchKernelGGL(grouped_fwd_kernel<8>, grid, dim3(256), lds, h->stream, p);
  else if (d->Cout <= 16) hipLaunchKernelGGL(grouped_fwd_kernel<16>, grid, dim3(256), lds, h->stream, p);
  else hipLaunchKernelGGL(grouped_fwd_kernel<32>, grid, dim3(256), lds, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_conv2d_fwd");
  return 0;
}

extern "C" int pcnn_grouped_conv2d_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw, long long dw_sample_stride, void* workspace) {
  PCNN_REQUIRE(h, h && d && x && dz && dw && workspace, "pcnn_grouped_conv2d_wgrad: null argument");
  const int nout = d->kw * d->Cin * d->Cout;
  PCNN_REQUIRE(h, d->Cin * d->Cout <= 256 * GNE, "pcnn_grouped_conv2d_wgrad: Cin Cout = %d exceeds %d", d->Cin * d->Cout, 256 * GNE);
  GroupedWgradParams p;
  p.x = x; p.dz = dz; p.part = static_cast<float*>(workspace); p.dw = dw; p.dw_stride = dw_sample_stride;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.lddz = d->ldy;
  p.kh = d->kh; p.kw = d->kw; p.pt = d->pad_top; p.pl = d->pad_left; p.pad_mode = d->pad_mode; p.pad_value = d->pad_value;
  p.S = std::max(1, std::min(16, d->Ho / 8)); p.rows_per_strip = pcnn_cdiv(d->Ho, p.S); p.nout = nout;
  p.jw = std::min(d->kw, (256 * GNE) / (d->Cin * d->Cout)); p.jchunks = pcnn_cdiv(d->kw, p.jw);
  const size_t lds = ((size_t)(GXC + d->kw - 1) * d->Cin + (size_t)GXC * d->Cout) * sizeof(float);
  PCNN_REQUIRE(h, lds <= 64 * 1024, "pcnn_grouped_conv2d_wgrad: %d + %d channels do not fit the staging buffer", d->Cin, d->Cout);
  hipLaunchKernelGGL(grouped_wgrad_kernel, dim3((unsigned)p.S, (unsigned)(d->kh * p.jchunks), (unsigned)d->N), dim3(256), lds, h->stream, p);
  hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64((int64_t)d->N * d->kh * nout, 256), 1024)), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_conv2d_wgrad");
  return 0;
}

static GroupedDeconvParams deconv_params(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, long long ks, long long bs) {
  GroupedDeconvParams p;
  p.x = nullptr; p.k = nullptr; p.bias = nullptr; p.dy = nullptr; p.y = nullptr; p.dx = nullptr; p.dk = nullptr; p.dbias = nullptr;
  p.part = nullptr; p.S = 1; p.rows_per_strip = H;
  p.k_stride = ks; p.b_stride = bs; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.f = f;
  p.py = (H * f - Ho) / 2; p.px = (W * f - Wo) / 2;          // conv2d_transpose 'SAME' crop offset (the geometry is checked by the callers below)
  return p;
}

extern "C" int pcnn_grouped_deconv_fwd(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* x, const float* k,
                                       long long k_sample_stride, const float* bias, long long bias_sample_stride, float* y) {
  PCNN_REQUIRE(h, h && x && k && y && f >= 1, "pcnn_grouped_deconv_fwd: bad argument");
  PCNN_REQUIRE(h, N >= 1 && Cin >= 1 && Cout >= 1 && Ho >= 1 && Wo >= 1 && H == pcnn_cdiv(Ho, f) && W == pcnn_cdiv(Wo, f),
               "%s: padding='SAME' with stride %d needs a coarse grid of ceil(%d / %d) x ceil(%d / %d), got %d x %d", "pcnn_grouped_deconv_fwd", f, Ho, f, Wo, f, H, W);
  GroupedDeconvParams p = deconv_params(N, H, W, Cin, Ho, Wo, Cout, f, k_sample_stride, bias_sample_stride);
  p.x = x; p.k = k; p.bias = bias; p.y = y;
  const int64_t per = (int64_t)Ho * Wo * Cout;
  hipLaunchKernelGGL(grouped_deconv_fwd_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(per, 256), 4096), (unsigned)N), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_deconv_fwd");
  return 0;
}

extern "C" int pcnn_grouped_deconv_bwd_data(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* dy, const float* k,
                                            long long k_sample_stride, float* dx) {
  PCNN_REQUIRE(h, h && dy && k && dx && f >= 1, "pcnn_grouped_deconv_bwd_data: bad argument");
  PCNN_REQUIRE(h, N >= 1 && Cin >= 1 && Cout >= 1 && Ho >= 1 && Wo >= 1 && H == pcnn_cdiv(Ho, f) && W == pcnn_cdiv(Wo, f),
               "%s: padding='SAME' with stride %d needs a coarse grid of ceil(%d / %d) x ceil(%d / %d), got %d x %d", "pcnn_grouped_deconv_bwd_data", f, Ho, f, Wo, f, H, W);
  GroupedDeconvParams p = deconv_params(N, H, W, Cin, Ho, Wo, Cout, f, k_sample_stride, 0);
  p.dy = dy; p.k = k; p.dx = dx;
  const int64_t per = (int64_t)H * W * Cin;
  hipLaunchKernelGGL(grouped_deconv_bwd_data_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64(per, 256), 4096), (unsigned)N), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_deconv_bwd_data");
  return 0;
}

extern "C" int pcnn_grouped_deconv_bwd_filter(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* x, const float* dy,
                                              float* dk, long long k_sample_stride, float* dbias, long long bias_sample_stride) {
  PCNN_REQUIRE(h, h && x && dy && dk && f >= 1, "pcnn_grouped_deconv_bwd_filter: bad argument");
  PCNN_REQUIRE(h, N >= 1 && Cin >= 1 && Cout >= 1 && Ho >= 1 && Wo >= 1 && H == pcnn_cdiv(Ho, f) && W == pcnn_cdiv(Wo, f),
               "%s: padding='SAME' with stride %d needs a coarse grid of ceil(%d / %d) x ceil(%d / %d), got %d x %d", "pcnn_grouped_deconv_bwd_filter", f, Ho, f, Wo, f, H, W);
  GroupedDeconvParams p = deconv_params(N, H, W, Cin, Ho, Wo, Cout, f, k_sample_stride, bias_sample_stride);
  p.x = x; p.dy = dy; p.dk = dk; p.dbias = dbias;
  p.S = std::max(1, std::min(32, H / 2)); p.rows_per_strip = pcnn_cdiv(H, p.S); p.S = pcnn_cdiv(H, p.rows_per_strip);
  const int64_t rec = (int64_t)f * f * Cout * Cin + Cout;
  const size_t need = (size_t)N * p.S * rec * sizeof(float);
  if (h->aux_ws_bytes < need) {                              // handle-owned scratch (shared with the two-pass resize; one stream per handle)
    if (h->aux_ws) { pcnn_release(h, h->aux_ws); h->aux_ws = nullptr; h->aux_ws_bytes = 0; }
    if (hipMalloc(&h->aux_ws, need) != hipSuccess) PCNN_FAIL(h, "pcnn_grouped_deconv_bwd_filter: cannot allocate %zu B of scratch", need);
    h->aux_ws_bytes = need;
  }
  p.part = static_cast<float*>(h->aux_ws);
  hipLaunchKernelGGL(grouped_deconv_bwd_filter_kernel, dim3((unsigned)(f * f), (unsigned)p.S, (unsigned)N), dim3(256), 0, h->stream, p);
  hipLaunchKernelGGL(grouped_deconv_filter_reduce_kernel, dim3((unsigned)std::min<int64_t>(pcnn_cdiv64((int64_t)N * rec, 256), 1024)), dim3(256), 0, h->stream, p);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_deconv_bwd_filter");
  return 0;
}

// dbias[n][co] = sum over the pixels of sample n of dz[n][.][co] (one workgroup per sample, fixed summation order)
namespace {
__global__ __launch_bounds__(256) void grouped_bias_grad_kernel(const float* __restrict__ dz, int64_t HW, int C, int lddz, float* __restrict__ dbias, long long b_stride) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  const float* g = dz + (int64_t)n * HW * lddz;
  for (int co = 0; co < C; ++co) {
    float a = 0.f;
    for (int64_t t = threadIdx.x; t < HW; t += 256) a += g[t * lddz + co];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) dbias[(int64_t)n * b_stride + co] = red[0];
    __syncthreads();
  }
}
}  // namespace

extern "C" int pcnn_grouped_bias_grad(pcnn_handle h, int N, long long HW, int C, const float* dz, int lddz, float* dbias, long long bias_sample_stride) {
  PCNN_REQUIRE(h, h && dz && dbias && N >= 1 && C >= 1, "pcnn_grouped_bias_grad: bad argument");
  hipLaunchKernelGGL(grouped_bias_grad_kernel, dim3((unsigned)N), dim3(256), 0, h->stream, dz, (int64_t)HW, C, lddz, dbias, bias_sample_stride);
  PCNN_CHECK_LAUNCH(h, "pcnn_grouped_bias_grad");
  return 0;
}
