/* libpcnn - MI355X (gfx950) native kernels for the poisson_CNN hot path.  C-ABI drop-in boundary.
 *
 * The reference (aligirayhanozbay/poisson_CNN) has no FFI boundary of its own: every FLOP of its hot path runs
 * inside TensorFlow ops called from Python.  Each entry point below therefore names the TensorFlow op (and the
 * reference call site, path:line relative to poisson_CNN/) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - All tensors are float32, device memory, NHWC ("channels innermost").  A tensor argument is a base pointer
 *    plus an explicit channel stride `ld*` (floats between consecutive pixels), so a kernel can read/write a
 *    channel slice of a wider buffer (this is how tf.concat(axis=1), models/Homogeneous_Poisson_NN_Legacy.py:224,
 *    is realised without a copy).
 *  - The caller owns every tensor and every workspace that appears as an argument.  The handle owns three scratch regions it grows on demand
 *    (hipMalloc on first use, freed by pcnn_destroy): the packed-filter scratch of the direct convolutions (KBs), the x-interpolated rows of the
 *    two-pass resize, and the SPECTRAL WORKSPACE of the tiled spectral convolutions - tables, the filter spectrum and the tile spectra of the
 *    launch in flight, up to ~14 GB for an 8 x 1024^2 x 32-channel layer.  pcnn_set_workspace_limit caps the latter (the route then runs in
 *    smaller tile chunks); growing it synchronises the stream, so a training loop reaches its high-water mark in the first step.
 *  - Every call is asynchronous on the handle's stream and returns 0 on success, non-zero on error
 *    (pcnn_last_error(handle) gives the message).  No exceptions cross the boundary.
 *  - One handle per (thread, device, stream).
 *  - Environment variables read by the library are DEVELOPER switches for A/B timing and tests, not part of the interface; each is validated
 *    where it is read (an invalid value keeps the default): PCNN_MATH, PCNN_SPECTRAL, PCNN_SPEC_T (same as the pcnn_set_* calls),
 *    PCNN_SPEC_XFORM (transform kernels of the spectral route: "fft" in-register vector-ALU FFT, "mfma" DFT-as-GEMM), PCNN_WSPLIT32 / PCNN_WSPLIT64
 *    (weight-gradient partial sums: a positive multiple of 4, at most 64), PCNN_SPEC_MIXW, PCNN_SPEC_PACK, PCNN_SPEC_CHUNK, PCNN_SPEC_FENCE,
 *    PCNN_FWD64_RADIX, PCNN_STAGE_TH, PCNN_FUSED_STAGE, PCNN_SMALL_CONV, PCNN_GROUPED_VALU, PCNN_WG_OCC3, PCNN_SPLIT_MT2_MAXK (0 / 1 or a
 *    size), PCNN_RCCL_LIBRARY / PCNN_ROCFFT_LIBRARY (a path for dlopen).
 */
#ifndef PCNN_H
#define PCNN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcnn_handle_s* pcnn_handle;

enum { PCNN_PAD_CONSTANT = 0, PCNN_PAD_SYMMETRIC = 1, PCNN_PAD_REFLECT = 2 };   /* tf.pad modes */
enum { PCNN_ACT_LINEAR = 0, PCNN_ACT_LEAKY_RELU = 1, PCNN_ACT_TANH = 2, PCNN_ACT_RELU = 3 };
enum { PCNN_POOL_AVERAGE = 0, PCNN_POOL_MAX = 1 };
enum { PCNN_RESIZE_NEAREST = 0, PCNN_RESIZE_BILINEAR = 1, PCNN_RESIZE_BICUBIC = 2,
       PCNN_RESIZE_BICUBIC_LEGACY_ALIGN_CORNERS = 3 /* tf.compat.v1 resize_images(align_corners=True), dataset/utils/image_resize.py:20 */ };

int pcnn_create(int device, void* hip_stream, pcnn_handle* out);
int pcnn_destroy(pcnn_handle h);
int pcnn_set_stream(pcnn_handle h, void* hip_stream);
int pcnn_sync(pcnn_handle h);
const char* pcnn_last_error(pcnn_handle h);
int pcnn_version(void);
/* CRC-32C of a host buffer (continue from `crc`, 0 to start): the checksum of TensorFlow TensorBundle checkpoints
 * (train/utils.py:10-29 loads them; poisson_cnn_amd/tf_checkpoint.py reads and writes the format).  Host only. */
uint32_t pcnn_crc32c(const void* data, size_t n, uint32_t crc);

/* Data-parallel collective: what tf.distribute.MirroredStrategy does under `with dist_strategy.scope()`
 * (train/hpnn_legacy_train.py:37) - ONE sum all-reduce of the flat gradient bucket per optimizer step - over RCCL / xGMI, on the handle's
 * stream.  RCCL is loaded at the first call (dlopen); rendezvous is the caller's: rank 0 obtains the 128-byte id and ships it to the
 * other ranks over whatever channel the host program has, then every rank (one process per GPU, one handle) calls pcnn_comm_init.
 * pcnn_allreduce / pcnn_broadcast are in place, asynchronous, fp32. */
#define PCNN_UNIQUE_ID_BYTES 128
/* host-only probe (no handle, no GPU): 0 when RCCL can be bound, else 1 with the loader's message in `why`.  The environment variable
 * PCNN_RCCL_LIBRARY names the one library file to try instead of the default search (librccl.so.1, librccl.so, /opt/rocm/lib). */
int pcnn_collective_available(char* why, size_t why_bytes);
int pcnn_comm_unique_id(pcnn_handle h, void* id_out);
int pcnn_comm_init(pcnn_handle h, const void* id, int rank, int world_size);
int pcnn_comm_destroy(pcnn_handle h);
int pcnn_allreduce(pcnn_handle h, float* buf, size_t count);
int pcnn_broadcast(pcnn_handle h, float* buf, size_t count, int root);

/* Arithmetic of the convolution GEMMs.
 *   PCNN_MATH_FP32      (default): v_mfma_f32_32x32x2_f32, exact fp32 products, fp32 accumulate.
 *   PCNN_MATH_SPLIT_F16 : every fp32 operand is split into two fp16 halves (hi + lo, 22 significant bits, per-tile power-of-two
 *                         scaling) and a*b is accumulated in fp32 as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16: 5.3x fewer
 *                         matrix-pipe cycles at an accuracy that is measured to be BETTER than the fp32 fmaf chain
 *                         (tools/split_f16_test.hip: 5.4e-7 vs 1.5e-6 rel-L2 against fp64 at K = 7200). */
enum { PCNN_MATH_FP32 = 0, PCNN_MATH_SPLIT_F16 = 1 };
int pcnn_set_math_mode(pcnn_handle h, int mode);
int pcnn_get_math_mode(pcnn_handle h);

/* Algorithm of pcnn_conv2d_fwd / pcnn_conv2d_wgrad for wide filters (replaces the same tf.nn.conv2d / backprop-filter calls):
 *   tiled spectral convolution - overlap-save on 32 x 32 tiles (64 x 64 for 11..15 taps: csrc/spectral64.hip), the DFT applied as fp32 MFMA matrix products, per-frequency channel
 *   mixing, inverse transform fused with the conv epilogue (csrc/spectral_conv.hip).  Exact fp32 products and accumulation like the
 *   direct kernel, ~k*k/25 times fewer of them.  PCNN_SPECTRAL_AUTO (default; environment PCNN_SPECTRAL=-1|0|1): a cost model picks the
 *   route per layer and math mode; _OFF: always the direct implicit GEMM; _FORCE: spectral whenever the shape allows
 *   (kh, kw <= 15, Cin <= 64, Cout <= 32).  The handle owns the workspace (tile spectra), grown on demand. */
enum { PCNN_SPECTRAL_AUTO = -1, PCNN_SPECTRAL_OFF = 0, PCNN_SPECTRAL_FORCE = 1 };
int pcnn_set_spectral_mode(pcnn_handle h, int mode);
int pcnn_get_spectral_mode(pcnn_handle h);
/* Tile size of the spectral route: 0 (default; environment PCNN_SPEC_T) = per layer - 64 x 64 tiles for 13..15 taps on images of >= 36 such
 * tiles and for 11..12 taps on images of >= 256 such tiles, 32 x 32 otherwise (decided on ONE image: a sample's arithmetic never depends on
 * its batch neighbours); 32 / 64 = that size wherever the layer allows it (64: 9..15 taps, more than 16 channels on one side). */
int pcnn_set_spectral_tile(pcnn_handle h, int tile);
int pcnn_get_spectral_tile(pcnn_handle h);
/* Transform kernels of the spectral route (same spectrum layout, same arithmetic class - exact fp32 -, results agree to ~1e-7):
 *   PCNN_XFORM_MFMA: the DFT as a GEMM on the matrix cores (v_mfma_f32_32x32x2_f32; csrc/spectral_conv.hip, spectral64.hip);
 *   PCNN_XFORM_FFT:  in-register FFTs on the vector ALUs, lane = channel (csrc/spectral_fft.hip, fft_regs.h): fp32 MFMA has no rate advantage
 *                    over the vector ALUs on gfx950, so the FFT needs ~9x fewer issue cycles, half the registers and twice the waves per CU.
 * Default: PCNN_XFORM_FFT (measured 20-30 % faster per layer at 8 x 1024^2, profiles/r05_probe_xform*.txt); environment PCNN_SPEC_XFORM=fft|mfma sets the
 * default of new handles. */
enum { PCNN_XFORM_MFMA = 0, PCNN_XFORM_FFT = 1 };
int pcnn_set_spectral_transform(pcnn_handle h, int xform);
int pcnn_get_spectral_transform(pcnn_handle h);
/* Filter spectra across calls (round 5).  A filter's spectrum depends on the weights alone; by default (version 0) every spectral convolution call
 * recomputes it into the workspace - correct for any caller, ~10 us + a launch gap per call.  A caller that knows WHEN its weights change tells the
 * handle: pcnn_set_filter_version(h, v) with v != 0 promises that every filter pointer passed to pcnn_conv2d_fwd / pcnn_conv2d_bwd_spectral(_post)
 * while the version is v holds the same values as at the first call under v.  The handle then keeps each filter's spectrum (key: pointer, shape, tile
 * size, transform family) in a buffer of its own; a call under the same version reuses it, and the first call under a NEW version refreshes every
 * filter the handle knows in ONE launch per tile size.  Inference: set a version once - no filter work after the first call.  Training: bump the
 * version after each optimizer step - one or two launches per step instead of one per layer and direction.  The setting is per call sequence: set 0
 * again before passing filters whose contents the version does not describe (poisson_cnn_amd.ops does that for every call without `w_version`).
 * Contract: a cached filter pointer must stay readable until pcnn_filter_cache_clear / pcnn_destroy (the refresh reads every known filter); memory:
 * Cin x ceil(Cout/32) x T^2 x 128 B per filter (4 MB at 32 -> 32 channels and 32-point tiles, 16 MB at 64-point tiles), pcnn_filter_cache_stats.
 * Under stream capture nothing is allocated: a filter first seen there is transformed into the workspace as with version 0.
 * pcnn_filter_cache_stats: entries / bytes describe what the handle holds now; hits / fills / refreshes are cumulative over the handle's life (a clear does not reset them). */
int pcnn_set_filter_version(pcnn_handle h, uint64_t version);
int pcnn_filter_cache_clear(pcnn_handle h);
int pcnn_filter_cache_stats(pcnn_handle h, long long* entries, long long* bytes, long long* hits, long long* fills, long long* refreshes);
/* The one LARGE buffer a handle owns (besides small scratch and, when a filter version is set, the kept filter spectra above): the spectral workspace (tile spectra of the layer in flight, mixing matrices,
 * constant tables).  By default it grows to a whole layer per launch - 8 x 1024^2 x 32 channels at 15 taps: ~11 GB, sized for 288 GB of
 * HBM.  pcnn_set_workspace_limit caps it (bytes; 0 = no cap): the launches then cover fewer tiles each (never fewer than 32; a layer that
 * cannot run inside the cap fails with an ordinary error), so a host program that budgets device memory itself decides what the
 * library may take.  Growth happens inside a convolution call (stream synchronise + free + allocate) until the largest layer shape has
 * been seen: call the largest shape once up front, or set the cap, to keep allocation out of the steady state. */
int pcnn_set_workspace_limit(pcnn_handle h, size_t bytes);
/* hipGraph safety (poisson_cnn_amd/graphs.py; no reference counterpart - TensorFlow's tf.function graphs own their scratch): a graph
 * captured on this handle's stream has the handle-owned buffers' addresses baked in.  retain = 1: a buffer the handle outgrows (or that a
 * lower pcnn_set_workspace_limit releases) is NOT freed but parked until pcnn_destroy, so that earlier captures keep replaying into valid
 * memory; later calls use the new, larger buffer.  retain = 0 (default): outgrown buffers are freed after a stream synchronise. */
int pcnn_set_workspace_retain(pcnn_handle h, int retain);

/* ---- 2-D convolution: tf.pad + tf.nn.conv2d(VALID) + bias + activation (+ BN affine) (+ residual) ----------
 * Replaces pad_and_apply_convolution (utils/apply_advanced_padding_and_call_conv_layer.py:16-20), Keras
 * Conv2D(padding='same') (models/Homogeneous_Poisson_NN_Legacy.py:71,75,95; layers/Scaling.py:28), the fused
 * BatchNormalization that follows it (blocks/resnet.py:31-36) and the residual add of blocks/resnet.py:37.
 *   z[n,y,x,co]   = bias[co] + sum_{i,j,ci} xpad[n, y - pad_top + i, x - pad_left + j, ci] * w[i,j,ci,co]
 *   a             = act(z);   y = a * bn_scale[co] + bn_shift[co] (if bn_scale);   y += residual (if residual)
 * w is Keras HWIO (kh,kw,Cin,Cout), dense.  Out-of-image reads follow pad_mode (CONSTANT uses pad_value).
 * Ho/Wo need not equal H/W (the data-gradient call uses Ho = H + kh - 1).
 * If act_out != NULL the pre-BN activation `a` is also written there (needed by the backward pass). */
typedef struct {
  int N, H, W, Cin, ldx;          /* input  */
  int Ho, Wo, Cout, ldy;          /* output */
  int kh, kw, pad_top, pad_left;
  int pad_mode; float pad_value;
  int act; float act_alpha;
  int ld_res;                     /* channel stride of residual (if given) */
  int ld_act_out;                 /* channel stride of act_out  (if given) */
} pcnn_conv_desc;

int pcnn_conv2d_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias,
                    const float* bn_scale, const float* bn_shift, const float* residual, float* y, float* act_out);
/* Same, and y_absmax[0] = max|y| over the tensor (device float, NULL = skip): when y is the input of the next convolution, that
 * layer's weight gradient takes it as its x_absmax hint (pcnn_conv2d_wgrad_hint) */
int pcnn_conv2d_fwd_absmax(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, const float* bias,
                           const float* bn_scale, const float* bn_shift, const float* residual, float* y, float* act_out, float* y_absmax);

/* w (kh,kw,Cin,Cout) -> wt (kh,kw,Cout,Cin) with both taps reversed: the filter of the data-gradient convolution. */
int pcnn_conv2d_flip_transpose_weights(pcnn_handle h, const float* w, float* wt, int kh, int kw, int Cin, int Cout);
/* The same for n filters in ONE launch (a model's ~100 flipped filters are all re-formed when the weights change): `table_dev` is an array of n items in
 * DEVICE memory, ordered by `start` = the running sum of the previous items' element counts (start of item 0 = 0), total = the sum over all items. */
typedef struct pcnn_flip_item { const float* w; float* wt; int kh, kw, Cin, Cout; int64_t start; } pcnn_flip_item;
int pcnn_conv2d_flip_transpose_table(pcnn_handle h, const pcnn_flip_item* table_dev, int n, int64_t total);

/* Filter gradient of the convolution above (tf.nn.conv2d backprop-filter):
 *   dw[i,j,ci,co] = sum_{n,y,x} xpad[n, y - pad_top + i, x - pad_left + j, ci] * dz[n,y,x,co]
 * d describes the FORWARD convolution (x is its input, dz the gradient at its pre-activation output, ldy = its
 * channel stride).  workspace must hold pcnn_conv2d_wgrad_workspace(d) bytes.  The result is deterministic. */
size_t pcnn_conv2d_wgrad_workspace(const pcnn_conv_desc* d);
int pcnn_conv2d_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw,
                      void* workspace, size_t workspace_bytes);
/* Same with optional hints: x_absmax / dz_absmax point to device floats holding max|x| / max|dz| (exact, or any finite upper bound
 * within a factor 2^10: the scale only has to keep the fp16 halves in range), NULL = compute here.  Ignored in the fp32 math mode. */
int pcnn_conv2d_wgrad_hint(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw,
                           void* workspace, size_t workspace_bytes, const float* x_absmax, const float* dz_absmax);

/* Both gradients of a wide-filter layer in one call on the spectral route (csrc/spectral_conv.hip): the spectrum of dz's tile windows is
 * computed once and serves the data gradient (d/dx of the same tf.nn.conv2d) AND the filter gradient, which is formed input-partitioned.
 * d: the FORWARD convolution; dg: its data-gradient convolution (input dz with channel stride dg->ldx, filter w_flipped = the output of
 * pcnn_conv2d_flip_transpose_weights, output dx (N,dg->Ho,dg->Wo,Cin) with stride dg->ldy - for SYMMETRIC / REFLECT layers the gradient on
 * the padded domain, Ho = H + kh - 1; residual (stride dg->ld_res) is added to dx if given).  Eligibility (cost model, Cout <= 32, zero
 * constant padding) must be asked first; an ineligible layer uses pcnn_conv2d_wgrad + pcnn_conv2d_fwd as before. */
/* One narrow resnet stage as ONE launch (csrc/conv_small.hip, round 4): blocks/resnet.py:29-39 with 3 x 3 filters, C = 4 / 8 channels on both
 * sides of all three convolutions, zero CONSTANT padding, no BatchNormalization, activation linear / relu / leaky relu:
 *   o0 = act(conv0(x) + b0);  a1 = act(conv1(o0) + b1);  o1 = x + a1;  y = act(conv2(o1) + b2)
 * x, y and the optional outputs are dense NHWC (N, H, W, C), 16-byte aligned; filters (3, 3, C, C); biases may be NULL.  o0 / a1 / o1 (each may be
 * NULL) receive what the three pcnn_conv2d_fwd calls of the unfused chain leave behind for the backward pass (conv1's input, conv1's act_out,
 * conv2's input): with them the existing backward calls run unchanged.  Same fp32 FMA chain per output value as the narrow forward kernel. */
int pcnn_resnet3_fwd_eligible(pcnn_handle h, int C, int act);
int pcnn_resnet3_fwd(pcnn_handle h, int N, int H, int W, int C, int act, float act_alpha, const float* x, const float* w0, const float* b0,
                     const float* w1, const float* b1, const float* w2, const float* b2, float* o0, float* a1, float* o1, float* y);

int pcnn_conv2d_bwd_spectral_eligible(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg);
int pcnn_conv2d_bwd_spectral(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg, const float* x, const float* dz, const float* w_flipped,
                             const float* residual, float* dx, float* dw);
/* The same call with the activation backward of the PRODUCING layer fused into the data gradient's epilogue (round 4): in a chain
 * a_prev = act(conv_prev(...)); y = conv(a_prev) the gradient dx that this call produces is at once multiplied by act'(a_prev) - which is what
 * pcnn_conv2d_epilogue_bwd would do in a separate pass over the tensor before conv_prev's own backward (blocks/resnet.py:29-39 chains three
 * such layers; models/Homogeneous_Poisson_NN_Legacy.py:86-96 seven such blocks):
 *   v = dgrad(dz) (+ residual);  raw_out (optional, stride ld_raw) = v;  dx = v * act'(act_out);  dbias (optional, Cin floats) = sum over pixels of dx.
 * act_out (stride ld_act_out) is conv_prev's saved activation output, act / act_alpha its activation.  Available where the data gradient has
 * one channel group (Cin <= 32; both the 32-point and - since the second session of round 4 - the 64-point inverse kernel carry the epilogue; ask
 * _post_eligible); otherwise call pcnn_conv2d_bwd_spectral and pcnn_conv2d_epilogue_bwd. */
typedef struct pcnn_post_desc {
  const float* act_out; int ld_act_out; int act; float act_alpha;
  float* dbias; float* raw_out; int ld_raw;
} pcnn_post_desc;
int pcnn_conv2d_bwd_spectral_post_eligible(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg);
int pcnn_conv2d_bwd_spectral_post(pcnn_handle h, const pcnn_conv_desc* d, const pcnn_conv_desc* dg, const float* x, const float* dz, const float* w_flipped,
                                  const float* residual, float* dx, float* dw, const pcnn_post_desc* post);

/* The same fusion for the NARROW layers (<= 16 channels, 3x3 / 5x5: the tail of the final stack, models/Homogeneous_Poisson_NN_Legacy.py:86-96), whose data
 * gradient runs on the vector-ALU kernel (csrc/conv_small.hip) and not through pcnn_conv2d_bwd_spectral: dx = conv(dz; w_flipped) with the data-gradient
 * descriptor dg (as one would pass it to pcnn_conv2d_fwd: zero padding kh - 1 - pad_top / kw - 1 - pad_left, linear epilogue) + residual, then `post` exactly as
 * above - raw_out receives the sum, dx receives it times act'(act_out), dbias the per-channel sums of dx (partial sums per workgroup in the handle's scratch,
 * reduced in a fixed order: deterministic).  dx and raw_out are bit-identical to pcnn_conv2d_fwd followed by pcnn_conv2d_epilogue_bwd; dbias differs from that
 * pair in summation order only.  Replaces the tape's tf.nn.conv2d_backprop_input + the activation gradient of the producing layer (tf.GradientTape,
 * models/Homogeneous_Poisson_NN_Legacy.py:265-270).  Eligible: the narrow route for dg, a channel count of dx that is a multiple of 4, 16-byte aligned
 * tensors with channel strides that are multiples of 4; otherwise call pcnn_conv2d_fwd and pcnn_conv2d_epilogue_bwd. */
int pcnn_conv2d_dgrad_post_eligible(pcnn_handle h, const pcnn_conv_desc* dg, const float* dz, const float* residual, const float* dx, const pcnn_post_desc* post);
int pcnn_conv2d_dgrad_post(pcnn_handle h, const pcnn_conv_desc* dg, const float* dz, const float* w_flipped, const float* residual, float* dx,
                           const pcnn_post_desc* post);

/* Diagnostics (tests / tools only; no reference counterpart): the tile spectra of one image exactly as the forward transform of a spectral
 * convolution writes them, with the transform kernels the handle currently selects - out: [tile groups][channel groups][T*T rows][32] floats
 * (layout: csrc/spectral_common.h).  tests/test_gpu_spectral_fft.py and test_gpu_spectral64.py compare the kernel families row by row. */
int pcnn_debug_forward_spectrum32(pcnn_handle h, int H, int W, int C, const float* x, int Vy, int Vx, int oy, int ox, int pad_mode, float pad_value,
                                  int ylim, int xlim, int pack, float* out, size_t out_floats);
int pcnn_debug_tile_spectrum64(pcnn_handle h, int H, int W, int C, const float* x, int ylim, int xlim, float* out);

/* Backward of the fused epilogue: given dy (gradient at y) and the saved activation a = act(z),
 *   dz = dy * bn_scale[c] * act'(z)      (act' recovered from a: leaky -> a>0 ? 1 : alpha, tanh -> 1-a^2)
 * and the per-channel sums  dbias[c] = sum dz,  dgamma_hat[c] = sum dy*a,  dbeta[c] = sum dy  (the caller turns
 * dgamma_hat/dbeta into BN gamma/beta gradients).  Any of dbias/dsum_dy_a/dsum_dy may be NULL.
 * npix = N*H*W.  workspace: pcnn_colsum_workspace(C) bytes. */
size_t pcnn_colsum_workspace(int C);
int pcnn_conv2d_epilogue_bwd(pcnn_handle h, int64_t npix, int C, const float* dy, int lddy, const float* a, int lda,
                             const float* bn_scale, int act, float act_alpha, float* dz, int lddz,
                             float* dbias, float* dsum_dy_a, float* dsum_dy, void* workspace, size_t workspace_bytes);
/* Same, and additionally dz_absmax[0] = max|dz| over the tensor (device float, NULL = skip): the weight-gradient kernels of the
 * split math mode scale dz by this maximum - computing it here saves them a pass over dz (pcnn_conv2d_wgrad_hint) */
int pcnn_conv2d_epilogue_bwd_absmax(pcnn_handle h, int64_t npix, int C, const float* dy, int lddy, const float* a, int lda,
                                    const float* bn_scale, int act, float act_alpha, float* dz, int lddz, float* dbias,
                                    float* dsum_dy_a, float* dsum_dy, float* dz_absmax, void* workspace, size_t workspace_bytes);

/* Adjoint of tf.pad SYMMETRIC / REFLECT / CONSTANT: folds the gradient on the padded domain
 * gp (N, H+pt+pb, W+pl+pr, C) back onto the un-padded image: gx[n,y,x,c] = sum of gp over all padded positions
 * that read (y,x).  accumulate != 0 adds into gx. */
int pcnn_pad_fold_bwd(pcnn_handle h, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int pad_mode,
                      const float* gp, int ldgp, float* gx, int ldgx, int accumulate);
/* The same fold AND the activation backward of the layer that produced the convolution's input, in one pass (round 5; the counterpart of
 * pcnn_conv2d_bwd_spectral_post for SYMMETRIC / REFLECT-padded layers, whose data gradient arrives on the padded domain): g = fold(gp) [+ add_to],
 * post->raw_out = g (optional), dz = g * bn_scale * act'(post->act_out), post->dbias[c] = sum dz, dsum_dy_a[c] = sum g * act_out, dsum_dy[c] = sum g
 * (each optional; bn_scale NULL = 1: the arguments of pcnn_conv2d_epilogue_bwd, whose arithmetic this is operation for operation; fixed summation
 * order).  dx itself never reaches memory.  Eligibility (channels and every channel stride a multiple of 4, 16-byte aligned tensors, C <= 256) must be asked first; workspace:
 * pcnn_colsum_workspace(C) bytes. */
int pcnn_pad_fold_bwd_post_eligible(int C, int ldgp, int ld_add, const void* gp, const void* add_to, const pcnn_post_desc* post, int lddz, const void* dz);
int pcnn_pad_fold_bwd_post(pcnn_handle h, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int pad_mode, const float* gp, int ldgp,
                           const float* add_to, int ld_add, const pcnn_post_desc* post, const float* bn_scale, float* dsum_dy_a, float* dsum_dy,
                           float* dz, int lddz, void* workspace, size_t workspace_bytes);

/* Inference-mode BatchNormalization(axis=1, epsilon) folded to a per-channel affine (blocks/resnet.py:26-27,
 * models/Homogeneous_Poisson_NN_Legacy.py:55):  scale = gamma / sqrt(var + eps), shift = beta - mean * scale, for n channels
 * (all BN layers of a model are processed in one call).  bwd: given S1 = sum dy*a and S2 = sum dy from
 * pcnn_conv2d_epilogue_bwd:  dgamma = (S1 - mean*S2) / sqrt(var+eps),  dbeta = S2. */
int pcnn_bn_fold(pcnn_handle h, int n, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                 float* scale, float* shift);
int pcnn_bn_fold_bwd(pcnn_handle h, int n, const float* s_dy_a, const float* s_dy, const float* mean, const float* var, float eps,
                     float* dgamma, float* dbeta);

/* Training-mode BatchNormalization (fused semantics; optional - the reference's train_step most likely runs BN in inference
 * mode).  Forward: the conv writes the activation a; sum_a / sum_a2 come from pcnn_conv2d_epilogue_bwd(dy=a, a=a,
 * act=linear) (its s_dy and s_dy_a sums); pcnn_bn_train_finalize turns them into batch mean / 1/sqrt(var_biased+eps) / scale /
 * shift and updates the moving statistics (momentum; unbiased variance); pcnn_channel_affine applies y = a*scale + shift (+residual).
 * Backward: given S1 = sum dy*a, S2 = sum dy:  dgamma = inv_std*(S1 - mean*S2), dbeta = S2,
 *   da = scale * (dy - S2/n - xhat * dgamma/n),  xhat = (a - mean)*inv_std.   scratch_2C: 2*C floats. */
int pcnn_channel_affine(pcnn_handle h, int64_t npix, int C, const float* x, int ldx, const float* scale, const float* shift,
                        const float* residual, int ld_res, float* y, int ldy);
int pcnn_bn_train_finalize(pcnn_handle h, int C, int64_t npix, const float* sum_a, const float* sum_a2, const float* gamma, const float* beta,
                           float eps, float momentum, float* moving_mean, float* moving_var, float* mean, float* inv_std, float* scale,
                           float* shift);
int pcnn_bn_train_bwd(pcnn_handle h, int64_t npix, int C, const float* dy, int lddy, const float* a, int lda, const float* scale,
                      const float* mean, const float* inv_std, const float* s_dy_a, const float* s_dy, float* dgamma, float* dbeta,
                      float* scratch_2C, float* da, int ldda);

/* ---- pooling: tf.keras.layers.{Average,Max}Pooling2D(pool_size=f, strides=f, padding='same') ----------------
 * (utils/get_pooling_method.py:3-6; blocks/bottleneck_block.py:36-37; layers/Scaling.py:29).
 * Ho = ceil(H/f); window origin o*f - (Ho*f-H)/2; the average divides by the number of valid elements. */
int pcnn_pool2d_fwd(pcnn_handle h, int kind, int N, int H, int W, int C, int f, const float* x, int ldx, float* y, int ldy);
int pcnn_pool2d_bwd(pcnn_handle h, int kind, int N, int H, int W, int C, int f, const float* x, int ldx,
                    const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx, int accumulate);

/* ---- transposed convolution with kernel == stride (layers/deconvupscale.py:103-108) ---------------------------
 * tf.nn.conv2d_transpose(x, k, output_shape=(N,Cout,H,W), strides=f, padding='SAME') + bias, k is (f,f,Cout,Cin):
 *   y[n,Y,X,co] = bias[co] + sum_ci x[n,(Y+py)/f,(X+px)/f,ci] * k[(Y+py)%f,(X+px)%f,co,ci],  py=(h*f-H)/2.
 * y = beta*y + alpha*(...) so the 8-way branch merge (models/Homogeneous_Poisson_NN_Legacy.py:222) accumulates in place. */
int pcnn_deconv_fwd(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx,
                    const float* k, const float* bias, float alpha, float beta, float* y, int ldy);
int pcnn_deconv_bwd_data(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* dy, int lddy,
                         const float* k, float alpha, float* dx, int lddx);
/* dk (f,f,Cout,Cin) and dbias (Cout) ; workspace: pcnn_deconv_wgrad_workspace bytes */
size_t pcnn_deconv_wgrad_workspace(int N, int hc, int wc, int Cin, int Cout, int f);
int pcnn_deconv_bwd_filter(pcnn_handle h, int N, int hc, int wc, int Cin, int H, int W, int Cout, int f, const float* x, int ldx,
                           const float* dy, int lddy, float alpha, float* dk, float* dbias, void* workspace, size_t workspace_bytes);

/* ---- tf.image.resize(antialias=False, half-pixel centres) (layers/Upsample.py:56-59) ------------------------
 * Separable 4-tap gather: idx_y/wt_y are (Ho,4) tables, idx_x/wt_x (Wo,4) (nearest/bilinear use fewer taps with
 * zero weights).  Tables come from pcnn_resize_tables (host, float32 arithmetic of the TF kernels).
 * y = beta*y + alpha*resize(x). */
int pcnn_resize_tables(int method, int n_in, int n_out, int32_t* idx /*n_out*4*/, float* wt /*n_out*4*/);
int pcnn_resize_fwd(pcnn_handle h, int N, int hc, int wc, int C, int Ho, int Wo, const float* x, int ldx,
                    const int32_t* idx_y, const float* wt_y, const int32_t* idx_x, const float* wt_x,
                    float alpha, float beta, float* y, int ldy);
/* The resize branches of a merge in one pass over the destination (round 6; models/Homogeneous_Poisson_NN_Legacy.py:215-224: the three multilinear bottleneck
 * branches are summed into the merge buffer): y = beta*y + alpha*(resize(x_0) + ... + resize(x_{nsrc-1})), nsrc = 2 or 3, accumulated in the order and with the
 * roundings of nsrc consecutive pcnn_resize_fwd calls (beta, then 1, 1): bit-identical to them, with one read-modify-write of y instead of nsrc.  Eligible: C a
 * multiple of 4, 16-byte aligned tensors, channel strides multiples of 4. */
typedef struct pcnn_resize_src {
  const float* x; int hc, wc, ldx;
  const int32_t* idx_y; const float* wt_y; const int32_t* idx_x; const float* wt_x;
} pcnn_resize_src;
int pcnn_resize_fwd_multi_eligible(int N, int C, int Ho, int Wo, int nsrc, const pcnn_resize_src* src, const float* y, int ldy);
int pcnn_resize_fwd_multi(pcnn_handle h, int N, int C, int Ho, int Wo, int nsrc, const pcnn_resize_src* src, float alpha, float beta, float* y, int ldy);
/* dx = alpha * resize^T(dy); tmp must hold N*hc*Wo*C floats */
int pcnn_resize_bwd(pcnn_handle h, int N, int hc, int wc, int C, int Ho, int Wo, const float* dy, int lddy,
                    const int32_t* idx_y, const float* wt_y, const int32_t* idx_x, const float* wt_x,
                    float alpha, float* tmp, float* dx, int lddx);

/* ---- per-sample-filter ("metalearning") convolutions: ONE launch for the whole batch -------------------------------------------
 * Replace the tf.map_fn over samples of layers/metalearning_conv.py:148-169 (tf.pad + tf.nn.conv{1,2}d + bias per sample, filter and bias
 * emitted by the layer's hyper-network) and layers/metalearning_deconvupscale.py:104-137 (conv2d_transpose, kernel = stride).  Sample n
 * uses the filter at w + n * w_sample_stride (HWIO (kh,kw,Cin,Cout); transposed convolution (f,f,Cout,Cin)) and the bias at
 * bias + n * bias_sample_stride.  d->N = batch, <= 32 output channels, <= 31 taps; 1-D layers are kh = 1.
 *   _deconv_*  conv2d_transpose(kernel = stride = f, padding='SAME') exactly as TensorFlow crops it (layers/metalearning_deconvupscale.py:13-16):
 *           the coarse grid must be H = ceil(Ho / f), W = ceil(Wo / f) (TensorFlow raises otherwise, and so do these calls) and output pixel
 *           Y reads coarse row (Y + py) / f through tap (Y + py) % f with py = (H f - Ho) / 2 (likewise in x) - the same offset as
 *           pcnn_deconv_fwd.  _bwd_filter sums strips of coarse rows in a fixed order (deterministic; partials in the handle's scratch).
 *   _fwd    flip_transpose = 0: y = act(conv(pad(x), w_n) + b_n).  flip_transpose = 1: the same kernel as the DATA gradient - x is dz, the
 *           stored filter is the forward filter (kh,kw,Cout_of_this_call,Cin_of_this_call), read flipped and transposed.
 *   _wgrad  dw_n = filter gradient of sample n (not summed over the batch); workspace: pcnn_grouped_conv2d_wgrad_workspace(d) bytes.
 *   Routes: layers of <= 16 output and <= 16 input channels (every layer of the reference's metalearning configurations) run as a GROUPED
 *   IMPLICIT GEMM on the matrix cores (v_mfma_f32_4x4x1_16B_f32: 4 output channels x 64 pixels x one (tap, channel) step per instruction,
 *   exact fp32, the filter of sixteen K steps in one register through the instruction's A-broadcast); wider layers on the vector ALUs.
 *   pcnn_grouped_conv2d_uses_mfma(d, what) reports the route (what = 0: _fwd, 1: _wgrad); environment PCNN_GROUPED_VALU=1 forces the
 *   vector-ALU kernels. */
size_t pcnn_grouped_conv2d_wgrad_workspace(const pcnn_conv_desc* d);
int pcnn_grouped_conv2d_uses_mfma(const pcnn_conv_desc* d, int what);
int pcnn_grouped_conv2d_fwd(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* w, long long w_sample_stride, const float* bias,
                            long long bias_sample_stride, int flip_transpose, float* y);
int pcnn_grouped_conv2d_wgrad(pcnn_handle h, const pcnn_conv_desc* d, const float* x, const float* dz, float* dw, long long dw_sample_stride, void* workspace);
int pcnn_grouped_bias_grad(pcnn_handle h, int N, long long HW, int C, const float* dz, int lddz, float* dbias, long long bias_sample_stride);
int pcnn_grouped_deconv_fwd(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* x, const float* k,
                            long long k_sample_stride, const float* bias, long long bias_sample_stride, float* y);
int pcnn_grouped_deconv_bwd_data(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* dy, const float* k,
                                 long long k_sample_stride, float* dx);
int pcnn_grouped_deconv_bwd_filter(pcnn_handle h, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int f, const float* x, const float* dy,
                                   float* dk, long long k_sample_stride, float* dbias, long long bias_sample_stride);

/* ---- small dense layers (tf.keras.layers.Dense: models/Homogeneous_Poisson_NN_Legacy.py:99-102, layers/Scaling.py:31-33)
 * y[n,o] = act(b[o] + sum_i x[n,i] w[i,o]);  bwd: given dy and y, dx, dw (+=), db (+=).  b / db may be NULL: a Dense layer without bias
 * (use_bias=False reaches every Dense layer of the metalearning hyper-networks, layers/metalearning_conv.py:113,128) */
int pcnn_dense_fwd(pcnn_handle h, int N, int In, int Out, const float* x, const float* w, const float* b, int act, float alpha, float* y);
int pcnn_dense_bwd(pcnn_handle h, int N, int In, int Out, const float* x, const float* w, const float* y, const float* dy,
                   int act, float alpha, float* dx, float* dw, float* db);

/* ---- elementwise / layout helpers -------------------------------------------------------------------------- */
/* input assembly: out[n,y,x,0]=rhs, [1]=cos(pi*y/(H-1)), [2]=cos(pi*x/(W-1))  (models/..Legacy.py:172-180,198) */
int pcnn_assemble_input(pcnn_handle h, int N, int H, int W, const float* rhs, int use_pos, float* out, int ldo);
/* Strided convolutions (blocks/bottleneck_block.py:28-34, downsampling_method='conv'): the reference pads exactly as for stride 1
 * (utils/apply_advanced_padding_and_call_conv_layer.py:9-11) and lets the VALID convolution stride, i.e. it keeps every stride-th output
 * of the stride-1 result.  adjoint = 0: y (N, ceil(H/s), ceil(W/s), C) = x[:, ::s, ::s, :];  adjoint = 1: x is the coarse gradient, y the
 * full-resolution one (zero except at the kept positions). */
int pcnn_subsample(pcnn_handle h, int N, int H, int W, int C, int stride, const float* x, int ldx, float* y, int ldy, int adjoint);
/* y = alpha*x + beta*y over npix x C with channel strides */
int pcnn_axpby(pcnn_handle h, int64_t npix, int C, float alpha, const float* x, int ldx, float beta, float* y, int ldy);
/* y[n,p,c] = x[n,p,c]*s[n,c]  (tf.einsum('ijkl,ij->ijkl'), models/..Legacy.py:231); bwd gives dx and ds */
int pcnn_channel_scale_fwd(pcnn_handle h, int N, int64_t hw, int C, const float* x, int ldx, const float* s, float* y, int ldy);
int pcnn_channel_scale_bwd(pcnn_handle h, int N, int64_t hw, int C, const float* x, int ldx, const float* s, const float* dy, int lddy,
                           float* dx, int lddx, float* ds, void* workspace, size_t workspace_bytes);
size_t pcnn_channel_scale_workspace(int N, int64_t hw, int C);
/* pcnn_channel_scale_bwd with the activation backward of the layer that produced x fused in (round 6): x is that layer's saved activation output, so
 * dz = dy s act'(x) is written instead of dx - what pcnn_channel_scale_bwd followed by pcnn_conv2d_epilogue_bwd gives, bit for bit - and dbias (may be NULL)
 * receives the per-channel sums of dz in a fixed order.  workspace: 2 * pcnn_channel_scale_workspace(N, hw, C) bytes.  (The einsum of
 * models/Homogeneous_Poisson_NN_Legacy.py:231 follows the leaky-ReLU of post_merge_resnet's last convolution, :227-229.) */
int pcnn_channel_scale_bwd_post(pcnn_handle h, int N, int64_t hw, int C, const float* x, int ldx, const float* s, const float* dy, int lddy,
                                float* dz, int lddz, float* ds, int act, float act_alpha, float* dbias, void* workspace, size_t workspace_bytes);
/* per-sample scalar scale: y[n,..] = x[n,..]*(1+g[n]) (layers/Scaling.py:55); bwd: dx, dg[n] = sum dy*x */
int pcnn_sample_scale_fwd(pcnn_handle h, int N, int64_t per, const float* x, const float* g, float* y);
int pcnn_sample_scale_bwd(pcnn_handle h, int N, int64_t per, const float* x, const float* g, const float* dy, float* dx, float* dg);
/* BC ring (models/..Legacy.py:251): Dirichlet -> ring = 0 ; Neumann -> ring = adjacent interior.  1 channel. */
int pcnn_bc_ring_fwd(pcnn_handle h, int N, int H, int W, int neumann, const float* x, float* y);
int pcnn_bc_ring_bwd(pcnn_handle h, int N, int H, int W, int neumann, const float* dy, float* dx);
/* Spatial pyramid max-pool over channels and spatial bins (layers/SpatialPyramidPool.py:35-66).
 * bins: nb x 4 int32 (y0,y1,x0,x1) on device; out (N, nb); argmax (N, nb) int32 flat index into (H,W,C) for bwd */
int pcnn_spp_max_fwd(pcnn_handle h, int N, int H, int W, int C, int nb, const int32_t* bins, const float* x, float* out, int32_t* argmax);
int pcnn_spp_max_bwd(pcnn_handle h, int N, int H, int W, int C, int nb, const int32_t* argmax, const float* dout, float* dx);
/* Jacobi post-smoother (layers/JacobiIterationLayer.py:43-66), 3x3 second-order stencil, one sweep */
int pcnn_jacobi_sweep(pcnn_handle h, int N, int H, int W, const float* u, const float* rhs, const float* dx /*N x 2*/, float* out);
int pcnn_jacobi_sweep_bwd(pcnn_handle h, int N, int H, int W, const float* dout, const float* dx /*N x 2*/, float* du);

/* ---- loss (losses/loss_wrapper.py:53-71, losses/integral_loss.py:126-179) --------------------------------------
 * per-sample partial sums: out[n] = {sum|p-t|, sum (p-t)^2, sum G*(p-t)^2, max|t|}; G is the (H,W) quadrature map.
 * bwd: dpred[n,p] = c_mae[n]*sign(p-t) + (c_mse[n] + c_int[n]*G[p]) * 2 (p-t) */
int pcnn_loss_partials(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G, float* out /*N x 4*/);
int pcnn_loss_bwd(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G,
                  const float* c_mae, const float* c_mse, const float* c_int, float* dpred);
/* loss_wrapper bookkeeping on the N per-sample partials (losses/loss_wrapper.py:45-71): per-sample weights 1/peak^p,
 * division by the GLOBAL batch size, the scalar loss (+ *extra if given), the loss_bwd coefficients and the `mse` metric
 * of train_step (models/Homogeneous_Poisson_NN_Legacy.py:291). */
int pcnn_loss_coefficients(pcnn_handle h, int N, int64_t hw, const float* partials, float w_mae, float w_mse, float w_int,
                           int scale_by_peak, int global_batch_size, const float* extra, float* loss, float* c_mae, float* c_mse,
                           float* c_int, float* mse);
/* The same three with the integral term's exponent explicit (losses/integral_loss.py:88,153 `Lp_norm_power`: sum G (target - pred)^p; the
 * per-sample weight becomes 1 / peak^p, losses/loss_wrapper.py:68).  The entry points above are these with p = 2 (every shipped config). */
int pcnn_loss_partials_p(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G, float lp_power, float* partials);
int pcnn_loss_bwd_p(pcnn_handle h, int N, int64_t hw, const float* pred, const float* target, const float* G, const float* c_mae,
                    const float* c_mse, const float* c_int, float lp_power, float* dpred);
int pcnn_loss_coefficients_p(pcnn_handle h, int N, int64_t hw, const float* partials, float w_mae, float w_mse, float w_int, float lp_power,
                             int scale_by_peak, int global_batch_size, const float* extra, float* loss, float* c_mae, float* c_mse,
                             float* c_int, float* mse);
/* FD-Laplacian residual loss (losses/physics_informed_loss.py:35-50): per-sample sum of (rhs - conv(pred,kern_n))^2 over
 * the interior; kern (N, s, s); bwd accumulates into dpred.  */
int pcnn_pi_loss_partials(pcnn_handle h, int N, int H, int W, int s, const float* pred, const float* rhs, const float* kern, float* out /*N*/);
int pcnn_pi_loss_bwd(pcnn_handle h, int N, int H, int W, int s, const float* pred, const float* rhs, const float* kern,
                     const float* coef /*N*/, float* dpred);

/* ---- optimizer: tf.keras.optimizers.Adam (train/utils.py:3-8), flat parameter bucket ---------------------------- */
int pcnn_adam_step(pcnn_handle h, int64_t n, float* w, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                   float eps, int step, float grad_scale);
/* Adam(amsgrad=True): vhat = max(vhat, v) is the denominator's second moment (vhat == NULL: plain Adam) */
int pcnn_adam_amsgrad_step(pcnn_handle h, int64_t n, float* w, const float* g, float* m, float* v, float* vhat, float lr, float beta1,
                           float beta2, float eps, int step, float grad_scale);
int pcnn_sgd_step(pcnn_handle h, int64_t n, float* w, const float* g, float lr, float grad_scale);
/* tf.keras.optimizers.SGD(momentum, nesterov) (train/utils.py:7-8 hands optimizer_parameters to it): v = momentum v - lr g; w += v
 * (nesterov: w += momentum v - lr g). */
int pcnn_sgd_momentum_step(pcnn_handle h, int64_t n, float* w, const float* g, float* velocity, float lr, float momentum, int nesterov,
                           float grad_scale);

/* ---- dataset: reference-solution generators (poisson_CNN/dataset) ------------------------------------------- */
/* Dirichlet 5-point FD Poisson solve by DST-I diagonalisation, fp64 on the f64 matrix cores; replaces
 * multigrid_poisson_solve + poisson_RHS (dataset/solvers/multigrid.py:98-150, dataset/solvers/cholesky.py:45-119):
 *   A u_int = -dx^2 f_int + (boundary values folded onto the first interior ring),  A = pyamg.gallery.poisson((H-2, W-2)).
 * rhs (N,H,W) fp32; left/right (N,W) are the values on axis -2 index 0 / -1, bottom/top (N,H) those on axis -1 index
 * 0 / -1 (multigrid.py:145-148); dx (N).  S_h ((H-2)^2), lam_h (H-2), S_w, lam_w: device copies of pcnn_dst_setup output.
 * tmp: 2*N*(H-2)*(W-2) doubles.  soln (N,H,W) fp32 (fp64 solve, fp32 I/O like the reference's cast at
 * dataset/generators/numerical.py:131). */
int pcnn_dst_setup(int n, double* S /* (n-2)^2, host */, double* lam /* n-2, host */);
int pcnn_fd_poisson_dst(pcnn_handle h, int N, int H, int W, const float* rhs, const float* left, const float* right,
                        const float* bottom, const float* top, const float* dx, const double* S_h, const double* lam_h,
                        const double* S_w, const double* lam_w, double* tmp, float* soln);
/* The same solve with the two DST-I passes done by rocFFT (BASELINE.json north star: "HIP stencil + rocFFT kernel"): odd extension of the
 * right-hand side to 2 (H-1) x 2 (W-1), batched real 2-D FFT, division by the eigenvalues, the same FFT again - O(n^2 log n) against the
 * 8 n^3 FLOP of the GEMM form above, fp64 throughout.  rocFFT is bound at run time (dlopen; PCNN_ROCFFT_LIBRARY overrides the name); plans are
 * kept per (device, shape, batch).  lam_h / lam_w: the eigenvalues pcnn_dst_setup returns; workspace: pcnn_fd_poisson_fft_workspace bytes.
 * Measured per sample (tools/bench_fd_solver.py): 512^2 0.035 (GEMM) vs 0.076 ms, 1024^2 0.25 vs 0.35, 2048^2 1.86 vs 1.66 - the odd extension
 * quadruples the data and its lengths are awkward (2046 = 2 3 11 31), so poisson_cnn_amd.dataset takes this route from 2048 points per axis
 * (solver='auto'); below that the fp64 matrix cores win and the GEMM form is this route's checker in the tests. */
size_t pcnn_fd_poisson_fft_workspace(int N, int H, int W);
int pcnn_fd_poisson_fft(pcnn_handle h, int N, int H, int W, const float* rhs, const float* left, const float* right, const float* bottom,
                        const float* top, const float* dx, const double* lam_h, const double* lam_w, void* workspace, float* soln);
/* Mixed Dirichlet / Neumann 5-point solve on the same vertex-centred grid (the consumer in the reference is the
 * pressure projection of Navier_Stokes_2D/solvers.py:225-335, whose pure-Neumann system carries a zero-integral constraint, :258-259).
 * neumann_mask bit 0/1/2/3: left / right / bottom / top edge is Neumann - its array then holds du/dn (outward normal), its nodes are
 * unknowns closed by the second-order ghost node; otherwise the array holds the Dirichlet values.  All four Neumann: the singular mode
 * is dropped (solution with zero trapezoidal integral; the data need not be compatible).  Per axis the caller passes the eigen-decomposition
 * of the 1-D operator on that axis' mh (mw) unknowns: Vinv_h, V_h (mh x mh, row-major), lam_h; VinvT_w, VT_w (mw x mw, transposed), lam_w.
 * tmp: 2*N*mh*mw doubles. */
int pcnn_fd_poisson_mixed(pcnn_handle h, int N, int H, int W, int neumann_mask, const float* rhs, const float* left, const float* right,
                          const float* bottom, const float* top, const float* dx, const double* Vinv_h, const double* V_h,
                          const double* lam_h, const double* VinvT_w, const double* VT_w, const double* lam_w, double* tmp, float* soln);
/* C[b] = A[b] * B[b] in fp64 (row-major, stride 0 broadcasts an operand) on v_mfma_f64_16x16x4_f64 */
int pcnn_batched_gemm_f64(pcnn_handle h, int batch, int M, int Nn, int K, const double* A, int64_t strideA, int lda,
                          const double* B, int64_t strideB, int ldb, double* C, int64_t strideC, int ldc);
/* Separable series synthesis out[n,a,b] (+)= sum_{A<ka,B<kb} c[n,A,B] f((A+1) x_a) f((B+1) y_b), x = linspace(0,pi,H),
 * y = linspace(0,pi,W), f = sin (trig 0) or cos (trig 1) (dataset/utils/generate_smooth_function.py:45-62). */
int pcnn_series_synthesis(pcnn_handle h, int N, int H, int W, int ka, int kb, const float* coef, int trig, int accumulate, float* out);
/* out[n,a,b] (+)= sum_{r<R} U[n,r,a] V[n,r,b]: the Taylor component X(x)Y(y), X''Y + XY'' (dataset/generators/reverse.py:231-256) */
int pcnn_separable_sum(pcnn_handle h, int N, int H, int W, int R, const float* U, const float* V, int accumulate, float* out);
/* x[n,:] *= target[n] / max|x[n,:]| (dataset/utils/set_max_magnitude.py:14-50); factors (optional) receives the scale */
int pcnn_set_max_magnitude(pcnn_handle h, int N, int64_t per, const float* target, float* x, float* factors);
/* x[n,:] *= s[n] */
int pcnn_scale_samples(pcnn_handle h, int N, int64_t per, const float* s, float* x);

/* tf.keras.layers.LayerNormalization() over the last axis of an (N, F) matrix (epsilon 1e-3 in Keras): the optional final layer of the
 * metalearning hyper-networks (layers/metalearning_conv.py:128-129).  fwd also returns the per-row mean and 1/sqrt(var + eps) for bwd. */
int pcnn_layernorm_fwd(pcnn_handle h, int N, int F, const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean, float* rstd);
int pcnn_layernorm_bwd(pcnn_handle h, int N, int F, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy,
                       float* dx, float* dgamma, float* dbeta);

/* Row softmax of an (N, F) matrix - the 'softmax' activation of a tf.keras Dense layer (models/Dirichlet_BC_NN_Metalearning.py:69-76, its
 * __main__ config :236-239) - and its backward dx = y (dy - sum_f dy y).  dx may alias dy. */
int pcnn_softmax_fwd(pcnn_handle h, int N, int F, const float* x, float* y);
int pcnn_softmax_bwd(pcnn_handle h, int N, int F, const float* y, const float* dy, float* dx);

/* ---- Dirichlet_BC_NN_Legacy_2 / Poisson_CNN_Legacy (kernels in csrc/dbcnn.hip) ------------------------------ */
/* out[n,y,0:3] = {bc[n,y], 1, cos(pi y/(L-1))}: the boundary input with its positional embeddings
 * (models/Dirichlet_BC_NN_Legacy.py:113-124,136-139); 1-D tensors are NHWC with H = 1 */
int pcnn_dbc_assemble_input(pcnn_handle h, int N, int L, const float* bc, float* out, int ldo);
/* Spatial pyramid AVERAGE pool over channels and spatial bins (layers/SpatialPyramidPool.py:10-11,35-66); bins as pcnn_spp_max_fwd */
int pcnn_spp_avg_fwd(pcnn_handle h, int N, int H, int W, int C, int ldx, int nb, const int32_t* bins, const float* x, float* out);
int pcnn_spp_avg_bwd(pcnn_handle h, int N, int H, int W, int C, int lddx, int nb, const int32_t* bins, const float* dout, float* dx);
/* tf.einsum('bmy,mx,bm->bmxy', f, sinh_table, d) written as NHWC (N,X,L,M+2) with the two positional-embedding channels
 * cos(pi x/(X-1)), cos(pi y/(L-1)) appended (models/Dirichlet_BC_NN_Legacy.py:150-156).  f: (N,L,M) stride ldf; sinh_table (M,X); d (N,M) */
int pcnn_dbc_expand_fwd(pcnn_handle h, int N, int X, int L, int M, const float* f, int ldf, const float* sinh_table, const float* d, float* out, int ldo);
size_t pcnn_dbc_expand_bwd_workspace(int N, int L, int M);
int pcnn_dbc_expand_bwd(pcnn_handle h, int N, int X, int L, int M, const float* dout, int lddo, const float* f, int ldf, const float* sinh_table,
                        const float* d, float* df, int lddf, float* dd, void* workspace, size_t workspace_bytes);
/* adjoint of pcnn_set_max_magnitude (dataset/utils/set_max_magnitude.py:14-25 inside a GradientTape): x = the UNSCALED input */
int pcnn_set_max_magnitude_bwd(pcnn_handle h, int N, int64_t per, const float* target, const float* x, const float* dy, float* dx);
/* y[n,0,:] = bc[n,:] (bc == NULL: zeros = the adjoint): tf.concat([expand_dims(bc), out[...,1:,:]], 2) (models/Dirichlet_BC_NN_Legacy.py:164) */
int pcnn_set_first_row(pcnn_handle h, int N, int X, int L, const float* bc, float* y);
/* flip_and_rotate_tensor for (N,H,W) fields (dataset/utils/flip_and_rotate_tensor.py:4-47): out = reverse(transpose?(in)) along the flagged
 * axes, scaled per sample by alpha[n] (NULL = 1) and optionally accumulated into out (the five-term sum of models/Poisson_CNN_Legacy.py:48) */
int pcnn_flip_rotate(pcnn_handle h, int N, int Ho, int Wo, int transpose, int flip_y, int flip_x, const float* in, const float* alpha,
                     int accumulate, float* out);

#ifdef __cplusplus
}
#endif
#endif
