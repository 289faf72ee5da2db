/*
 * ocr_hip.h — C ABI of libocr_hip.so: the MI355X (gfx950) kernels behind the
 * EAST / PixelLink training + decode hot path of BowieHsu/tensorflow_ocr.
 *
 * The reference has no FFI: its hot path is TensorFlow-1.4 library ops reached
 * through Python call sites.  Each entry point below names
 * the reference call site(s) whose TF op it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (the Python host
 *     allocates through PyTorch-ROCm); the library allocates nothing persistent;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 = OCR_OK, negative = error (never throws across the boundary);
 *   - activations are NHWC, weights are HWIO at the API (TF layout), "f16" tensors
 *     are IEEE half, reductions and master weights are f32.
 */
#ifndef OCR_HIP_H_
#define OCR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported entry point changes its argument list (round 3 added arguments to
 * ocr_conv2d_bnred_f16, ocr_conv2d_bnred_tail_f16, ocr_bn_add_relu_f16, ocr_bn_relu_pool_idx_f16; round 4: the
 * batched head entry points and the seed-rank argument of ocr_link_cc_directed; round 5: the guest kernels).  The Python host refuses a
 * library whose ocr_abi_version() differs from the value it was written against (_lib.ABI_VERSION). */
#define OCR_ABI_VERSION 7

enum {
  OCR_OK = 0,
  OCR_ERR_INVALID_ARG = -1,
  OCR_ERR_UNSUPPORTED = -2,
  OCR_ERR_HIP = -3,
  OCR_ERR_WORKSPACE = -4,
  OCR_ERR_RCCL = -5            /* ocr_comm_last_error() has the RCCL message */
};

int ocr_abi_version(void);
const char* ocr_status_string(int status);
/* "f16" (libocr_hip.so: IEEE half storage, f16 MFMA) or "bf16" (libocr_hip_bf16.so: the same sources
 * built with bfloat16 storage and bf16 MFMA — BASELINE.json configs[3] "ResNet-v1-50 ... bf16").  The
 * `_f16` suffix of the entry points below names the 16-bit storage slot in both libraries. */
const char* ocr_storage_dtype(void);
/* HOST routine (no GPU): CRC-32C of n bytes continuing from `seed` (0 to start) — the per-tensor /
 * per-block checksum of TensorFlow checkpoint bundles (saver.save / saver.restore,
 * multigpu_train.py:188-189, test.py:146-150; tensorflow_ocr_amd/tf_bundle.py). */
uint32_t ocr_crc32c(const void* data, size_t n, uint32_t seed);

/* ------------------------------------------------------------------------- *
 * Convolution (slim.conv2d: nets/vgg.py:14-39, nets/resnet_v1.py:97-105,193,
 * nets/resnet_utils.py:111-122; backward = tf.gradients of the same ops,
 * multigpu_train.py:129).
 * ------------------------------------------------------------------------- */
typedef struct {
  int32_t n, h, w, cin;       /* input  [n,h,w,cin]  */
  int32_t oh, ow, cout;       /* output [n,oh,ow,cout] */
  int32_t kh, kw;
  int32_t stride, dilation;
  int32_t pad_top, pad_left;  /* TF SAME / conv2d_same padding resolved by host */
  int32_t flip_taps;          /* 1: use tap (kh-1-ky, kw-1-kx) — the dgrad form */
  int32_t flags;              /* OCR_CONV_* */
} ocr_conv_desc;

enum {
  OCR_CONV_BIAS = 1,       /* y += bias[cout]                                  */
  OCR_CONV_RELU = 2,       /* y = max(y,0)                                     */
  OCR_CONV_STATS = 4,      /* also emit per-tile column sum / sum-of-squares   */
  OCR_CONV_ACCUM_F16 = 8   /* y += previous y (dgrad into an existing grad)    */
};

/* Implicit-GEMM convolution, f16 in / f32 accumulate / f16 out, on MFMA.
 *   x      [n,h,w,cin] f16
 *   w_kc   [kh*kw][cout][cin] f16 (reduce dim contiguous; produced by
 *          ocr_pack_weights_f16)
 *   bias   [cout] f32 or NULL
 *   y      [n,oh,ow,cout] f16
 *   stats  [ocr_conv2d_num_mtiles][2][cout] f32 or NULL (OCR_CONV_STATS): per
 *          output tile, the sum and sum of squares of the f16-rounded outputs;
 *          reduced by ocr_bn_finalize.
 * Requires cin % 32 == 0 and cout % 64 == 0. */
int ocr_conv2d_f16(const ocr_conv_desc* d, const void* x, const void* w_kc,
                   const void* bias, void* y, void* stats, void* stream);
int ocr_conv2d_num_mtiles(const ocr_conv_desc* d);

/* Name of the kernel instantiation ocr_conv2d_f16 launches for `d`:
 * "conv_igemm_kernel<BN,CK,WCO,M16,TH>" (cout tile, channel chunk, cout waves, 16x16x32 MFMA, tile rows),
 * as it appears (mangled) in rocprofv3 kernel traces.  Measurement only. */
int ocr_conv2d_variant(const ocr_conv_desc* d, char* out, size_t cap);

/* conv3x3 + bias + ReLU + 2x2/2 max-pool (SAME: windows over an odd edge ignore the missing pixels) in one kernel for
 * 64 -> 64 channel layers on maps of >= 128 tiles (the persistent 64-channel kernel; OCR_ERR_UNSUPPORTED otherwise):
 * pooled [n][ceil(oh/2)][ceil(ow/2)][cout] f16 and argmax_u8 (may be NULL; ocr_maxpool_f16's format, position dy*2+dx of
 * the first maximum) are the ONLY outputs — for bias nets whose full-resolution activation nobody reads
 * (nets/vgg.py:17-18 under nets/pixellink.py:41-48: conv1_2 -> pool1).  d->flags: OCR_CONV_BIAS | OCR_CONV_RELU. */
int ocr_conv2d_relu_pool_f16(const ocr_conv_desc* d, const void* x, const void* w_kc, const void* bias, void* pooled,
                             void* argmax_u8, void* stream);

/* Input-gradient convolution fused with the batch-norm BACKWARD reduction of the layer below:
 * y (= gradient w.r.t. that layer's activation) is written as usual and `partial`
 * [ocr_conv2d_num_mtiles][2][cout] receives (sum dz, sum dz*xhat) per tile, dz = y * [relu(bn(bn_y)) > 0],
 * so ocr_bn_relu_bwd_apply_f16 can skip the reduction pass over the two tensors.
 * store_masked != 0: the STORED value is dz itself (the gradient past the layer's ReLU), not the raw gradient — for
 * layers without batch norm (PixelLink's VGG: activation = relu(conv + bias), nets/pixellink.py:41-48) pass their
 * activation as bn_y with scale 1, shift 0, mean 0, invstd 1: y then holds the gradient of (conv + bias) directly,
 * partial row kind 0 sums to the bias gradient (ocr_bn_bwd_sums) and ocr_bias_relu_bwd_f16's pass disappears. */
int ocr_conv2d_bnred_f16(const ocr_conv_desc* d, const void* x, const void* w_kc, void* y, void* partial,
                         const void* bn_y, const void* bn_scale, const void* bn_shift,
                         const void* bn_mean, const void* bn_invstd, int bn_relu, int store_masked, void* stream);
/* The same for the convolution that consumes conv1_1's activation (conv1_2, nets/vgg.py:17; 64 -> 64 channels: the
 * persistent 64-channel kernel only, OCR_ERR_UNSUPPORTED otherwise): conv1_1's y is not read but evaluated again in the
 * epilogue from the prepared image x4 [n][h][w][4] and conv1_1's packed weights (ocr_pack_weights_first_f16) by the
 * forward's own MFMA sequence — the same 16-bit values — so that y need not be stored at all: the forward then runs
 * ocr_conv2d_first_f16 with y = NULL (statistics only), ocr_conv2d_first_bn_relu_f16 for the activation, and the
 * backward this entry and ocr_conv2d_first_wgrad_bn_f16 with w_first. */
int ocr_conv2d_bnred_first_f16(const ocr_conv_desc* d, const void* x, const void* w_kc, void* y, void* partial,
                               const void* x4, const void* w_first, const void* bn_scale, const void* bn_shift,
                               const void* bn_mean, const void* bn_invstd, int bn_relu, void* stream);
/* ... and WITHOUT storing the gradient at all (round 5): conv1_1's weight gradient is the only other reader of conv1_2's
 * input gradient (conv1_1 has no input gradient of its own: nets/vgg.py:14-17), and with the batch-norm backward apply
 * dy = A dz + B y + C per channel, V the 27-value image patches and y = V W,
 *     dW = V^T dy = A .* (V^T dz) + B .* (M W) + C .* m,     M = V^T V, m = V^T 1  (ocr_conv2d_first_moments_keep_f16)
 * so the launch only has to leave S1 = V^T dz: s1_blocks [ocr_conv2d_bnred_first_wgrad_blocks(d)][32][64] f32, one block
 * per workgroup (slot ky*10 + kx*3 + c of the 32), accumulated on the matrix cores in the epilogue; `partial` as above.
 * ocr_bn_bwd_coefficients then turns `partial` into (A, B, C) and ocr_conv2d_first_wgrad_sums_f32 finishes dW — the
 * 1 GiB gradient tensor (32 x 512^2 x 64) is neither written nor read, and ocr_conv2d_first_wgrad_bn_f16's pass over it
 * disappears.  ocr_conv2d_bnred_first_wgrad_blocks: -1 where the launch is unsupported. */
int ocr_conv2d_bnred_first_wgrad_blocks(const ocr_conv_desc* d);
int ocr_conv2d_bnred_first_wgrad_f16(const ocr_conv_desc* d, const void* x, const void* w_kc, void* partial,
                                     const void* x4, const void* w_first, const void* bn_scale, const void* bn_shift,
                                     const void* bn_mean, const void* bn_invstd, int bn_relu, void* s1_blocks,
                                     void* stream);
int ocr_conv2d_first_wgrad_sums_f32(const void* s1_blocks, int blocks, const void* moments_f64, const void* w_first,
                                    int cout, const void* coef_a, const void* coef_b, const void* coef_c,
                                    void* dw_hwio_f32, void* stream);
/* Column sums of such partial rows [T][2][c] -> out0 (kind 0), out1 (kind 1); fixed order, one launch. */
int ocr_bn_bwd_sums(const void* partial, int T, int c, void* out0, void* out1, void* workspace, size_t ws_bytes,
                    void* stream);

/* Input-gradient convolution into the OUTPUT of a ResNet bottleneck, out = relu(shortcut + bn(bn_y))
 * (nets/resnet_v1.py:108-110), as the LAST contribution to that output's gradient (OCR_CONV_ACCUM_F16 adds
 * what the other consumers left in y): the stored value is dz = [tail_out > 0] * gradient — the gradient
 * past the ReLU, i.e. of the shortcut and of bn(bn_y) alike — and `partial` [ocr_conv2d_num_mtiles][2][cout]
 * receives (sum dz, sum dz*xhat(bn_y)) for ocr_bn_relu_bwd_apply_f16 (relu = 0).  Replaces one masking pass
 * and the reduction pass over the bottleneck's widest tensors.  1x1 and generic 3x3 tile kernels only:
 * OCR_ERR_UNSUPPORTED never occurs for 1x1 convolutions.
 * sub_grad (may be NULL): the gradient of subsample(out, 2) = out[:, ::2, ::2]
 * (nets/resnet_utils.py:59-75, the stride-2 identity shortcut), [n][ceil(oh/2)][ceil(ow/2)][cout] f16, added at
 * the even positions before the mask — instead of a zero-inserted full-size tensor under OCR_CONV_ACCUM_F16.
 * tail_mask_bits (may be NULL): the ReLU mask of tail_out as bits, one byte per 8 channels (written by
 * ocr_bn_add_relu_f16 / ocr_conv2d_pw_bnaddrelu_f16); when given it is read INSTEAD of tail_out (1/16 of the bytes). */
int ocr_conv2d_bnred_tail_f16(const ocr_conv_desc* d, const void* x, const void* w_kc, void* y, void* partial,
                              const void* bn_y, const void* bn_mean, const void* bn_invstd,
                              const void* tail_out, const void* tail_mask_bits, const void* sub_grad, void* stream);

/* 1x1 convolutions of a ResNet bottleneck with the element-wise pass in front of them APPLIED WHILE THE PIXEL
 * OPERAND IS LOADED (conv_pwx_kernel; 1x1, stride 1, cin >= 128, cin % 64 == 0, pixel count % 32 == 0, else
 * OCR_ERR_UNSUPPORTED), training mode:
 *  - ocr_conv2d_pw_bnaddrelu_f16: the input is the previous unit's output
 *      x = relu(bn(prev_y) + shortcut)        (nets/resnet_v1.py:107; sc_scale/sc_shift: projection shortcut's BN)
 *    computed on load and written to x_out [n,h,w,cin] (+ mask_bits, optional) for the other consumers;
 *    y = conv1x1(x) with BN partial sums under OCR_CONV_STATS — replaces ocr_bn_add_relu_f16 + ocr_conv2d_f16 and one
 *    full read of x;
 *  - ocr_conv2d_pw_bnbwd_bnred_f16 (d = the input-gradient form): the operand is the batch-norm backward apply
 *      dy = A*dz + B*y_above + C              (ocr_bn_bwd_coefficients)
 *    computed on load and written to dy_out (the weight gradient reads it next); dx = conv1x1^T(dy) with the fused
 *    BN-backward reduction of the layer below as in ocr_conv2d_bnred_f16 — replaces the apply half of
 *    ocr_bn_relu_bwd_apply_f16 and one full read of dy. */
int ocr_conv2d_pw_bnaddrelu_f16(const ocr_conv_desc* d, const void* prev_y, const void* prev_scale,
                                const void* prev_shift, const void* shortcut, const void* sc_scale,
                                const void* sc_shift, void* x_out, void* mask_bits, const void* w_kc, void* y,
                                void* stats, void* stream);
/*  - ocr_conv2d_pw_bnrelu_f16: the input is x = relu(prev_y*prev_scale + prev_shift), the batch norm + ReLU of the
 *    convolution before it (a bottleneck's conv2 -> conv3, nets/resnet_v1.py:99-105), computed on load with
 *    ocr_bn_relu_f16's arithmetic and written to x_out for the weight gradient — replaces ocr_bn_relu_f16 +
 *    ocr_conv2d_f16 and one full read of x.  cin >= 128. */
int ocr_conv2d_pw_bnrelu_f16(const ocr_conv_desc* d, const void* prev_y, const void* prev_scale, const void* prev_shift,
                             void* x_out, const void* w_kc, void* y, void* stats, void* stream);
int ocr_conv2d_pw_bnbwd_bnred_f16(const ocr_conv_desc* d, const void* dz, const void* y_above, const void* coef_a,
                                  const void* coef_b, const void* coef_c, void* dy_out, const void* w_kc, void* dx,
                                  void* partial, const void* bn_y, const void* bn_scale, const void* bn_shift,
                                  const void* bn_mean, const void* bn_invstd, int bn_relu, void* stream);
/* The same loader in front of the pointwise kernel's OTHER epilogues, and for a batch norm that is followed by a ReLU
 * (a bottleneck's conv1; the projection shortcut's BN has none): relu_shift (nullable) = that BN's shift, dz is then the
 * ACTIVATION's gradient and dy = A*(dz * [A*y_above + relu_shift > 0]) + B*y_above + C (A is the BN's scale).  Epilogue:
 * plain store, accumulate into dx (OCR_CONV_ACCUM_F16 in d->flags), or — bn_y ... sub_grad given as for
 * ocr_conv2d_bnred_tail_f16 — the bottleneck tail (the gradient past the previous unit's output ReLU + that unit's
 * BN-backward sums in `partial`); all of them null: no tail, `partial` unused.  Replaces the apply pass
 * ocr_bn_relu_bwd_apply_f16 in front of ocr_conv2d_bnred_tail_f16 / ocr_conv2d_f16.  cout of the convolution >= 128. */
int ocr_conv2d_pw_bnbwd_tail_f16(const ocr_conv_desc* d, const void* dz, const void* y_above, const void* coef_a,
                                 const void* coef_b, const void* coef_c, const void* relu_shift, void* dy_out,
                                 const void* w_kc, void* dx, void* partial, const void* bn_y, const void* bn_mean,
                                 const void* bn_invstd, const void* tail_out, const void* tail_mask_bits,
                                 const void* sub_grad, void* stream);

/* First-layer convolution (cin = 3, images [n,h,w,4] f16 with channel 3 zero,
 * produced by ocr_prep_images): 3x3 stride 1, pad 1.
 *   w_first [3][cout][16] f16: for tap row ky, k = kx*4 + c (kx<3, c<3), else 0.
 * nets/vgg.py:14 (conv1_1). */
int ocr_conv2d_first_f16(int n, int h, int w, int cout, const void* x4,
                         const void* w_first, const void* bias, int flags,
                         void* y, void* stats, void* stream);
int ocr_conv2d_first_num_mtiles(int n, int h, int w);

/* Weight gradient: dw[ky,kx,ci,co] = sum_{n,y,x} x[n, y*s+ky*d-pt, x*s+kx*d-pl, ci]
 * * dy[n,y,x,co], f16 operands, f32 result (HWIO).  `workspace` holds split-K
 * partial slabs; its size comes from ocr_conv2d_wgrad_workspace.  When
 * `accumulate` != 0 the result is added to dw. */
int ocr_conv2d_wgrad_f16(const ocr_conv_desc* d, const void* x, const void* dy,
                         void* dw_hwio_f32, void* workspace, size_t ws_bytes,
                         void* stream);
size_t ocr_conv2d_wgrad_workspace(const ocr_conv_desc* d);
/* The two halves of ocr_conv2d_wgrad_f16 as calls of their own — the split-K slab kernel, and the fixed-order sum of its
 * slabs into dw — so that the recorded step can issue the first as the host of a guest pass (ocr_bn_relu_bwd_apply_
 * affine_f16 ...) and the second behind the join: the slab sum is bandwidth- and cache-sensitive (13 us alone, 60-370 us
 * beside an HBM-streaming guest).  `workspace` (ocr_conv2d_wgrad_workspace bytes) must stay the launch's own until the
 * reduce has run.  Results are those of ocr_conv2d_wgrad_f16 bit for bit. */
int ocr_conv2d_wgrad_slabs_f16(const ocr_conv_desc* d, const void* x, const void* dy, void* workspace, size_t ws_bytes,
                               void* stream);
int ocr_conv2d_wgrad_reduce_f32(const ocr_conv_desc* d, const void* workspace, void* dw_hwio_f32, void* stream);
/* Registers per lane that stay free on a SIMD beside the kernel ocr_conv2d_wgrad_slabs_f16 launches for `d` (56 beside
 * wgrad3_kernel<9,128>, 248 beside <9,64>, 80 beside the 256 x 256 pointwise tile; 0 = none / unknown): whether a guest
 * pass (<= 56) can be placed beside it.  The recorded step holds back only weight gradients that can host. */
int ocr_conv2d_wgrad_guest_room(const ocr_conv_desc* d);

/* First-layer weight gradient (cin=3 from the [n,h,w,4] f16 image). dw [3,3,3,cout] f32. */
int ocr_conv2d_first_wgrad_f16(int n, int h, int w, int cout, const void* x4,
                               const void* dy, void* dw_hwio_f32,
                               void* workspace, size_t ws_bytes, void* stream);
/* The same weight gradient with conv1_1's batch-norm backward APPLY computed while the tile is staged: `da` is the
 * gradient of the layer's activation relu(bn(bn_y)); dy = A*dz + B*bn_y + C, dz = da * [bn_y*A + bn_shift rounds to a
 * positive 16-bit value] (relu) — coefficients from ocr_bn_bwd_coefficients (A = the layer's scale).  conv1_1 has no
 * input gradient, so nothing else reads dy: the apply pass and its 1 GiB output disappear (nets/vgg.py:14).
 * With w_first (the forward's packed weights) the kernel does not read bn_y either (it may be NULL): the tile's y is
 * evaluated again from the image halo it stages anyway, by the forward's own MFMA sequence (bit-identical 16-bit y). */
int ocr_conv2d_first_wgrad_bn_f16(int n, int h, int w, int cout, const void* x4, const void* da, const void* bn_y,
                                  const void* w_first, const void* bn_shift, const void* coef_a, const void* coef_b,
                                  const void* coef_c, int relu, void* dw, void* workspace, size_t ws_bytes,
                                  void* stream);
/* conv1_1 of the batch-norm nets, SECOND pass (the first, ocr_conv2d_first_f16 with OCR_CONV_STATS, produced the
 * statistics): a = relu(scale * y + shift) with y evaluated again from the image — 67 MB read instead of the 1 GiB of
 * y that ocr_bn_relu_f16 reads at 32 x 512^2; same values as that pass on the stored y. */
int ocr_conv2d_first_bn_relu_f16(int n, int h, int w, int cout, const void* x4, const void* w_first,
                                 const void* scale, const void* shift, int relu, void* a, void* stream);
size_t ocr_conv2d_first_wgrad_workspace(int n, int h, int w, int cout);

/* ResNet root convolution: 7x7 stride 2, cin = 3 (image [n,h,w,4] f16), explicit (3,3) padding of
 * resnet_utils.conv2d_same (nets/resnet_v1.py:193): y [n,(h-1)/2+1,(w-1)/2+1,cout].
 * w_stem [7][2][cout][16] f16 from ocr_pack_weights_stem_f16; dw [7,7,3,cout] f32. */
int ocr_conv2d_stem_num_mtiles(int n, int h, int w);
int ocr_pack_weights_stem_f16(const void* w_hwio_f32, int cout, void* w_stem, void* stream);
int ocr_conv2d_stem_f16(int n, int h, int w, int cout, const void* x4, const void* w_stem,
                        const void* bias, int flags, void* y, void* stats, void* stream);
size_t ocr_conv2d_stem_wgrad_workspace(int n, int h, int w, int cout);
int ocr_conv2d_stem_wgrad_f16(int n, int h, int w, int cout, const void* x4, const void* dy,
                              void* dw_hwio_f32, void* workspace, size_t ws_bytes, void* stream);
/* ... and with the root convolution's batch-norm backward apply computed while the dy tile is staged (as
 * ocr_conv2d_first_wgrad_bn_f16; nets/resnet_v1.py:193): da = gradient of relu(bn(bn_y)), coefficients from
 * ocr_bn_relu_bwd_reduce_f16. */
int ocr_conv2d_stem_wgrad_bn_f16(int n, int h, int w, int cout, const void* x4, const void* da, const void* bn_y,
                                 const void* bn_shift, const void* coef_a, const void* coef_b, const void* coef_c,
                                 int relu, void* dw, void* workspace, size_t ws_bytes, void* stream);
/* f32 HWIO master weights -> the f16 operand layouts of the MFMA kernels (done once per optimiser
 * step).  w_kc [taps][cout][cin] feeds ocr_conv2d_f16 forward; w_ck [taps][cin][cout] (a plain
 * cast) feeds the input gradient: call ocr_conv2d_f16 on dy with cin/cout swapped, flip_taps = 1
 * and pad' = dilation*(k-1) - pad.  Either output may be NULL. */
int ocr_pack_weights_f16(const void* w_hwio_f32, int taps, int cin, int cout, void* w_kc,
                         void* w_ck, void* stream);
int ocr_pack_weights_first_f16(const void* w_hwio_f32, int cout, void* w_first, void* stream);
/* conv1_1's batch-norm statistics from the IMAGE's second moments instead of a pass over the 64-channel output (round 5):
 * sum y_c = w_c . m, sum y_c^2 = w_c^T M w_c with m, M the sums of the 27-value patches and of their outer products —
 * one MFMA per 16 pixels (csrc/conv_first.hip).  stats_row [2][cout] f32 = what ocr_conv2d_first_f16(y = NULL,
 * OCR_CONV_STATS) + a sum over its partial rows would give for the f32 convolution outputs (the reference's fused batch
 * norm sees f32: nets/vgg.py:14), i.e. ocr_bn_finalize's input with T = 1; 1e-7-relative apart from the sums of the 16-bit
 * roundings the evaluating pass takes. */
size_t ocr_conv2d_first_moments_workspace(void);
int ocr_conv2d_first_moments_f16(int n, int h, int w, int cout, const void* x4, const void* w_first, void* stats_row,
                                 void* workspace, size_t ws_bytes, void* stream);
/* ... keeping the moments: moments_f64 [32][32] doubles (rows / columns k = (ky*3+kx)*3 + c; row 27 = the patch sums m,
 * [27][27] = the pixel count) for ocr_conv2d_first_wgrad_sums_f32; NULL = ocr_conv2d_first_moments_f16. */
int ocr_conv2d_first_moments_keep_f16(int n, int h, int w, int cout, const void* x4, const void* w_first,
                                      void* stats_row, void* moments_f64, void* workspace, size_t ws_bytes, void* stream);
/* ocr_pack_weights_f16 for n layers in ONE launch (the re-pack that follows every optimiser step).
 * ocr_pack_weights_batch_table fills a HOST table of ocr_pack_weights_batch_table_bytes(n) bytes from n
 * (weights, taps, cin, cout, w_kc, w_ck) tuples of DEVICE pointers (w_kc[i] or w_ck[i] may be NULL) and returns the
 * launch grid; the caller copies the table to device memory once (the pointers are static) and passes the
 * device copy to ocr_pack_weights_batch_f16 every step.  ld_ck (may be NULL): row length of w_ck[i], 0 = cout[i]; the
 * fuse heads' pair (w_kc32 [32][cin], w_ck32 [cin][32]; rows / columns >= cout stay as the caller zeroed them) is
 * (taps 1, cout, ld_ck 32). */
size_t ocr_pack_weights_batch_table_bytes(int n);
int ocr_pack_weights_batch_table(int n, const void* const* w_hwio_f32, const int* taps, const int* cin,
                                 const int* cout, void* const* w_kc, void* const* w_ck, const int* ld_ck,
                                 void* table_host, int* grid_out);
int ocr_pack_weights_batch_f16(const void* table_dev, int n, int grid, void* stream);
/* head weights f32 [cin][cout<=32] -> w_kc32 f16 [32][cin] and w_ck32 f16 [cin][32], zero padded */
int ocr_pack_weights_small_f16(const void* w_f32, int cin, int cout, void* w_kc32, void* w_ck32,
                               void* stream);

/* ------------------------------------------------------------------------- *
 * Image preparation: mean_image_subtraction (nets/model.py:18-31) + f16 cast.
 * images f32 [npix][3] -> f16 [npix][4] (4th channel 0).
 * ------------------------------------------------------------------------- */
int ocr_prep_images_f16(const void* images_f32, int64_t npix, float m0, float m1, float m2,
                        void* out_f16x4, void* stream);
/* The PixelLink input pipeline's normalisation folded into the same pass: (x - m) / div (IEEE
 * division) — the `ssd_vgg_preprocessing` output train_pixellink.py:150-154 hands the queue; the
 * counterparts use (120, 120, 120) / 60. */
int ocr_prep_images_norm_f16(const void* images_f32, int64_t npix, float m0, float m1, float m2,
                             float div, void* out_f16x4, void* stream);

/* ------------------------------------------------------------------------- *
 * slim.batch_norm (decay .997, eps 1e-5, scale; nets/resnet_utils.py:232-246,
 * nets/model_vgg_16.py:144-157) in training mode, fused with ReLU and the
 * following slim.max_pool2d 2x2/2 SAME (nets/vgg.py:16-28).
 * ------------------------------------------------------------------------- */
/* partial [T][2][C] f32 (sum, sum of squares per row block; from OCR_CONV_STATS or ocr_sc_stats)
 * -> scale/shift (a = y*scale + shift), saved mean / 1/std, moving-average update with the
 * unbiased variance.  gamma/beta/moving_* may be NULL.  workspace >= ocr_bn_reduce_workspace. */
size_t ocr_bn_reduce_workspace(int T, int C);
int ocr_bn_finalize(const void* partial, int T, int C, double count, const void* gamma,
                    const void* beta, float eps, float decay, void* moving_mean, void* moving_var,
                    void* scale, void* shift, void* save_mean, void* save_invstd, void* workspace,
                    size_t ws_bytes, void* stream);
int ocr_bn_inference_params(const void* gamma, const void* beta, const void* moving_mean,
                            const void* moving_var, float eps, int C, void* scale, void* shift,
                            void* stream);
/* a = [relu](y*scale+shift) f16; pool = 0: a_full; pool = 2: a_pool [n,ceil(h/2),ceil(w/2),c]
 * (+ a_full if non-NULL). */
int ocr_bn_relu_f16(const void* y, const void* scale, const void* shift, int n, int h, int w, int c,
                    int relu, int pool, void* a_full, void* a_pool, void* stream);
/* Backward of the same fused op.  da_full: gradient w.r.t. the full-resolution activation (may be
 * NULL when pool = 2); da_pool: gradient w.r.t. the pooled activation, routed to the first maximum
 * of each window.  Writes dgamma, dbeta [c] f32 and dy f16 (gradient w.r.t. the conv output).
 * partial: f32 [ocr_bn_bwd_num_partials][2][c]; workspace >= ocr_bn_reduce_workspace(T, c). */
int ocr_bn_bwd_num_partials(int n, int h, int w, int c, int pool);
int ocr_bn_relu_bwd_f16(const void* y, const void* scale, const void* shift, const void* save_mean,
                        const void* save_invstd, const void* da_full, const void* da_pool, int n,
                        int h, int w, int c, int relu, int pool, void* dgamma, void* dbeta, void* dy,
                        void* partial, void* workspace, size_t ws_bytes, void* stream);

/* Per-channel sum / sum of squares partials [ocr_channel_stats_num_partials][2][c] of an f16
 * tensor [npix][c] (batch norm after a sub-sampled conv output). */
int ocr_channel_stats_num_partials(int64_t npix, int c);
int ocr_channel_stats_f16(const void* x, int64_t npix, int c, void* partial, void* stream);
/* ResNet bottleneck tail (nets/resnet_v1.py:107): out = relu(bn(y) + shortcut); its ReLU
 * backward dz = dout * [out > 0]; a += b on f16 gradients. */
/* sc_scale / sc_shift (both or neither): the shortcut is a PROJECTION whose own batch norm is applied here
 * (its normalised copy is never stored); mask_bits (optional): one byte per 8 channels, bit e = out[..+e] > 0. */
int ocr_bn_add_relu_f16(const void* y, const void* scale, const void* shift, const void* shortcut,
                        const void* sc_scale, const void* sc_shift, int64_t npix, int c, void* out,
                        void* mask_bits, void* stream);
/* BN-backward sums [T][2][c] (sum dz, sum dz*xhat) -> dgamma, dbeta and the coefficients of the apply step as an
 * affine map dy = A*dz + B*y + C, for ocr_conv2d_pw_bnbwd_bnred_f16.  count = elements per channel. */
int ocr_bn_bwd_coefficients(const void* partial, int T, int c, double count, const void* scale,
                            const void* save_mean, const void* save_invstd, void* dgamma, void* dbeta,
                            void* coef_a, void* coef_b, void* coef_c, void* workspace, size_t ws_bytes,
                            void* stream);
/* The reduction pass of ocr_bn_relu_bwd_f16 (sum dz, sum dz*xhat over y and da_full) + the same finalize: dgamma, dbeta
 * and (A, B, C), no apply pass — for layers whose dy has a single reader that applies it on load, or whose apply pass
 * runs as a guest (ocr_bn_relu_bwd_apply_affine_f16 ...).  da_pool (nullable; round 5): the gradient of the layer's
 * 2x2/2 max-pool [n,ceil(h/2),ceil(w/2),c], routed to each window's first maximum as ocr_bn_relu_bwd_f16(pool = 2) does.
 * partial: [ocr_bn_bwd_num_partials(n,h,w,c, da_pool ? 2 : 0)][2][c] f32 scratch. */
int ocr_bn_relu_bwd_reduce_f16(const void* y, const void* scale, const void* shift, const void* save_mean,
                               const void* save_invstd, const void* da_full, const void* da_pool, int n, int h, int w,
                               int c, int relu, void* dgamma, void* dbeta, void* coef_a, void* coef_b, void* coef_c,
                               void* partial, void* workspace, size_t ws_bytes, void* stream);
/* The same with the max-pool backward that produces da_full folded in: da_full is GATHERED from the pool's pooled
 * gradient and first-maximum index (k x k / stride window, pads as ocr_maxpool_f16; da_pooled, argmax: [n,oh,ow,c]) —
 * the value ocr_maxpool_bwd_f16 stores, bit for bit — summed into the partials while in registers, and written to
 * da_full_out [n,h,w,c] f16 (nullable) for the reader that applies the coefficients (ocr_conv2d_stem_wgrad_bn_f16):
 * one pass instead of a gather pass and a reduce pass over the full-resolution tensor.  n*h*w*c < 2^31. */
int ocr_bn_relu_bwd_reduce_pooled_f16(const void* y, const void* scale, const void* shift, const void* save_mean,
                                      const void* save_invstd, const void* da_pooled, const void* argmax, int n, int h,
                                      int w, int c, int k, int stride, int pad_top, int pad_left, int oh, int ow,
                                      int relu, void* da_full_out, void* dgamma, void* dbeta, void* coef_a, void* coef_b,
                                      void* coef_c, void* partial, void* workspace, size_t ws_bytes, void* stream);
int ocr_relu_bwd_f16(const void* out, const void* dout, int64_t n, void* dz, void* stream);
int ocr_add_inplace_f16(void* a, const void* b, int64_t n, void* stream);

/* unpool = tf.image.resize_bilinear x2 with TF-1.4 legacy sampling on f16 NHWC maps (EAST merge
 * branch, nets/model_vgg_16.py:15-16,121): x [n,lh,lw,c] -> y [n,2lh,2lw,c]; and its transpose. */
int ocr_unpool_f16(const void* x, int n, int lh, int lw, int c, void* y, void* stream);
int ocr_unpool_bwd_f16(const void* dy, int n, int lh, int lw, int c, void* dx, int accumulate, void* stream);
/* y [n,2lh,2lw,c] += unpool(t_low [n,lh,lw,c]) in place (sampling of ocr_unpool_f16), with the per-channel
 * (sum, sum of squares) partials [ocr_channel_stats_num_partials(n*4*lh*lw, c)][2][c] of the result (nullable): the
 * last step of EAST's merge convolution conv1x1(concat(unpool(g), f)) evaluated as unpool(conv_a(g)) + conv_b(f)
 * (nets/model_vgg_16.py:118-121 — the 1x1 convolution and the resize act on different axes and commute). */
int ocr_unpool_add_stats_f16(const void* t_low, int n, int lh, int lw, int c, void* y, void* partial, void* stream);

/* Backward of slim.conv2d's bias + ReLU (nets/pixellink.py:41-48): dz = da * [a > 0] (a = stored
 * conv output), dbias [c] = column sums.  partial: f32 [ocr_bias_relu_bwd_num_partials][c]. */
int ocr_bias_relu_bwd_num_partials(int64_t npix, int c);
int ocr_bias_relu_bwd_f16(const void* a, const void* da, int64_t npix, int c, int relu, void* dz,
                          void* dbias, void* partial, void* stream);

/* GUEST forms of the two apply passes below (csrc/guest_bn.hip): the apply step of the batch-norm (+ReLU) backward of
 * slim.conv2d under nets/model_vgg_16.py:144 (nets/vgg.py:14-39) as the affine map dy = A*dz + B*y + C with the
 * coefficients of ocr_bn_bwd_coefficients (A = coef_a = the layer's forward scale), dz = da * [fma(y, A, shift) rounds
 * to a positive 16-bit value] (relu) — ONE launch, no workspace, <= 56 registers per lane and no LDS, so that its waves
 * are placed beside a resident weight-gradient workgroup (456 of the 512 registers per lane): the recorded step runs
 * this pass of layer L-1 on a second stream beside held-back ocr_conv2d_wgrad_slabs_f16 launches.  max_workgroups: 0 =
 * 4 per CU (the pass alone); 256 = one per CU, what is resident beside such a host (csrc/guest_bn.hip).  Tensors < 2 GiB
 * (OCR_ERR_UNSUPPORTED otherwise, and for odd h / w in the pooled form: callers keep the general entry points). */
int ocr_bn_relu_bwd_apply_affine_f16(const void* y, const void* da, const void* scale, const void* shift,
                                     const void* coef_b, const void* coef_c, int n, int h, int w, int c, int relu,
                                     void* dy, int max_workgroups, void* stream);
/* The REDUCTION pass of an end-point layer's batch-norm backward (several consumers contributed to the activation's
 * gradient, so no convolution's epilogue summed it: conv3_3 / conv4_3 / conv5_3 / fc7 of nets/vgg.py under
 * nets/model_vgg_16.py:160-172) as a guest as well: partial rows [ocr_bn_relu_bwd_reduce_rows_count(...)][2][c] f32 of
 * (sum dz, sum dz*xhat), dz = (da_full + [first max] * da_pool) * [fma(y, scale, shift) rounds to a positive value], for
 * ocr_bn_bwd_coefficients.  da_pool / argmax_u8: both (a pooled end point: even h, w, relu) or neither.  The grid is at
 * most one workgroup per CU (max_workgroups: 0 = 256; a smaller value must be given to the row count too), so the rows do
 * not depend on where the recorded step places the launch.  <= 56 registers per lane, no LDS. */
int ocr_bn_relu_bwd_reduce_rows_count(int n, int h, int w, int c, int pooled, int max_workgroups);
int ocr_bn_relu_bwd_reduce_rows_f16(const void* y, const void* da_full, const void* da_pool, const void* argmax_u8,
                                    const void* scale, const void* shift, const void* save_mean, const void* save_invstd,
                                    int n, int h, int w, int c, int relu, void* partial, int max_workgroups, void* stream);
/* pooled END-POINT layers (conv3_3 / conv4_3: the heads read the full-resolution activation, the pool feeds the trunk):
 * dz = (da_full + [first max] * da_pool) * [ReLU mask of the position]; argmax_u8 from ocr_bn_relu_pool_idx_f16 called
 * with a_full; coefficients from ocr_bn_relu_bwd_reduce_f16(da_pool).  Even h, w. */
int ocr_bn_relu_poolfull_bwd_apply_affine_f16(const void* y, const void* da_full, const void* da_pool,
                                              const void* argmax_u8, const void* scale, const void* shift,
                                              const void* coef_b, const void* coef_c, int n, int h, int w, int c, int relu,
                                              void* dy, int max_workgroups, void* stream);
int ocr_bn_relu_pool_bwd_idx_apply_affine_f16(const void* y, const void* argmax_u8, const void* da_pool,
                                              const void* coef_a, const void* coef_b, const void* coef_c, int n, int h,
                                              int w, int c, int relu, void* dy, int max_workgroups, void* stream);

/* ocr_bn_relu_bwd_f16 without its reduction pass (partials [T][2][c] from ocr_conv2d_bnred_f16) */
int ocr_bn_relu_bwd_apply_f16(const void* y, const void* scale, const void* shift, const void* save_mean,
                              const void* save_invstd, const void* da_full, int n, int h, int w, int c,
                              int relu, const void* partial, int T, void* dgamma, void* dbeta, void* dy,
                              void* workspace, size_t ws_bytes, void* stream);
/* Pooled conv+BN+ReLU layers whose only consumer is the 2x2/2 max-pool (conv1_2, conv2_2 of
 * nets/vgg.py:18,22): the forward also stores, one byte per pooled element [n][oh][ow][c], the first-max
 * position (bits 0-1 = dy*2+dx) and whether the pooled activation is positive (bit 2); the backward routes da_pool through it (tf.nn.max_pool's gradient
 * goes to the first maximum) without recomputing the candidates' activations.  Same results as
 * ocr_bn_relu_f16(pool=2) / ocr_bn_relu_bwd_f16(pool=2, da_full=NULL). */
int ocr_bn_relu_pool_idx_f16(const void* y, const void* scale, const void* shift, int n, int h, int w, int c,
                             int relu, void* a_full, void* a_pool, void* argmax_u8, void* y_pool, void* stream);
int ocr_bn_relu_pool_bwd_idx_f16(const void* y, const void* scale, const void* save_mean, const void* save_invstd,
                                 const void* a_pool, const void* argmax_u8, const void* da_pool, int n, int h,
                                 int w, int c, int relu, void* dgamma, void* dbeta, void* dy, void* partial,
                                 void* workspace, size_t ws_bytes, void* stream);
/* y_pool (optional output of ocr_bn_relu_pool_idx_f16, [n][oh][ow][c] f16): the conv output y AT each window's first
 * maximum.  The layer's dz is zero away from those positions, so the BN-backward sums of the whole layer are sums over
 * the POOLED positions: the input-gradient kernel that produces da_pool takes them in its epilogue
 * (ocr_conv2d_bnred_f16 with bn_y = y_pool and the layer's scale / shift / mean / invstd), and the backward is this
 * apply-only form — the reduction pass over y (1 GiB at conv1_2 of nets/vgg.py:17) disappears. */
int ocr_bn_relu_pool_bwd_idx_apply_f16(const void* y, const void* scale, const void* save_mean,
                                       const void* save_invstd, const void* argmax_u8, const void* da_pool, int n,
                                       int h, int w, int c, int relu, const void* partial, int T, void* dgamma,
                                       void* dbeta, void* dy, void* workspace, size_t ws_bytes, void* stream);


/* slim.max_pool2d k x k / stride SAME as a standalone op (pool5 3x3/1 nets/vgg.py:32; ResNet
 * pool1 3x3/2 nets/resnet_v1.py:194; subsample 1x1/s nets/resnet_utils.py:74), f16 NHWC.
 * argmax (optional, u8 [n,oh,ow,c]): window position ky*k+kx of the FIRST maximum (TF's gradient
 * routing); the backward pass takes either it (no re-scan of x) or x itself. */
int ocr_maxpool_f16(const void* x, int n, int h, int w, int c, int k, int stride, int pad_top,
                    int pad_left, int oh, int ow, void* y, void* argmax, void* stream);
/* The same pool over relu(bn_y*scale + shift) (each element rounded to 16 bits as ocr_bn_relu_f16 stores it), read
 * from the RAW conv output: the activation is never written when the pool is its only reader (ResNet root,
 * nets/resnet_v1.py:193-194). */
int ocr_bn_relu_maxpool_f16(const void* bn_y, const void* scale, const void* shift, int relu, int n, int h, int w,
                            int c, int k, int stride, int pad_top, int pad_left, int oh, int ow, void* y,
                            void* argmax, void* stream);
int ocr_maxpool_bwd_f16(const void* x, const void* argmax, const void* dy, int n, int h, int w, int c,
                        int k, int stride, int pad_top, int pad_left, int oh, int ow, void* dx,
                        int accumulate, void* stream);

/* ------------------------------------------------------------------------- *
 * PixelLink fuse heads (nets/model_vgg_16.py:160-175, nets/pixellink.py:55-67,
 * nets/model.py:129-141): wide f16 feature -> few f32 channels, and the f32
 * "small-channel" tensors [P][C] the unpool+add pyramid works on.
 * ------------------------------------------------------------------------- */
int ocr_conv1x1_small_f16(const void* x, const void* w_kc32, const void* bias, int P, int cin,
                          int cout, void* out_f32, void* stream);
int ocr_conv1x1_small_dgrad_f16(const void* dz_f32, const void* w_ck32, int P, int cin, int cout,
                                float grad_scale, void* dx_f16, int accumulate, void* stream);
size_t ocr_conv1x1_small_wgrad_workspace(int P, int cin, int cout);
int ocr_conv1x1_small_wgrad_f16(const void* x, const void* dz_f32, int P, int cin, int cout,
                                void* dw_f32, void* workspace, size_t ws_bytes, void* stream);
/* per-channel sum / sum-of-squares partials [ocr_sc_num_partials][2][C] of x f32 [P][C] */
int ocr_sc_num_partials(int P, int C);
int ocr_sc_stats(const void* x, int P, int C, void* partial, void* stream);
/* out[n,h,w,C] = [act(za*sa+ha)] + [act(zb*sb+hb)] + [unpool2x(prev[n,h/2,w/2,C])]; any subset;
 * unpool = tf.image.resize_bilinear x2 with TF-1.4 legacy sampling (nets/model.py:14-15). */
int ocr_sc_fuse(const void* za, const void* sa, const void* ha, const void* zb, const void* sb,
                const void* hb, const void* prev, int n, int h, int w, int C, int relu, void* out,
                void* stream);
int ocr_sc_unpool_bwd(const void* dout, int n, int lh, int lw, int C, void* dprev, void* stream);
/* BN(+ReLU) backward on f32 [P][C]; partial must hold (ocr_sc_num_partials + 1)*2*C floats */
int ocr_sc_bn_bwd(const void* z, const void* scale, const void* shift, const void* save_mean,
                  const void* save_invstd, const void* dout, int P, int C, int relu, void* dgamma,
                  void* dbeta, void* dz, void* partial, void* stream);
/* out[c] = sum_p x[p][c]; partial: (ocr_sc_num_partials + 1) * 2 * C floats */
int ocr_sc_colsum(const void* x, int P, int C, void* out, void* partial, void* stream);
/* sigmoid heads (F_score / geo_map, nets/model_vgg_16.py:129-131) and their gradient, f32 */
int ocr_sc_sigmoid(const void* z, int64_t n, void* out, void* stream);
int ocr_sc_sigmoid_bwd(const void* out, const void* dout, int64_t n, void* dz, void* stream);
/* both heads from ONE merged pre-activation map z [P][C] (one pass over the feature for the two convolutions):
 * out0 [P][c0] = sigmoid(z[:, :c0]), out1 [P][C-c0] = sigmoid(z[:, c0:]); and dz [P][C] from the two activation maps and
 * their gradients (a null gradient counts as zero). */
int ocr_sc_sigmoid_split(const void* z, int P, int C, int c0, void* out0, void* out1, void* stream);
int ocr_sc_sigmoid_split_bwd(const void* out0, const void* dout0, const void* out1, const void* dout1, int P, int C,
                             int c0, void* dz, void* stream);
/* pointwise f32 conv on channel slices: out[p][oo+co] = b[co] + sum_ci x[p][xo+ci] w[ci][co] */
int ocr_sc_pointwise_fwd(const void* x, int ldx, int xo, int cin, const void* w, const void* bias,
                         int P, void* out, int ldo, int oo, int cout, void* stream);
int ocr_sc_pointwise_dgrad(const void* dout, int ldo, int oo, int cout, const void* w, int P,
                           void* dx, int ldx, int xo, int cin, void* stream);
size_t ocr_sc_pointwise_wgrad_workspace(int cin, int cout);
int ocr_sc_pointwise_wgrad(const void* x, int ldx, int xo, int cin, const void* dout, int ldo,
                           int oo, int cout, int P, void* dw, void* db, void* workspace,
                           size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Dice loss (nets/model_vgg_16.py:179-225 == nets/model.py:145-159):
 * loss = 2*dice(pixel) + sum_8 dice(link_i), labels/mask 1 channel, predictions
 * pc (pixel) / G per direction (link) channels with TF broadcasting.
 * sums27: (I, sum y*m, sum p*m) x 9; loss10: [total, 9 dice terms].
 * ------------------------------------------------------------------------- */
size_t ocr_dice_workspace(int P);
int ocr_dice_loss_fwd(const void* y_true_pixel, const void* y_pred_pixel, int pc,
                      const void* y_true_link, const void* y_pred_link, int G,
                      const void* training_mask, int P, void* sums27, void* loss10,
                      void* workspace, size_t ws_bytes, void* stream);
int ocr_dice_loss_bwd(const void* y_true_pixel, int pc, const void* y_true_link, int G,
                      const void* training_mask, int P, const void* sums27, float grad_scale,
                      void* d_pred_pixel, void* d_pred_link, void* stream);

/* ------------------------------------------------------------------------- *
 * Softmax cross-entropy losses with online hard negative mining / focal links.
 *   pixel_rule 0: nets/model.py:204-261 `loss` (OHNM: per image the k = min(neg_ratio*n_pos,
 *                 n_neg) negatives with the smallest P(neg), tie-inclusive; pixel term
 *                 sum(CE*selected)/n_pos_batch, 0 without positives)
 *   pixel_rule 1: nets/model_vgg_16.py:243-282 `ohem_loss` (positives only)
 *   pixel_rule 2: nets/pixellink.py:88-263 `PixelLinkNet.build_loss` (mean over all pixels)
 *   label_rule 0: pos = (label == 1), neg = (label == 0);  1: pos = (label > 0), neg = !pos
 *   link_gate  1: link weights times the selected pixel mask, unguarded division (model.py);
 *              0: ungated, zero-guarded (pixellink.py:198-212)
 *   focal      1: link CE -> -alpha_t (1-p_t)^gamma log p_t (build-defined)
 * total = sum_8 link_i + 2*pixel.  Tensors f32: pixel_logits [n][hw][2], link_logits
 * [n][hw][16], pixel_labels [n][hw], link_labels [n][hw][8].  Outputs: ohnm_threshold [n],
 * sums34, loss10 = [total, pixel, link_0..7].
 * ------------------------------------------------------------------------- */
typedef struct {
  int32_t n, hw;
  int32_t pixel_rule, label_rule, link_gate, focal;
  float neg_ratio, alpha, gamma;
} ocr_softmax_loss_desc;
size_t ocr_softmax_loss_workspace(const ocr_softmax_loss_desc* d);
int ocr_softmax_loss_fwd(const ocr_softmax_loss_desc* d, const void* pixel_logits,
                         const void* link_logits, const void* pixel_labels, const void* link_labels,
                         void* ohnm_threshold, void* sums34, void* loss10, void* workspace,
                         size_t ws_bytes, void* stream);
/* mask_u8 [n*hw] = 1 where the pixel CE term counts (positives + mined negatives): the
 * `selected` map of OHNM_batch (nets/model.py:186-197).  The mining score P(neg) is a fixed
 * sequence of IEEE f32 operations (loss_softmax.hip: det_exp) shared with the CPU checker, so the
 * threshold and this mask are bit-exact index outputs. */
int ocr_softmax_loss_selected(const ocr_softmax_loss_desc* d, const void* pixel_logits,
                              const void* pixel_labels, const void* ohnm_threshold, void* mask_u8,
                              void* stream);
int ocr_softmax_loss_bwd(const ocr_softmax_loss_desc* d, const void* pixel_logits,
                         const void* link_logits, const void* pixel_labels, const void* link_labels,
                         const void* ohnm_threshold, const void* sums34, float grad_scale,
                         void* d_pixel_logits, void* d_link_logits, void* stream);

/* ------------------------------------------------------------------------- *
 * f32 INFERENCE PRECISION (round 4; csrc/f32_infer.hip): the forward graph with f32 storage and arithmetic, the
 * convolutions on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMAs) — Graph(precision="f32"), test.py /
 * test_pixellink*.py --precision f32 — so that score / link maps are within the north star's 1e-3 of the reference
 * ON THE PRODUCT LIBRARY (16-bit storage costs ~1e-2 of the logit range after 16 layers).  Forward only.
 * Tensors NHWC f32; conv weights in the TF HWIO master layout [kh][kw][cin][cout] (no packing).
 * ------------------------------------------------------------------------- */
/* flags: OCR_CONV_BIAS, OCR_CONV_RELU, OCR_CONV_ACCUM_F16 (here: y += conv, f32); any kernel size, stride, dilation */
int ocr_conv2d_f32_mfma(const ocr_conv_desc* d, const void* x, const void* w_hwio, const void* bias, void* y,
                        void* stream);
int ocr_channel_stats_f32_num_partials(int64_t npix, int c);
/* partial [T][2][c] f32 = per-strip (sum, sum of squares): input of ocr_bn_finalize */
int ocr_channel_stats_f32(const void* x, int64_t npix, int c, void* partial, void* stream);
int ocr_bn_relu_f32(const void* y, const void* scale, const void* shift, int n, int h, int w, int c, int relu,
                    int pool, void* a_full, void* a_pool, void* stream);
int ocr_maxpool_f32(const void* x, int n, int h, int w, int c, int k, int stride, int pad_top, int pad_left,
                    int oh, int ow, void* y, void* stream);
int ocr_prep_images_f32(const void* images, int64_t npix, float mean_r, float mean_g, float mean_b, void* out,
                        void* stream);
int ocr_bn_add_relu_f32(const void* y, const void* scale, const void* shift, const void* shortcut,
                        int64_t npix, int c, void* out, void* stream);
int ocr_unpool_f32(const void* x, int n, int lh, int lw, int c, void* y, void* stream);

/* ------------------------------------------------------------------------- *
 * Fuse heads, BATCHED (round 4): one launch per kernel kind over the (up to four) feature maps the heads read
 * (nets/model_vgg_16.py:160-172: fc7, conv5_3, conv4_3, conv3_3; nets/pixellink.py:58-67; nets/model.py:129-141) instead
 * of one per map, and the two predication convolutions (:166,173 / :61,67) as one pass over the fused tensor.  `items` are
 * HOST arrays of the descriptors below (copied into the kernel arguments, like ocr_conv_desc); the pointers INSIDE them
 * are device pointers.  Convolution outputs and input gradients equal the per-map entry points' (ocr_conv1x1_small_*)
 * bit for bit; the forward statistics use another (equally fixed) summation order, the weight gradient rounds f32 dz on
 * load, the batched finalisations sum partial rows in f64, and the 16 -> 16 predication pass agrees with
 * ocr_sc_pointwise_* to the last ulp or two (the compiler schedules the 16-term FMA chains of the two kernels differently).
 * ------------------------------------------------------------------------- */
typedef struct {
  const void* x;             /* f16 [P][cin] feature map */
  const void* w_kc32;        /* f16 [32][cin], rows >= cout zero */
  const void* bias;          /* f32 [cout] or NULL */
  void* out;                 /* f32 [P][cout] */
  void* stats_partial;       /* NULL, or f32 [rows][2][cout]: per-workgroup sums of out and out^2, rows =
                                ocr_conv1x1_small_batch_rows(P) <= 1024 (for ocr_bn_finalize_batch) */
  int32_t P, cin, cout;
} ocr_head_conv_item;
int ocr_conv1x1_small_batch_rows(int P);
int ocr_conv1x1_small_batch_f16(const ocr_head_conv_item* items, int count, void* stream);
typedef struct {
  const void* dz;            /* f32 [P][cout] */
  const void* w_ck32;        /* f16 [cin][32] */
  void* dx;                  /* f16 [P][cin] */
  int32_t P, cin, cout, accumulate;
} ocr_head_dgrad_item;
int ocr_conv1x1_small_dgrad_batch_f16(const ocr_head_dgrad_item* items, int count, float grad_scale, void* stream);
typedef struct {
  const void* x;             /* f16 [P][cin], cin % 128 == 0 */
  const void* dz;            /* f32 [P][cout] (rounded to the 16-bit storage type on load) */
  void* dw;                  /* f32 [cin][cout] */
  void* slab;                /* scratch: ocr_conv1x1_small_wgrad_batch_slab_bytes(P, cin) bytes of its own */
  int32_t P, cin, cout;
} ocr_head_wgrad_item;
size_t ocr_conv1x1_small_wgrad_batch_slab_bytes(int P, int cin);
int ocr_conv1x1_small_wgrad_batch_f16(const ocr_head_wgrad_item* items, int count, void* stream);
/* batched finalisation of SMALL batch-norm reductions (T <= 2048 rows, C <= 32; up to 8 per launch): fields as the
 * arguments of ocr_bn_finalize / ocr_bn_bwd_sums */
typedef struct {
  const void* partial;       /* f32 [T][2][C] */
  int32_t T, C;
  double count;
  const void *gamma, *beta;
  void *moving_mean, *moving_var, *scale, *shift, *save_mean, *save_invstd;
} ocr_bn_finalize_item;
int ocr_bn_finalize_batch(const ocr_bn_finalize_item* items, int count, float eps, float decay, void* stream);
typedef struct {
  const void* partial;       /* f32 [T][2][C] */
  int32_t T, C;
  void *out0, *out1;         /* column sums of kind 0 / kind 1 */
} ocr_bn_sums_item;
int ocr_bn_bwd_sums_batch(const ocr_bn_sums_item* items, int count, void* stream);
typedef struct {
  const void *z, *scale, *shift, *save_mean, *save_invstd, *dout;
  void *dgamma, *dbeta, *dz;
  void* partial;             /* scratch of its own: ocr_sc_num_partials(P, C) * 2 * C floats */
  int32_t P, C, relu;
} ocr_sc_bn_bwd_item;
int ocr_sc_bn_bwd_batch(const ocr_sc_bn_bwd_item* items, int count, void* stream);
typedef struct {
  const void* x;             /* f32 [P][C] */
  void* out;                 /* f32 [C]: column sums */
  void* partial;             /* scratch of its own: (ocr_sc_num_partials(P, C) + 1) * 2 * C floats */
  int32_t P, C;
} ocr_sc_colsum_item;
int ocr_sc_colsum_batch(const ocr_sc_colsum_item* items, int count, void* stream);
typedef struct {
  const void *z, *scale, *shift;
  void* out;                 /* out = act(z * scale[c] + shift[c]) */
  int64_t total;             /* elements */
  int32_t C;
} ocr_sc_act_item;
int ocr_sc_act_batch(const ocr_sc_act_item* items, int count, int relu, void* stream);
/* x18 f32 [P][18] (channels 0..1: pixel head, 2..17: link head): z_px = x18[:, :2] w_px (+ b_px) [P][2], z_lk =
 * x18[:, 2:] w_lk (+ b_lk) [P][16]; with partial_* (f32 [rows][2][2] / [rows][2][16], rows =
 * ocr_sc_pointwise_pair_num_partials(P)) also the batch-norm statistics partials of both outputs. */
int ocr_sc_pointwise_pair_num_partials(int P);
int ocr_sc_pointwise_pair_fwd(const void* x18, const void* w_px, const void* b_px, const void* w_lk, const void* b_lk,
                              int P, void* z_px, void* z_lk, void* partial_px, void* partial_lk, void* stream);
size_t ocr_sc_pointwise_pair_bwd_workspace(void);
int ocr_sc_pointwise_pair_bwd(const void* x18, const void* dz_px, const void* dz_lk, const void* w_px, const void* w_lk,
                              int P, void* dx18, void* dw_px, void* db_px, void* dw_lk, void* db_lk, void* workspace,
                              size_t ws_bytes, void* stream);

/* The reference's loss HELPERS as entry points of their own (callable names of SURVEY 8b; the training losses above fuse
 * them).  get_pos_and_neg_masks (nets/model.py:199-201): pos = labels == 1, neg = labels == 0 (label_rule 1: > 0 / not)
 * as bytes. */
int ocr_label_masks(const void* labels_f32, int64_t count, int label_rule, void* pos_u8, void* neg_u8, void* stream);
/* OHNM_single_image / OHNM_batch (nets/model.py:161-197) on GIVEN scores [n][hw] (P(negative class), >= 0) and byte masks:
 * per image k = min(n_pos * neg_ratio, #negatives), threshold = k-th smallest score among the negatives, selected_neg =
 * neg & score <= threshold (1.0 / 0.0; all zero when n_pos == 0), selected = float(pos) + selected_neg.  n_pos_i32 [n]
 * overrides the count of pos_mask (OHNM_single_image's argument; pos_mask may then be NULL); either output may be NULL. */
int ocr_ohnm_select(const void* scores, const void* pos_mask_u8, const void* neg_mask_u8, const void* n_pos_i32,
                    int n, int hw, float neg_ratio, void* selected_neg_f32, void* selected_f32, void* stream);
/* cal_link_loss (nets/model_vgg_16.py:227-241) for one direction: link_gt element r at [r * gt_stride], logit pair r at
 * [r * pred_stride .. +1] (tf.split slices of the 8- / 16-channel maps are strided rows), w_pixel [count] f32.
 * sums4 = (sum CE Wpos, sum Wpos, sum CE Wneg, sum Wneg), loss1 = s0/s1 + s2/s3 (unguarded like the reference). */
size_t ocr_link_ce_workspace(int64_t count);
int ocr_link_ce_fwd(const void* link_gt, int gt_stride, const void* link_pred, int pred_stride, const void* w_pixel,
                    int64_t count, void* sums4, void* loss1, void* workspace, size_t ws_bytes, void* stream);
int ocr_link_ce_bwd(const void* link_gt, int gt_stride, const void* link_pred, int pred_stride, const void* w_pixel,
                    int64_t count, const void* sums4, float grad_scale, void* d_link_pred, int d_stride, void* stream);

/* ------------------------------------------------------------------------- *
 * PixelLink decode (test_pixellink_fast.py:53-64,110-178; tool/pixellink_fn.py:120-158).
 * ------------------------------------------------------------------------- */
/* softmax over [pairs][2] (slim.softmax(pixel_cls), nets/pixellink.py:71) */
int ocr_softmax_pairs(const void* logits, int64_t pairs, void* probs, void* stream);
/* link_logits [m][16] -> [8][m][2]: tf.stack of the eight per-direction softmaxes */
int ocr_link_softmax_stack(const void* link_logits, int64_t m, void* out, void* stream);
/* pixel_detect: score_map [n,h,w,1], link_scores [8,n,h,w,2] -> uint8 [h,w] for batch element 0:
 * score > score_map_thresh and every link[i,0,y,x,1] >= link_thresh */
int ocr_pixel_detect(const void* score_map, const void* link_scores, int n, int h, int w,
                     float score_map_thresh, float link_thresh, void* mask_u8, void* stream);
/* Link-gated connected components, batched.  pixel_score [n,h,w]; link_score element
 * (d, img, y, x) at ((d*n+img)*h*w + y*w+x)*link_elem_stride + link_elem_offset.  Outputs: labels
 * int32 [n,h,w] (0 = background, dense ids 1..K in ascending order of each component's smallest
 * pixel index, only components with more than min_size pixels), ncomp [n], comps
 * [n][max_comps][2] = (smallest pixel index, size). */
size_t ocr_link_cc_workspace(int n, int h, int w);
int ocr_link_cc(const void* pixel_score, const void* link_score, int link_elem_stride,
                int link_elem_offset, int n, int h, int w, float pixel_thresh, float link_thresh,
                int min_size, void* labels_i32, void* ncomp_i32, void* comps_i32, int max_comps,
                void* workspace, size_t ws_bytes, void* stream);
/* The reference's own grouping rule, exactly (test_pixellink_fast.py:153-178): DIRECTED reachability from each
 * unassigned key, in the order `for i in graph.keys()` meets the keys, through unassigned pixels; a set gets a gid
 * only if it has more than min_size members, sets that fail stay 0 and can be collected again.  Refines
 * ocr_link_cc's weakly-connected components (directed sets never leave one): union_labels / union_ncomp are
 * ocr_link_cc's outputs for the SAME maps and thresholds; labels (a different buffer), ncomp, comps
 * [n][max_comps][2] = (seed pixel, size) as for ocr_link_cc, gids in the order the script meets its successful
 * seeds.  seed_order_i32 [n][h*w] (device): the keys (y*w + x of the interior segment pixels) in the script's
 * iteration order, padded with -1 — the Python-2 dict order, from ocr_py27_dict_order below; NULL = ascending
 * pixel index (the order rounds 1-3 of this build used).  Bit-exact against
 * oracle.ocr_oracle.link_cc_reference_dfs(key_order="py27" | "ascending").
 * Cost: ONE 1024-thread workgroup per image, rounds x sweeps of full-image scans: ~1 ms per 256 x 256 map, but
 * seconds per image at 720 x 1280 with many failing seeds — an exactness mode, opt-in, never the throughput path. */
size_t ocr_link_cc_directed_workspace(int n, int h, int w);
int ocr_link_cc_directed(const void* pixel_score, const void* link_score, int link_elem_stride,
                         int link_elem_offset, int n, int h, int w, float pixel_thresh, float link_thresh,
                         int min_size, const void* union_labels_i32, const void* union_ncomp_i32,
                         const void* seed_order_i32, void* labels_i32, void* ncomp_i32, void* comps_i32,
                         int max_comps, void* workspace, size_t ws_bytes, void* stream);
/* HOST routine (both pointers are HOST memory; no GPU work): the iteration order of the CPython-2.7 dict the script
 * fills with the keys y*w + x of the interior pixels with pixel_score_host[y*w + x] > pixel_thresh (the f32 comparison
 * the kernels make), x outer / y inner (test_pixellink_fast.py:111,119-150,171).  Writes the keys in that order to order_host[0 .. count), -1 beyond, up to
 * h*w entries; returns count (>= 0) or a negative status.  Sequential by nature (every probe depends on the table's
 * history): ~1 us per key. */
int ocr_py27_dict_order(const float* pixel_score_host, float pixel_thresh, int h, int w, int32_t* order_host);

/* ------------------------------------------------------------------------- *
 * Locality-aware NMS (EAST, Zhou et al. CVPR 2017, Algorithm 1).  ABSENT from the reference tree
 * (SURVEY.md D2); build-defined, pinned bit for bit against oracle/lanms_oracle.c.
 * boxes f32 [n_images][max_k][9] (x1,y1,..,x4,y4,score; row-major scan order), counts i32
 * [n_images].  Outputs: merged [n_images][max_k][9], n_merged [n_images], keep_idx
 * [n_images][max_k] (indices into merged, in keep order), n_keep [n_images].
 * ------------------------------------------------------------------------- */
size_t ocr_lanms_workspace(int n_images, int max_k);
int ocr_lanms(const void* boxes, const void* counts, int n_images, int max_k, float iou_thresh,
              void* merged, void* n_merged, void* keep_idx, void* n_keep, void* workspace,
              size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * One oriented box per component: the device side of
 *   rectangle = cv2.minAreaRect(show_xy); box = np.int0(cv2.boxPoints(rectangle))
 * (test_pixellink_fast.py:193-202, test_pixellink.py:207-216; cv2 = OpenCV 3.x convexHull +
 * rotatingCalipers, restated in oracle/cvgeom_oracle.c and matched bit for bit).
 * labels int32 [n][h][w] with ids 1..ncomp[img] (ocr_link_cc's output).  A component's points are
 * X = (int)(x*scale_x), Y = (int)(y*scale_y) (scale >= 1, h <= 1024).  Outputs, per
 * (image, id-1) in [n][max_comps]: hull_n = number of convex-hull vertices (0: id without pixels),
 * hull_head [4] = the first two hull vertices (x0,y0,x1,y1) in OpenCV's clockwise order,
 * calipers [6] = (corner x, corner y, edge-1 vector, edge-2 vector) of the minimum-area rectangle
 * (zeros when hull_n <= 2).  RotatedRect / boxPoints are O(1) per box on the host.
 * ------------------------------------------------------------------------- */
size_t ocr_min_area_rects_workspace(int n, int h, int w, int max_comps);
int ocr_min_area_rects(const void* labels_i32, const void* ncomp_i32, int n, int h, int w, int max_comps,
                       double scale_x, double scale_y, void* hull_n_i32, void* hull_head_i32,
                       void* calipers_f32, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * EAST-script decode (test.py:45-74,182-201): the `pixel_detect` twin, the regions cv2.findContours
 * (RETR_TREE) traces, and one oriented box per contour.
 * ------------------------------------------------------------------------- */
/* mask = score > score_thresh (score f32 [h][w]); first_second int32 [2][8] (pre-filled with INT_MAX)
 * receives, per link channel c, the raster indices of the first and the second pixel with
 * link16[.., 2c+1] < link_thresh — the only two `argwhere` rows test.py:70-72 uses. */
int ocr_east_pixel_detect(const void* score_f32, const void* link16_f32, int h, int w, float score_thresh,
                          float link_thresh, void* mask_u8, void* first_second_i32, void* stream);
int ocr_zero_pixels_u8(void* mask_u8, const void* idx_i32, int count, void* stream);
/* Parent links of cv2.findContours(mask, RETR_TREE, ...) (test.py:182), from the two ocr_mask_cc
 * labellings of one mask (labels/comps: 8-connected components of the 1-pixels; zlabels/zcomps:
 * 4-connected regions of the 0-pixels): parent_c[i] = 0-region left of component i's first pixel,
 * parent_z[j] = component left of region j's first pixel; 0 = column 0 (the frame).  The host derives
 * OpenCV's list order from them (newest sibling first, pre-order). */
int ocr_contour_parents(const void* labels_i32, const void* zlabels_i32, const void* comps_i32, int ncomp,
                        const void* zcomps_i32, int nregions, int h, int w, void* parent_c_i32,
                        void* parent_z_i32, void* stream);
/* Connected components of the pixels with (mask != 0) == (value != 0), connectivity 4 or 8; outputs as
 * ocr_link_cc (dense ids in ascending order of the smallest pixel index).  Workspace:
 * ocr_link_cc_workspace(n, h, w). */
int ocr_mask_cc(const void* mask_u8, int value, int connectivity, int n, int h, int w, void* labels_i32,
                void* ncomp_i32, void* comps_i32, int max_comps, void* workspace, size_t ws_bytes, void* stream);
/* Inner contours: zlabels = 4-connected components of the 0-pixels (ocr_mask_cc value 0).  A region
 * touching the image edge is background; any other is a hole whose contour is the set of 1-pixels
 * with a 4-neighbour in it.  Per region id: hull_n (0 for background regions), hull_head, calipers as
 * ocr_min_area_rects. */
size_t ocr_hole_border_rects_workspace(int n, int h, int w, int max_regions);
int ocr_hole_border_rects(const void* mask_u8, const void* zlabels_i32, const void* nregions_i32, int n, int h,
                          int w, int max_regions, double scale_x, double scale_y, void* hull_n_i32,
                          void* hull_head_i32, void* calipers_f32, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Ground-truth label maps: datasets/icdar.py:486-539 `generate_rbox` (+ the
 * generator's [::4,::4] subsample, :632-634) and tool/pixellink_fn.py:53-110 `generate_rbox`, both on
 * cv2.fillPoly rasters of the text polygons (OpenCV restated in oracle/cvgeom_oracle.c, matched
 * bit for bit).
 * polys int32 [n][max_polys][verts][2] (x,y as the reference casts them: .astype(np.int32)), counts
 * int32 [n], ignore uint8 [n][max_polys] (icdar: `tag or min(poly_h, poly_w) < min_text_size`).
 * cover uint32 [n][h][w]: bits 0-7 = 1 + smallest index of a polygon whose raster contains the pixel
 * (0: none), bits 8-15 = 1 + largest such index (= the reference's poly_mask), bit 16 = covered by an
 * ignored polygon.  max_polys <= 254 (poly_mask is uint8), 3 <= verts <= 8.
 * ------------------------------------------------------------------------- */
int ocr_poly_cover(const void* polys_i32, const void* counts_i32, const void* ignore_u8, int n, int max_polys,
                   int verts, int h, int w, void* cover_u32, void* stream);
/* icdar labels at every `step`-th pixel: score f32 [n][oh][ow][1], geo f32 [n][oh][ow][8] (channel
 * order and transposed directions of icdar.py:522-537 / valid_link :83-105, index -1 wraps),
 * training mask f32 [n][oh][ow][1]; oh = ceil(h/step).  h == w (the reference's border rule). */
int ocr_icdar_labels(const void* cover_u32, int n, int h, int w, int step, void* score_f32, void* geo_f32,
                     void* mask_f32, void* stream);
/* pixellink_fn labels: INTER_NEAREST resize of score / poly_mask to [new_h][new_w], link = neighbour
 * has the same poly_mask value (true directions), borders 1: score f32 [n][new_h][new_w], link f32
 * [n][new_h][new_w][8]. */
int ocr_pixellink_labels(const void* cover_u32, int n, int h, int w, int new_h, int new_w, void* score_f32,
                         void* link_f32, void* stream);

/* Evaluation (tool/bboxes.py:246-283 np_bboxes_jaccard, called from bboxes_matching :171-240): pixel
 * counts of the filled rasters (cv2.drawContours thickness=-1 = the fillPoly raster) of every
 * detection x ground-truth pair inside a mask_h x mask_w image: inter = |A and B|, union = |A or B|,
 * int32 [nd][ng].  dets int32 [nd][verts][2], gts int32 [ng][verts][2]; 3 <= verts <= 8. */
int ocr_quad_iou(const void* dets_i32, int nd, const void* gts_i32, int ng, int verts, int mask_h, int mask_w,
                 void* inter_i32, void* union_i32, void* stream);

/* Strided 3x3 convolutions of ResNet-v1 (nets/resnet_utils.py:85-122 conv2d_same with stride 2) as one
 * stride-1 2x2 convolution over a space-to-depth copy: xs[n][Y][X][(a*2+b)*c + ch] = x[n][2Y+a][2X+b][ch]
 * (h, w even; c a multiple of 8), weights w22 f32 [2][2][4c][k] from w33 f32 [3][3][c][k] (zero where a tap
 * has no source), the weight gradient mapped back tap by tap, the input gradient un-shuffled (optionally added
 * to an existing gradient). */
int ocr_space_to_depth_f16(const void* x, int n, int h, int w, int c, void* xs, void* stream);
int ocr_depth_to_space_f16(const void* xs, int n, int h, int w, int c, void* x, int accumulate, void* stream);
int ocr_weights_s2d_f32(const void* w33_f32, int cin, int cout, void* w22_f32, void* stream);
int ocr_weights_s2d_grad_f32(const void* dw22_f32, int cin, int cout, void* dw33_f32, void* stream);

/* cv2.resize(im, dsize=(dw, dh)) (default INTER_LINEAR, 8-bit fixed-point path) + astype(float32):
 * datasets/icdar.py:615,630.  src uint8 [H][W][cn] -> dst f32 [dh][dw][cn]. */
int ocr_resize_linear_u8(const void* src_u8, int H, int W, int cn, void* dst_f32, int dh, int dw,
                         void* stream);

/* cv2.resize(plane, (dw, dh), interpolation=cv2.INTER_CUBIC) on float32 score maps, the up-sampling
 * of the full-resolution decode (test_pixellink.py:97-98 `b_score * 255` then resize = pre_scale 255;
 * :108-109 resize then `* 255` = post_scale 255).  src f32 [planes][h][w] -> dst f32 [planes][dh][dw]. */
int ocr_resize_cubic_f32(const void* src_f32, int planes, int h, int w, void* dst_f32, int dh, int dw,
                         float pre_scale, float post_scale, void* stream);

/* ------------------------------------------------------------------------- *
 * Optimisers over the flat parameter buffer: elements [0, n_regularized) also get
 * the slim.l2_regularizer gradient weight_decay*w.  g is multiplied by
 * inv_loss_scale first.  ema (may be NULL) follows ExponentialMovingAverage.
 * multigpu_train.py:103-107,137-142; train_pixellink.py:243.
 * ------------------------------------------------------------------------- */
int ocr_adam_step(void* w, const void* g, void* m, void* v, void* ema, int64_t n,
                  int64_t n_regularized, float lr_t, float beta1, float beta2, float eps,
                  float weight_decay, float inv_loss_scale, float ema_decay, void* stream);
int ocr_momentum_step(void* w, const void* g, void* accum, void* ema, int64_t n,
                      int64_t n_regularized, float lr, float momentum, float weight_decay,
                      float inv_loss_scale, float ema_decay, void* stream);
/* out_f32[0] = scale * sum(x[i]^2), f64 accumulation in a fixed order (bitwise reproducible): the
 * REGULARIZATION_LOSSES term of `total_loss` (multigpu_train.py:36,183-184) is
 * weight_decay/2 * sum(w^2) over the regularised range (slim.l2_regularizer, nets/model.py:103).
 * x must be 16-byte aligned. */
size_t ocr_sum_squares_workspace(int64_t n);
int ocr_sum_squares_f32(const void* x, int64_t n, float scale, void* out_f32, void* workspace,
                        size_t ws_bytes, void* stream);
int ocr_scale_f32(void* x, int64_t n, float s, void* stream);
int ocr_fill_f32(void* x, int64_t n, float value, void* stream);   /* n 4-byte words */

/* ------------------------------------------------------------------------- *
 * Data-parallel exchange.  Replaces `average_gradients`
 * (multigpu_train.py:70-85: concat + reduce_mean of the per-tower gradients, called
 * from the tower loop :118-133) and `sum_gradients` (train_pixellink.py:179-194):
 * ONE in-place all-reduce(SUM) per contiguous bucket of the tower's flat f32 gradient
 * buffer; the mean's 1/world factor is folded into the optimiser's `inv_loss_scale`.
 *
 * The RCCL communicator is created and owned by the host layer (one process per GPU):
 * rank 0 calls ocr_comm_unique_id and hands the OCR_COMM_ID_BYTES bytes to every rank
 * through its own rendezvous (torch.distributed's store in this build); each rank then
 * calls ocr_comm_init_rank ON ITS DEVICE and keeps the opaque handle.  RCCL is bound at
 * first use (dlsym of the librccl already in the process, else dlopen): ocr_comm_available()
 * = 0 and every call below returns OCR_ERR_RCCL when there is none.
 * ocr_allreduce_bucket is asynchronous on `stream` like every other entry point; ordering
 * against the compute stream is the caller's, with ocr_event_record / ocr_stream_wait_event
 * (hipEventRecord / hipStreamWaitEvent on caller-visible handles), so a recorded step is a
 * flat list of C-ABI calls, exchange included.
 * ------------------------------------------------------------------------- */
#define OCR_COMM_ID_BYTES 128
enum { OCR_DT_F32 = 0, OCR_DT_F16 = 1, OCR_DT_BF16 = 2, OCR_DT_I32 = 3 };
enum { OCR_RED_SUM = 0, OCR_RED_MAX = 1 };
int ocr_comm_available(void);
const char* ocr_comm_last_error(void);
int ocr_comm_unique_id(void* id_out /* OCR_COMM_ID_BYTES host bytes */);
int ocr_comm_init_rank(void** comm_out, int nranks, const void* id, int rank);
int ocr_comm_size(void* comm);            /* number of ranks (>= 1), negative = error */
int ocr_comm_destroy(void* comm);
int ocr_allreduce_bucket(void* comm, void* buf, size_t count, int dtype, int op, void* stream);
int ocr_event_create(void** event_out);   /* timing disabled */
int ocr_event_destroy(void* event);
int ocr_event_record(void* event, void* stream);
int ocr_stream_wait_event(void* stream, void* event);
/* MEASUREMENT AID, not part of the exchange: a one-GPU stand-in for the device side of an N-rank ring all-reduce —
 * `workgroups` persistent 256-thread workgroups that read and re-write `bytes` of `buf` (values unchanged) in 256 KB
 * chunks paced at one xGMI link's rate (link_gbps; 2 x bytes through HBM in 2 * bytes / rate) — launched on the comm
 * stream where the step launches ocr_allreduce_bucket (bench.py `exchange.proxy`: what a comm kernel costs the conv
 * workgroups it shares the chip with).  stats_u64x8 (device; set ONCE by the caller to {~0, 0, 0, 0, 0, 0, 0, 0}; may be NULL): [3] += busy
 * ticks (100 MHz) of every launch, [4] += 1. */
int ocr_comm_proxy(void* buf, size_t bytes, int workgroups, float link_gbps, void* stats_u64x8, void* stream);
/* Footprint of the stand-in for the ocr_comm_proxy launches that follow (process-wide; a recorded step replays its calls
 * unchanged): fat = 0: 18 registers, no LDS (the default); fat = 1: ~128 VGPRs and 64 KB of LDS per workgroup, the order of
 * RCCL's generic all-reduce kernels — the two bracket what a real ring's device side costs the kernels it shares the chip
 * with.  workgroups > 0 overrides the launch's own count (RCCL runs 16-32 channels); 0 keeps it. */
int ocr_comm_proxy_set_footprint(int fat, int workgroups);

#ifdef __cplusplus
}
#endif
#endif /* OCR_HIP_H_ */
