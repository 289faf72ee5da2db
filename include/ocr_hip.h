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

#define OCR_ABI_VERSION 1

enum {
  OCR_OK = 0,
  OCR_ERR_INVALID_ARG = -1,
  OCR_ERR_UNSUPPORTED = -2,
  OCR_ERR_HIP = -3,
  OCR_ERR_WORKSPACE = -4
};

int ocr_abi_version(void);
const char* ocr_status_string(int status);

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

/* First-layer weight gradient (cin=3 from the [n,h,w,4] f16 image). dw [3,3,3,cout] f32. */
int ocr_conv2d_first_wgrad_f16(int n, int h, int w, int cout, const void* x4,
                               const void* dy, void* dw_hwio_f32,
                               void* workspace, size_t ws_bytes, void* stream);
size_t ocr_conv2d_first_wgrad_workspace(int n, int h, int w, int cout);

#ifdef __cplusplus
}
#endif
#endif /* OCR_HIP_H_ */
