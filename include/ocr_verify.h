/*
 * ocr_verify.h — C ABI of libocr_verify.so: TEST INFRASTRUCTURE, not product.
 *
 * The f32 VERIFICATION precision of the forward graph (tensorflow_ocr_amd/csrc/verify_f32.hip): plain
 * direct-convolution f32 kernels with which whole-graph outputs are checked against the f32 CPU
 * oracle at the north star's 1e-3 (tests/test_gpu_f32_verify.py, Graph(precision="f32")).  The
 * product library libocr_hip.so exports none of these; the product path never loads this library
 * (layers_f32.py does, on request).  Conventions as in ocr_hip.h.
 */
#ifndef OCR_VERIFY_H_
#define OCR_VERIFY_H_

#include "ocr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- *
 * f32 VERIFICATION precision of the forward trunk (csrc/verify_f32.hip): the same graph with f32
 * storage and arithmetic, plain direct kernels, no backward.  Selected by Graph(precision="f32");
 * exists so that end-to-end outputs can be checked against the f32 oracle at the north star's 1e-3.
 * Tensors NHWC f32; conv weights in the TF HWIO master layout [kh][kw][cin][cout] (no packing).
 * ------------------------------------------------------------------------- */
/* flags: OCR_CONV_BIAS, OCR_CONV_RELU, OCR_CONV_ACCUM_F16 (here: y += conv, f32) */
int ocr_conv2d_f32(const ocr_conv_desc* d, const void* x, const void* w_hwio, const void* bias, void* y,
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

#ifdef __cplusplus
}
#endif
#endif /* OCR_VERIFY_H_ */
