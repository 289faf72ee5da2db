/*
 * ocr_verify.h — C ABI of libocr_verify.so: TEST INFRASTRUCTURE, not product.
 *
 * A plain direct f32 convolution (tensorflow_ocr_amd/csrc/verify_f32.hip) against which tests hold the product
 * library's matrix-core f32 convolution (ocr_conv2d_f32_mfma), and with which the f32 precision can be re-run
 * end to end (Graph(precision="f32", f32_conv="direct") / OCR_F32_CONV=direct).  The product library
 * libocr_hip.so exports none of this header; the product path never loads this library.  Conventions as in ocr_hip.h.
 */
#ifndef OCR_VERIFY_H_
#define OCR_VERIFY_H_

#include "ocr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- *
 * The independent CHECKER of the product library's f32 precision (ocr_hip.h: ocr_conv2d_f32_mfma + the
 * element-wise ocr_*_f32 kernels, Graph(precision="f32")): a plain direct convolution, one thread per output
 * element, fmaf chain in (ky, kx, ci) order.  Tensors NHWC f32; weights in the TF HWIO master layout.
 * flags: OCR_CONV_BIAS, OCR_CONV_RELU, OCR_CONV_ACCUM_F16 (here: y += conv, f32).
 * ------------------------------------------------------------------------- */
int ocr_conv2d_f32(const ocr_conv_desc* d, const void* x, const void* w_hwio, const void* bias, void* y,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCR_VERIFY_H_ */
