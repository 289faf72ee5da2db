// PLAIN f32 direct convolution: the independent CHECKER of the f32 precision (include/ocr_verify.h; test infrastructure,
// libocr_verify.so).  One thread per output element on the vector ALUs, fmaf chain in (ky, kx, ci) order.  The f32
// precision itself — Graph(precision="f32"): v_mfma_f32_32x32x2_f32 convolution + the element-wise f32 kernels — lives in
// the product library (csrc/f32_infer.hip); tests hold its convolution against this one.
#include "common.h"
#include "../../include/ocr_verify.h"

namespace {

struct CvP {
  int n, h, w, cin, oh, ow, cout, kh, kw, stride, dil, pt, pl, flags;
};

// y[n,oy,ox,co] = sum_{ky,kx,ci} x[n, oy*s+ky*d-pt, ox*s+kx*d-pl, ci] * w[ky,kx,ci,co]   (HWIO weights)
__global__ void conv_f32_kernel(CvP p, const float* __restrict__ x, const float* __restrict__ w,
                                const float* __restrict__ bias, float* __restrict__ y) {
  const size_t total = (size_t)p.n * p.oh * p.ow * p.cout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int co = (int)(i % p.cout);
    size_t u = i / p.cout;
    const int ox = (int)(u % p.ow);
    u /= p.ow;
    const int oy = (int)(u % p.oh);
    const int img = (int)(u / p.oh);
    float acc = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
      const int iy = oy * p.stride + ky * p.dil - p.pt;
      if (iy < 0 || iy >= p.h) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ix = ox * p.stride + kx * p.dil - p.pl;
        if (ix < 0 || ix >= p.w) continue;
        const float* xp = x + (((size_t)img * p.h + iy) * p.w + ix) * p.cin;
        const float* wp = w + ((size_t)(ky * p.kw + kx) * p.cin) * p.cout + co;
        for (int ci = 0; ci < p.cin; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * p.cout], acc);
      }
    }
    if (p.flags & OCR_CONV_BIAS) acc += bias[co];
    if ((p.flags & OCR_CONV_RELU) && acc < 0.f) acc = 0.f;
    if (p.flags & OCR_CONV_ACCUM_F16) acc += y[i];     // accumulate into y (concat-free 1x1 convs)
    y[i] = acc;
  }
}

unsigned vgrid(size_t work) {
  size_t b = (work + 255) / 256;
  if (b > 65536) b = 65536;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int ocr_conv2d_f32(const ocr_conv_desc* d, const void* x, const void* w_hwio, const void* bias,
                              void* y, void* stream) {
  OCR_CHECK_ARG(d && x && w_hwio && y);
  OCR_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->oh > 0 && d->ow > 0 && d->cin > 0 && d->cout > 0);
  OCR_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0);
  OCR_CHECK_ARG(!(d->flags & OCR_CONV_BIAS) || bias);
  OCR_CHECK_ARG(!(d->flags & OCR_CONV_STATS) && !d->flip_taps);
  CvP p{d->n, d->h, d->w, d->cin, d->oh, d->ow, d->cout, d->kh, d->kw, d->stride, d->dilation,
        d->pad_top, d->pad_left, d->flags};
  hipLaunchKernelGGL(conv_f32_kernel, dim3(vgrid((size_t)d->n * d->oh * d->ow * d->cout)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const float*>(x),
                     static_cast<const float*>(w_hwio), static_cast<const float*>(bias), static_cast<float*>(y));
  return ocr_launch_status();
}

