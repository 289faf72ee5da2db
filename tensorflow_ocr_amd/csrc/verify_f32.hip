// f32 VERIFICATION precision of the forward trunk: the same graph (host code, padding rules, batch
// norm formulas, heads) run with f32 storage and f32 arithmetic, so that end-to-end score maps can
// be compared with the f32 oracle at the 1e-3 tolerance the north star states — f16 storage alone
// costs ~2^-11 per layer, ~1e-2 of the logit range after 16 layers.
//
// These kernels are deliberately plain (direct convolution on the vector ALUs, one thread per
// output element): they are a checking mode selected by Graph(precision="f32"), not the product
// path, and they have no backward.  Reference call sites are the same as the f16 kernels':
// slim.conv2d (nets/vgg.py:14-39), slim.batch_norm under resnet_arg_scope
// (nets/model_vgg_16.py:144), slim.max_pool2d (nets/vgg.py:16-32), mean_image_subtraction
// (nets/model_vgg_16.py:19-32).
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

// per-block partial (sum, sum of squares) per channel over a strip of pixels: [T][2][C] f32
__global__ void channel_stats_f32_kernel(const float* __restrict__ x, size_t npix, int C, size_t strip,
                                         float* __restrict__ partial) {
  const size_t p0 = (size_t)blockIdx.x * strip;
  size_t p1 = p0 + strip;
  if (p1 > npix) p1 = npix;
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0, q = 0.0;
    for (size_t px = p0; px < p1; ++px) {
      const double v = x[px * C + c];
      s += v;
      q += v * v;
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c] = (float)s;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = (float)q;
  }
}

// a = [relu](y*scale + shift); pool = 2: also the 2x2/2 SAME max-pool of a
__global__ void bn_relu_f32_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int n, int h, int w, int c, int relu,
                                   float* __restrict__ a_full, float* __restrict__ a_pool, int pool) {
  if (!pool) {
    const size_t total = (size_t)n * h * w * c;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
      const int ch = (int)(i % c);
      float v = y[i] * scale[ch] + shift[ch];
      if (relu && v < 0.f) v = 0.f;
      a_full[i] = v;
    }
    return;
  }
  const int oh = (h + 1) / 2, ow = (w + 1) / 2;
  const size_t total = (size_t)n * oh * ow * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t u = i / c;
    const int ox = (int)(u % ow);
    u /= ow;
    const int oy = (int)(u % oh);
    const int img = (int)(u / oh);
    float m = -INFINITY;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) {
        const int iy = 2 * oy + dy, ix = 2 * ox + dx;
        if (iy >= h || ix >= w) continue;
        const size_t j = (((size_t)img * h + iy) * w + ix) * c + ch;
        float v = y[j] * scale[ch] + shift[ch];
        if (relu && v < 0.f) v = 0.f;
        if (a_full) a_full[j] = v;
        m = v > m ? v : m;
      }
    a_pool[i] = m;
  }
}

__global__ void maxpool_f32_kernel(const float* __restrict__ x, int n, int h, int w, int c, int k, int stride,
                                   int pt, int pl, int oh, int ow, float* __restrict__ y) {
  const size_t total = (size_t)n * oh * ow * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t u = i / c;
    const int ox = (int)(u % ow);
    u /= ow;
    const int oy = (int)(u % oh);
    const int img = (int)(u / oh);
    float m = -INFINITY;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const int iy = oy * stride + ky - pt, ix = ox * stride + kx - pl;
        if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
        const float v = x[(((size_t)img * h + iy) * w + ix) * c + ch];
        m = v > m ? v : m;
      }
    y[i] = m;
  }
}

__global__ void prep_images_f32_kernel(const float* __restrict__ im, size_t npix, float m0, float m1, float m2,
                                       float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    out[3 * i + 0] = im[3 * i + 0] - m0;
    out[3 * i + 1] = im[3 * i + 1] - m1;
    out[3 * i + 2] = im[3 * i + 2] - m2;
  }
}

// out = relu(y*scale + shift + shortcut)   (bottleneck tail, nets/resnet_v1.py:104-111)
__global__ void bn_add_relu_f32_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                       const float* __restrict__ shift, const float* __restrict__ sc,
                                       size_t total, int c, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const float v = y[i] * scale[ch] + shift[ch] + sc[i];
    out[i] = v > 0.f ? v : 0.f;
  }
}

// tf.image.resize_bilinear x2, TF-1.4 legacy sampling (see unpool_f16_kernel in bn_pool.hip)
__global__ void unpool_f32_kernel(const float* __restrict__ x, int n, int lh, int lw, int c,
                                  float* __restrict__ y) {
  const int H = 2 * lh, W = 2 * lw;
  const size_t total = (size_t)n * H * W * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t u = i / c;
    const int ox = (int)(u % W);
    u /= W;
    const int oy = (int)(u % H);
    const int img = (int)(u / H);
    const int y0 = oy >> 1, x0 = ox >> 1;
    const int y1 = (oy & 1) ? (y0 + 1 < lh ? y0 + 1 : lh - 1) : y0;
    const int x1 = (ox & 1) ? (x0 + 1 < lw ? x0 + 1 : lw - 1) : x0;
    const float wy = (oy & 1) ? 0.5f : 0.f, wx = (ox & 1) ? 0.5f : 0.f;
    const float* b = x + (size_t)img * lh * lw * c + ch;
    const float v00 = b[((size_t)y0 * lw + x0) * c], v01 = b[((size_t)y0 * lw + x1) * c];
    const float v10 = b[((size_t)y1 * lw + x0) * c], v11 = b[((size_t)y1 * lw + x1) * c];
    const float top = v00 + (v01 - v00) * wx, bot = v10 + (v11 - v10) * wx;
    y[i] = top + (bot - top) * wy;
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

extern "C" int ocr_channel_stats_f32_num_partials(int64_t npix, int c) {
  if (npix <= 0 || c <= 0) return OCR_ERR_INVALID_ARG;
  int64_t t = (npix + 63) / 64;
  return (int)(t > 1024 ? 1024 : t);
}

extern "C" int ocr_channel_stats_f32(const void* x, int64_t npix, int c, void* partial, void* stream) {
  OCR_CHECK_ARG(x && partial);
  const int T = ocr_channel_stats_f32_num_partials(npix, c);
  if (T < 0) return T;
  const size_t strip = ((size_t)npix + T - 1) / T;
  hipLaunchKernelGGL(channel_stats_f32_kernel, dim3(T), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(x), (size_t)npix, c, strip, static_cast<float*>(partial));
  return ocr_launch_status();
}

extern "C" int ocr_bn_relu_f32(const void* y, const void* scale, const void* shift, int n, int h, int w, int c,
                               int relu, int pool, void* a_full, void* a_pool, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && n > 0 && h > 0 && w > 0 && c > 0);
  OCR_CHECK_ARG(pool == 0 || pool == 2);
  OCR_CHECK_ARG(pool ? a_pool != nullptr : a_full != nullptr);
  const size_t total = pool ? (size_t)n * ((h + 1) / 2) * ((w + 1) / 2) * c : (size_t)n * h * w * c;
  hipLaunchKernelGGL(bn_relu_f32_kernel, dim3(vgrid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(y), static_cast<const float*>(scale),
                     static_cast<const float*>(shift), n, h, w, c, relu, static_cast<float*>(a_full),
                     static_cast<float*>(a_pool), pool);
  return ocr_launch_status();
}

extern "C" int ocr_maxpool_f32(const void* x, int n, int h, int w, int c, int k, int stride, int pad_top,
                               int pad_left, int oh, int ow, void* y, void* stream) {
  OCR_CHECK_ARG(x && y && n > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0 && c > 0);
  hipLaunchKernelGGL(maxpool_f32_kernel, dim3(vgrid((size_t)n * oh * ow * c)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(x), n, h, w, c, k, stride,
                     pad_top, pad_left, oh, ow, static_cast<float*>(y));
  return ocr_launch_status();
}

extern "C" int ocr_prep_images_f32(const void* images, int64_t npix, float mean_r, float mean_g, float mean_b,
                                   void* out, void* stream) {
  OCR_CHECK_ARG(images && out && npix > 0);
  hipLaunchKernelGGL(prep_images_f32_kernel, dim3(vgrid((size_t)npix)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(images), (size_t)npix, mean_r,
                     mean_g, mean_b, static_cast<float*>(out));
  return ocr_launch_status();
}

extern "C" int ocr_bn_add_relu_f32(const void* y, const void* scale, const void* shift, const void* shortcut,
                                   int64_t npix, int c, void* out, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && shortcut && out && npix > 0 && c > 0);
  hipLaunchKernelGGL(bn_add_relu_f32_kernel, dim3(vgrid((size_t)npix * c)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(y),
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(shortcut), (size_t)npix * c, c, static_cast<float*>(out));
  return ocr_launch_status();
}

extern "C" int ocr_unpool_f32(const void* x, int n, int lh, int lw, int c, void* y, void* stream) {
  OCR_CHECK_ARG(x && y && n > 0 && lh > 0 && lw > 0 && c > 0);
  hipLaunchKernelGGL(unpool_f32_kernel, dim3(vgrid((size_t)n * lh * lw * 4 * c)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(x), n, lh, lw, c,
                     static_cast<float*>(y));
  return ocr_launch_status();
}
