// f32 INFERENCE PRECISION of the product library (Graph(precision="f32"); test.py / test_pixellink*.py --precision f32):
// the same forward graph (host code, padding rules, batch-norm formulas, heads) with f32 storage and f32 arithmetic, the
// convolutions on the MATRIX CORES — v_mfma_f32_32x32x2_f32: exact f32 FMAs, f32 accumulate, 1/16 of the 16-bit MFMA rate
// (MI355X_MICROARCH.md: 155 TFLOP/s measured) — so that score / link maps meet the north star's "within 1e-3 of the
// reference" on the product path: 16-bit storage alone costs ~2^-11 per layer, ~1e-2 of the logit range after 16 layers
// (DESIGN.md section 4).  Forward only.  Reference call sites as for the 16-bit kernels: slim.conv2d (nets/vgg.py:14-39,
// nets/resnet_v1.py:97-105), slim.batch_norm (nets/model_vgg_16.py:144), slim.max_pool2d (nets/vgg.py:16-32),
// mean_image_subtraction (nets/model_vgg_16.py:19-32), tf.image.resize_bilinear (nets/model.py:14-15).
#include "common.h"

namespace {

struct F32P {
  int n, h, w, cin, oh, ow, cout, kh, kw, stride, dil, pt, pl, flags;
  int P;                      // n * oh * ow output pixels
};

typedef float f32x16v __attribute__((ext_vector_type(16)));

// Implicit GEMM, M = cout (A = weights), N = output pixels (B = activations), K = taps x cin.  Workgroup = 4 waves = 128
// consecutive output pixels (flattened n, oy, ox) x 64 couts; wave = 32 pixels x 64 couts = two 32x32 accumulator blocks.
// Per (tap, 16-channel chunk): the 128 x 16 activation slice (zero where the tap falls outside the image) and the 16 x 64
// HWIO weight slice are staged in LDS; a k-step of the MFMA takes 2 channels: lane l holds A[cout = l % 32][k = l / 32] and
// B[k = l / 32][pixel = l % 32].  Activation rows are 17 floats apart (32 lanes x stride 17: conflict-free), weight rows 64.
constexpr int kFM = 128, kFN = 64, kFK = 16, kFXS = kFK + 1;

__global__ __launch_bounds__(256) void conv_f32_mfma_kernel(F32P p, const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y) {
  __shared__ float xs[kFM * kFXS];
  __shared__ __attribute__((aligned(16))) float ws[kFK * kFN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, hh = lane >> 5;
  const int p0 = blockIdx.x * kFM, co0 = blockIdx.y * kFN;
  // the two activation pieces (4 channels of one pixel) this thread stages per chunk
  int s_img[2], s_oy[2], s_ox[2];
  bool s_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int px = (tid + 256 * i) >> 2;
    int q = p0 + px;
    s_ok[i] = q < p.P;
    if (!s_ok[i]) q = 0;
    s_ox[i] = q % p.ow;
    q /= p.ow;
    s_oy[i] = q % p.oh;
    s_img[i] = q / p.oh;
  }
  const int wk = tid >> 4, wc4 = (tid & 15) * 4;           // this thread's weight piece: row k, couts wc4..wc4+3
  const bool vec_c = (p.cin & 3) == 0 && ((uintptr_t)x & 15) == 0;
  const bool vec_o = (p.cout & 3) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0;
  f32x16v acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  for (int tap = 0; tap < p.kh * p.kw; ++tap) {
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    for (int c0 = 0; c0 < p.cin; c0 += kFK) {
      float xv[2][4], wv[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = ((tid + 256 * i) & 3) * 4;
        const int iy = s_oy[i] * p.stride + ky * p.dil - p.pt, ix = s_ox[i] * p.stride + kx * p.dil - p.pl;
        const bool in = s_ok[i] && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
        const float* xp = x + (((size_t)s_img[i] * p.h + (in ? iy : 0)) * p.w + (in ? ix : 0)) * p.cin + c0 + c4;
        if (in && vec_c && c0 + c4 + 3 < p.cin) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xp);
          xv[i][0] = v[0]; xv[i][1] = v[1]; xv[i][2] = v[2]; xv[i][3] = v[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[i][e] = (in && c0 + c4 + e < p.cin) ? xp[e] : 0.f;
        }
      }
      {
        const int ci = c0 + wk, co = co0 + wc4;
        const float* wp = w + ((size_t)tap * p.cin + (ci < p.cin ? ci : 0)) * p.cout + co;
        if (ci < p.cin && vec_o && co + 3 < p.cout) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(wp);
          wv[0] = v[0]; wv[1] = v[1]; wv[2] = v[2]; wv[3] = v[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[e] = (ci < p.cin && co + e < p.cout) ? wp[e] : 0.f;
        }
      }
      __syncthreads();                                     // the previous chunk's fragments have been read
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        float* d = xs + (idx >> 2) * kFXS + (idx & 3) * 4;
        d[0] = xv[i][0]; d[1] = xv[i][1]; d[2] = xv[i][2]; d[3] = xv[i][3];
      }
      *reinterpret_cast<f32x4*>(ws + wk * kFN + wc4) = f32x4{wv[0], wv[1], wv[2], wv[3]};
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < kFK / 2; ++kk) {
        const float b = xs[(wave * 32 + l32) * kFXS + kk * 2 + hh];
        const float a0 = ws[(kk * 2 + hh) * kFN + l32], a1 = ws[(kk * 2 + hh) * kFN + 32 + l32];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
      }
    }
  }
  // D block: lane (l32, hh), register e: cout = 32 j + (e & 3) + 8 (e >> 2) + 4 hh, pixel = l32
  const int q = p0 + wave * 32 + l32;
  if (q >= p.P) return;
  float* yp = y + (size_t)q * p.cout;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int co = co0 + 32 * j + 8 * g4 + 4 * hh;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[j][g4 * 4 + e];
        if (co + e < p.cout) {
          if (p.flags & OCR_CONV_BIAS) v[e] += bias[co + e];
          if ((p.flags & OCR_CONV_RELU) && v[e] < 0.f) v[e] = 0.f;
          if (p.flags & OCR_CONV_ACCUM_F16) v[e] += yp[co + e];        // accumulate into y (concat-free 1x1 convs)
        }
      }
      if (vec_o && co + 3 < p.cout) {
        *reinterpret_cast<f32x4*>(yp + co) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < p.cout) yp[co + e] = v[e];
      }
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

// flags: OCR_CONV_BIAS, OCR_CONV_RELU, OCR_CONV_ACCUM_F16 (here: y += conv, f32).  x f32 NHWC, w f32 HWIO (the TF master
// layout, no packing), y f32 NHWC.
extern "C" int ocr_conv2d_f32_mfma(const ocr_conv_desc* d, const void* x, const void* w_hwio, const void* bias, void* y,
                                   void* stream) {
  OCR_CHECK_ARG(d && x && w_hwio && y);
  OCR_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->oh > 0 && d->ow > 0 && d->cin > 0 && d->cout > 0);
  OCR_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0);
  OCR_CHECK_ARG(!(d->flags & OCR_CONV_BIAS) || bias);
  OCR_CHECK_SHAPE((size_t)d->n * d->oh * d->ow < (1ull << 31));
  F32P p{d->n, d->h, d->w, d->cin, d->oh, d->ow, d->cout, d->kh, d->kw, d->stride, d->dilation, d->pad_top, d->pad_left,
         d->flags, d->n * d->oh * d->ow};
  hipLaunchKernelGGL(conv_f32_mfma_kernel, dim3(ocr_cdiv(p.P, kFM), ocr_cdiv(p.cout, kFN)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const float*>(x), static_cast<const float*>(w_hwio),
                     static_cast<const float*>(bias), static_cast<float*>(y));
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
