// Input-image preparation of the training generator:
//     im = cv2.resize(im, dsize=(resize_w, resize_h)); images.append(im[:, :, ::-1].astype(np.float32))
// (datasets/icdar.py:615,630): uint8 [H][W][cn] of any size -> float32 [dh][dw][cn].
//
// cv2.resize's default INTER_LINEAR on 8-bit images is fixed point: 11-bit coefficients
// (cvRound of the float weights at half-pixel centres), a horizontal pass into int, and the vertical
// pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2; an exact 2x2 downscale is rerouted to
// INTER_AREA ((a+b+c+d+2)>>2).  Restated in oracle/cvgeom_oracle.c and matched bit for bit (built
// with -ffp-contract=off: the coordinate maths must round like the scalar original).  One thread
// per output element; HBM-bound: 4 source bytes read (L2-resident neighbours) + 4 bytes written.
#include <float.h>
#include "common.h"

namespace {

struct ResizeP {
  int H, W, cn, dh, dw, area2;
  double scale_x, scale_y;
};

__device__ __forceinline__ int coef(float v) {       // saturate_cast<short>(v) = cvRound, clamped
  int r = __float2int_rn(v);
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

__global__ void resize_linear_u8_kernel(ResizeP p, const unsigned char* __restrict__ src,
                                        float* __restrict__ dst) {
  const size_t total = (size_t)p.dh * p.dw * p.cn;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % p.cn);
    const int dx = (int)((i / p.cn) % p.dw), dy = (int)(i / ((size_t)p.cn * p.dw));
    int v;
    if (p.area2) {
      const unsigned char* s0 = src + ((size_t)(2 * dy) * p.W + 2 * dx) * p.cn + k;
      const unsigned char* s1 = s0 + (size_t)p.W * p.cn;
      v = (s0[0] + s0[p.cn] + s1[0] + s1[p.cn] + 2) >> 2;
    } else {
      float fx = (float)((dx + 0.5) * p.scale_x - 0.5);
      int sx = (int)floorf(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= p.W - 1) { fx = 0; sx = p.W - 1; }
      const int a0 = coef((1.f - fx) * 2048), a1 = coef(fx * 2048);
      float fy = (float)((dy + 0.5) * p.scale_y - 0.5);
      const int sy = (int)floorf(fy);
      fy -= sy;
      const int b0 = coef((1.f - fy) * 2048), b1 = coef(fy * 2048);
      const int r0 = min(max(sy, 0), p.H - 1), r1 = min(max(sy + 1, 0), p.H - 1);
      const int sx1 = sx + 1 < p.W ? sx + 1 : sx;
      const unsigned char* S0 = src + (size_t)r0 * p.W * p.cn;
      const unsigned char* S1 = src + (size_t)r1 * p.W * p.cn;
      const int h0 = S0[sx * p.cn + k] * a0 + S0[sx1 * p.cn + k] * a1;
      const int h1 = S1[sx * p.cn + k] * a0 + S1[sx1 * p.cn + k] * a1;
      v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
    dst[i] = (float)v;
  }
}

// cv2.resize(float32 plane, dsize, interpolation=cv2.INTER_CUBIC) — the score-map up-sampling of the
// full-resolution decode (test_pixellink.py:97-98,108-109).  OpenCV's float path: 4 taps per axis,
// Keys' kernel with A = -0.75 evaluated in float at the half-pixel-centre source coordinate
// (interpolateCubic), tap columns / rows clamped to the image, the horizontal sums
// S0*a0 + S1*a1 + S2*a2 + S3*a3 formed first (left to right, no contraction) and the vertical
// combination of the four row sums second.  `pre` multiplies the source samples (b_score * 255 before
// the resize), `post` the result (pixel_score * 255 after it).
struct CubicP {
  int planes, h, w, dh, dw;
  float pre, post;
  double scale_x, scale_y;
};

__device__ __forceinline__ void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

__global__ void resize_cubic_f32_kernel(CubicP p, const float* __restrict__ src, float* __restrict__ dst) {
  const size_t per = (size_t)p.dh * p.dw, total = per * p.planes;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pl = (int)(i / per);
    const int dy = (int)((i % per) / p.dw), dx = (int)(i % p.dw);
    float fx = (float)((dx + 0.5) * p.scale_x - 0.5);
    const int sx = (int)floorf(fx);
    fx -= sx;
    float fy = (float)((dy + 0.5) * p.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    float a[4], b[4];
    cubic_coeffs(fx, a);
    cubic_coeffs(fy, b);
    int xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xs[j] = min(max(sx - 1 + j, 0), p.w - 1);
    const float* S = src + (size_t)pl * p.h * p.w;
    float rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* R = S + (size_t)min(max(sy - 1 + k, 0), p.h - 1) * p.w;
      rows[k] = (R[xs[0]] * p.pre) * a[0] + (R[xs[1]] * p.pre) * a[1] + (R[xs[2]] * p.pre) * a[2] +
                (R[xs[3]] * p.pre) * a[3];
    }
    dst[i] = (rows[0] * b[0] + rows[1] * b[1] + rows[2] * b[2] + rows[3] * b[3]) * p.post;
  }
}

}  // namespace

extern "C" int ocr_resize_cubic_f32(const void* src_f32, int planes, int h, int w, void* dst_f32, int dh, int dw,
                                    float pre_scale, float post_scale, void* stream) {
  OCR_CHECK_ARG(src_f32 && dst_f32 && planes > 0 && h > 0 && w > 0 && dh > 0 && dw > 0);
  CubicP p;
  p.planes = planes; p.h = h; p.w = w; p.dh = dh; p.dw = dw;
  p.pre = pre_scale; p.post = post_scale;
  p.scale_x = 1. / ((double)dw / w);
  p.scale_y = 1. / ((double)dh / h);
  const size_t total = (size_t)planes * dh * dw;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(resize_cubic_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const float*>(src_f32),
                     static_cast<float*>(dst_f32));
  return ocr_launch_status();
}

extern "C" int ocr_resize_linear_u8(const void* src_u8, int H, int W, int cn, void* dst_f32, int dh, int dw,
                                    void* stream) {
  OCR_CHECK_ARG(src_u8 && dst_f32 && H > 0 && W > 0 && cn > 0 && dh > 0 && dw > 0);
  ResizeP p;
  p.H = H; p.W = W; p.cn = cn; p.dh = dh; p.dw = dw;
  const double inv_scale_x = (double)dw / W, inv_scale_y = (double)dh / H;
  p.scale_x = 1. / inv_scale_x;
  p.scale_y = 1. / inv_scale_y;
  const int iscale_x = (int)lrint(p.scale_x), iscale_y = (int)lrint(p.scale_y);
  p.area2 = fabs(p.scale_x - iscale_x) < DBL_EPSILON && fabs(p.scale_y - iscale_y) < DBL_EPSILON &&
            iscale_x == 2 && iscale_y == 2;
  const size_t total = (size_t)dh * dw * cn;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(resize_linear_u8_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const unsigned char*>(src_u8),
                     static_cast<float*>(dst_f32));
  return ocr_launch_status();
}
