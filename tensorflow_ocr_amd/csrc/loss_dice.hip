// Dice loss of nets/model_vgg_16.py:179-225 (same function at nets/model.py:145-159):
//   dice(y, p, m) = 1 - 2*sum(y*p*m) / (sum(y*m) + sum(p*m) + 1e-5)
//   loss = 2*dice(pixel) + sum_{i<8} dice(link_i)
// with TensorFlow broadcasting: the label / mask tensors have one channel, the
// prediction may have `pc` (pixel) or `G` (per link direction; tf.split(..., 8,
// axis=3)) channels, sum(y*m) is taken on the un-broadcast tensor.
//
// One streaming pass produces all 27 sums (wavefront shuffles -> one partial row
// per workgroup, summed in f64 by a single wave: no atomics, reproducible); the
// backward pass is the analytic elementwise gradient.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct DiceP {
  int P, pc, G;
};

__global__ __launch_bounds__(256) void dice_reduce_kernel(DiceP d, const float* __restrict__ ytp,
                                                          const float* __restrict__ ypp,
                                                          const float* __restrict__ ytl,
                                                          const float* __restrict__ ypl,
                                                          const float* __restrict__ mask,
                                                          float* __restrict__ partial) {
  __shared__ float red[4][27];
  float s[27];
#pragma unroll
  for (int j = 0; j < 27; ++j) s[j] = 0.f;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)d.P; p += (size_t)gridDim.x * 256) {
    const float m = mask[p];
    {
      const float y = ytp[p];
      float ps = 0.f;
      for (int j = 0; j < d.pc; ++j) ps += ypp[p * d.pc + j];
      s[0] += y * ps * m;
      s[1] += y * m;
      s[2] += ps * m;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float y = ytl[p * 8 + i];
      float ps = 0.f;
      for (int j = 0; j < d.G; ++j) ps += ypl[p * 8 * d.G + i * d.G + j];
      s[3 + 3 * i] += y * ps * m;
      s[4 + 3 * i] += y * m;
      s[5 + 3 * i] += ps * m;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 27; ++j) {
    float v = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    partial[(size_t)blockIdx.x * 27 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// pc = G = 2 (every net of the reference: 2 pixel logits, 8 x 2 link logits): whole 16-byte vectors per pixel — 4 of
// link predictions, 2 of link labels — instead of 27 scalar loads at run-time strides; same sums in the same order.
__global__ __launch_bounds__(256) void dice_reduce22_kernel(DiceP d, const float* __restrict__ ytp,
                                                            const float* __restrict__ ypp,
                                                            const float* __restrict__ ytl,
                                                            const float* __restrict__ ypl,
                                                            const float* __restrict__ mask,
                                                            float* __restrict__ partial) {
  __shared__ float red[4][27];
  float s[27];
#pragma unroll
  for (int j = 0; j < 27; ++j) s[j] = 0.f;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)d.P; p += (size_t)gridDim.x * 256) {
    const float m = mask[p];
    const float y0 = ytp[p];
    const f32x2 pp = *reinterpret_cast<const f32x2*>(ypp + p * 2);
    f32x4 yl[2], pl[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) yl[k] = *reinterpret_cast<const f32x4*>(ytl + p * 8 + k * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) pl[k] = *reinterpret_cast<const f32x4*>(ypl + p * 16 + k * 4);
    {
      const float ps = 0.f + pp[0] + pp[1];
      s[0] += y0 * ps * m;
      s[1] += y0 * m;
      s[2] += ps * m;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float y = yl[i >> 2][i & 3];
      const float ps = 0.f + pl[i >> 1][(i & 1) * 2] + pl[i >> 1][(i & 1) * 2 + 1];
      s[3 + 3 * i] += y * ps * m;
      s[4 + 3 * i] += y * m;
      s[5 + 3 * i] += ps * m;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 27; ++j) {
    float v = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    partial[(size_t)blockIdx.x * 27 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// sums[27] (I, A, B per map), loss[0] = total, loss[1..9] = the nine dice terms
__global__ __launch_bounds__(1024) void dice_finalize_kernel(const float* __restrict__ partial, int T,
                                                             float* __restrict__ sums, float* __restrict__ loss) {
  __shared__ double part[32][32];
  __shared__ double tot[27];
  const int j = threadIdx.x & 31, g = threadIdx.x >> 5;   // 1024 threads: 32 row groups x 32 cols
  double a = 0.0;
  if (j < 27)
    for (int t0 = g; t0 < T; t0 += 32 * 8) {              // rows g, g + 32, ...: eight loads in flight, added in row order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int t = t0 + 32 * k;
        const float x = partial[(size_t)(t < T ? t : 0) * 27 + j];
        v[k] = t < T ? x : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) a += (double)v[k];
    }
  part[g][j] = a;
  __syncthreads();
  if (threadIdx.x < 27) {
    double s = 0.0;
    for (int k = 0; k < 32; ++k) s += part[k][threadIdx.x];
    tot[threadIdx.x] = s;
    sums[threadIdx.x] = (float)s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double total = 0.0;
    for (int k = 0; k < 9; ++k) {
      const double I = tot[3 * k], U = tot[3 * k + 1] + tot[3 * k + 2] + 1e-5;
      const double dl = 1.0 - 2.0 * I / U;
      loss[1 + k] = (float)dl;
      total += (k == 0 ? 2.0 : 1.0) * dl;
    }
    loss[0] = (float)total;
  }
}

// pc = G = 1 (EAST's sigmoid heads, nets/model_vgg_16.py:129-131: one score map, 8 geometry maps): the link labels and
// predictions of a pixel are two 16-byte vectors each; same sums in the same order as the generic kernel.
__global__ __launch_bounds__(256) void dice_reduce11_kernel(DiceP d, const float* __restrict__ ytp,
                                                            const float* __restrict__ ypp,
                                                            const float* __restrict__ ytl,
                                                            const float* __restrict__ ypl,
                                                            const float* __restrict__ mask,
                                                            float* __restrict__ partial) {
  __shared__ float red[4][27];
  float s[27];
#pragma unroll
  for (int j = 0; j < 27; ++j) s[j] = 0.f;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)d.P; p += (size_t)gridDim.x * 256) {
    const float m = mask[p];
    const float y0 = ytp[p];
    const float p0 = ypp[p];
    f32x4 yl[2], pl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      yl[k] = *reinterpret_cast<const f32x4*>(ytl + p * 8 + k * 4);
      pl[k] = *reinterpret_cast<const f32x4*>(ypl + p * 8 + k * 4);
    }
    {
      const float ps = 0.f + p0;
      s[0] += y0 * ps * m;
      s[1] += y0 * m;
      s[2] += ps * m;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float y = yl[i >> 2][i & 3];
      const float ps = 0.f + pl[i >> 2][i & 3];
      s[3 + 3 * i] += y * ps * m;
      s[4 + 3 * i] += y * m;
      s[5 + 3 * i] += ps * m;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 27; ++j) {
    float v = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    partial[(size_t)blockIdx.x * 27 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void dice_bwd11_kernel(DiceP d, const float* __restrict__ ytp, const float* __restrict__ ytl,
                                  const float* __restrict__ mask, const float* __restrict__ sums, float gscale,
                                  float* __restrict__ dpp, float* __restrict__ dpl) {
  __shared__ float cI[9], cU[9];
  if (threadIdx.x < 9) {
    const int k = threadIdx.x;
    const float I = sums[3 * k], U = sums[3 * k + 1] + sums[3 * k + 2] + 1e-5f;
    const float wk = (k == 0 ? 2.f : 1.f) * gscale;
    cU[k] = -2.f * wk / U;
    cI[k] = 2.f * wk * I / (U * U);
  }
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)d.P; p += (size_t)gridDim.x * 256) {
    const float m = mask[p];
    dpp[p] = m * (ytp[p] * cU[0] + cI[0]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x4 yl = *reinterpret_cast<const f32x4*>(ytl + p * 8 + k * 4);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = m * (yl[e] * cU[1 + 4 * k + e] + cI[1 + 4 * k + e]);
      *reinterpret_cast<f32x4*>(dpl + p * 8 + k * 4) = o;
    }
  }
}

__global__ void dice_bwd_kernel(DiceP d, const float* __restrict__ ytp,
                                const float* __restrict__ ytl, const float* __restrict__ mask,
                                const float* __restrict__ sums, float gscale,
                                float* __restrict__ dpp, float* __restrict__ dpl) {
  __shared__ float cI[9], cU[9];
  if (threadIdx.x < 9) {
    const int k = threadIdx.x;
    const float I = sums[3 * k], U = sums[3 * k + 1] + sums[3 * k + 2] + 1e-5f;
    const float wk = (k == 0 ? 2.f : 1.f) * gscale;
    // d/dp [1 - 2I/U] = -2 (y m U - I m) / U^2 = m * (-2 y / U + 2 I / U^2)
    cU[k] = -2.f * wk / U;
    cI[k] = 2.f * wk * I / (U * U);
  }
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)d.P; p += (size_t)gridDim.x * 256) {
    const float m = mask[p];
    const float gp = m * (ytp[p] * cU[0] + cI[0]);
    for (int j = 0; j < d.pc; ++j) dpp[p * d.pc + j] = gp;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float gl = m * (ytl[p * 8 + i] * cU[1 + i] + cI[1 + i]);
      for (int j = 0; j < d.G; ++j) dpl[p * 8 * d.G + i * d.G + j] = gl;
    }
  }
}

__global__ void dice_bwd22_kernel(DiceP d, const float* __restrict__ ytp, const float* __restrict__ ytl,
                                  const float* __restrict__ mask, const float* __restrict__ sums, float gscale,
                                  float* __restrict__ dpp, float* __restrict__ dpl) {
  __shared__ float cI[9], cU[9];
  if (threadIdx.x < 9) {
    const int k = threadIdx.x;
    const float I = sums[3 * k], U = sums[3 * k + 1] + sums[3 * k + 2] + 1e-5f;
    const float wk = (k == 0 ? 2.f : 1.f) * gscale;
    cU[k] = -2.f * wk / U;
    cI[k] = 2.f * wk * I / (U * U);
  }
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)d.P; p += (size_t)gridDim.x * 256) {
    const float m = mask[p];
    const float gp = m * (ytp[p] * cU[0] + cI[0]);
    *reinterpret_cast<f32x2*>(dpp + p * 2) = f32x2{gp, gp};
    f32x4 yl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) yl[k] = *reinterpret_cast<const f32x4*>(ytl + p * 8 + k * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g0 = m * (yl[k >> 1][(k & 1) * 2] * cU[1 + 2 * k] + cI[1 + 2 * k]);
      const float g1 = m * (yl[k >> 1][(k & 1) * 2 + 1] * cU[2 + 2 * k] + cI[2 + 2 * k]);
      *reinterpret_cast<f32x4*>(dpl + p * 16 + k * 4) = f32x4{g0, g0, g1, g1};
    }
  }
}

int dice_blocks(int P) {
  int b = ocr_cdiv(P, 256 * 4);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return b;
}

}  // namespace

extern "C" size_t ocr_dice_workspace(int P) { return (size_t)dice_blocks(P) * 27 * sizeof(float); }

extern "C" int ocr_dice_loss_fwd(const void* y_true_pixel, const void* y_pred_pixel, int pc,
                                 const void* y_true_link, const void* y_pred_link, int G,
                                 const void* training_mask, int P, void* sums27, void* loss10,
                                 void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(y_true_pixel && y_pred_pixel && y_true_link && y_pred_link && training_mask);
  OCR_CHECK_ARG(sums27 && loss10 && workspace && P > 0 && pc >= 1 && G >= 1);
  if (ws_bytes < ocr_dice_workspace(P)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DiceP d{P, pc, G};
  const int T = dice_blocks(P);
  hipLaunchKernelGGL(pc == 2 && G == 2 ? dice_reduce22_kernel : pc == 1 && G == 1 ? dice_reduce11_kernel : dice_reduce_kernel,
                     dim3(T), dim3(256), 0, st, d,
                     static_cast<const float*>(y_true_pixel), static_cast<const float*>(y_pred_pixel),
                     static_cast<const float*>(y_true_link), static_cast<const float*>(y_pred_link),
                     static_cast<const float*>(training_mask), static_cast<float*>(workspace));
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(1024), 0, st,
                     static_cast<const float*>(workspace), T, static_cast<float*>(sums27),
                     static_cast<float*>(loss10));
  return ocr_launch_status();
}

extern "C" int ocr_dice_loss_bwd(const void* y_true_pixel, int pc, const void* y_true_link, int G,
                                 const void* training_mask, int P, const void* sums27,
                                 float grad_scale, void* d_pred_pixel, void* d_pred_link,
                                 void* stream) {
  OCR_CHECK_ARG(y_true_pixel && y_true_link && training_mask && sums27 && d_pred_pixel && d_pred_link);
  OCR_CHECK_ARG(P > 0 && pc >= 1 && G >= 1);
  DiceP d{P, pc, G};
  hipLaunchKernelGGL(pc == 2 && G == 2 ? dice_bwd22_kernel : pc == 1 && G == 1 ? dice_bwd11_kernel : dice_bwd_kernel,
                     dim3(dice_blocks(P) * 2), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d, static_cast<const float*>(y_true_pixel),
                     static_cast<const float*>(y_true_link), static_cast<const float*>(training_mask),
                     static_cast<const float*>(sums27), grad_scale,
                     static_cast<float*>(d_pred_pixel), static_cast<float*>(d_pred_link));
  return ocr_launch_status();
}
