// Parameter update and weight re-packing.
//
// The host keeps all trainable parameters of a tower in ONE flat f32 buffer
// (regularised conv weights first, then everything else) with gradients, Adam
// moments and EMA shadows in identically laid out buffers, so the whole update —
// L2 regulariser gradient, un-scaling of the f16 loss scale, Adam, exponential
// moving average — is a single streaming launch.
//
// Reference: tf.train.AdamOptimizer + exponential_decay + ExponentialMovingAverage
// (multigpu_train.py:103-107,137-142), slim.l2_regularizer (nets/model.py:103),
// MomentumOptimizer (train_pixellink.py:243).
#include "common.h"

namespace {

struct AdamP {
  float lr_t, beta1, beta2, eps, wd, inv_scale, ema_decay;
  long long n, n_reg;
};

__global__ void adam_kernel(AdamP a, float* __restrict__ w, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ ema) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
    float wi = w[i];
    float gi = g[i] * a.inv_scale;
    if (i < a.n_reg) gi += a.wd * wi;
    const float mi = a.beta1 * m[i] + (1.f - a.beta1) * gi;
    const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    wi -= a.lr_t * mi / (sqrtf(vi) + a.eps);
    w[i] = wi;
    if (ema) {
      const float s = ema[i];
      ema[i] = s - (1.f - a.ema_decay) * (s - wi);
    }
  }
}

struct MomP {
  float lr, momentum, wd, inv_scale, ema_decay;
  long long n, n_reg;
};

__global__ void momentum_kernel(MomP a, float* __restrict__ w, const float* __restrict__ g,
                                float* __restrict__ acc, float* __restrict__ ema) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
    float wi = w[i];
    float gi = g[i] * a.inv_scale;
    if (i < a.n_reg) gi += a.wd * wi;
    const float ai = a.momentum * acc[i] + gi;
    acc[i] = ai;
    wi -= a.lr * ai;
    w[i] = wi;
    if (ema) {
      const float s = ema[i];
      ema[i] = s - (1.f - a.ema_decay) * (s - wi);
    }
  }
}

// HWIO f32 [taps][cin][cout] -> w_kc f16 [taps][cout][cin] and w_ck f16 [taps][cin][cout]
__global__ void pack_weights_kernel(const float* __restrict__ w, int taps, int cin, int cout,
                                    half_t* __restrict__ w_kc, half_t* __restrict__ w_ck) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* wt = w + (size_t)tap * cin * cout;
  for (int k = ty; k < 32; k += 8) {
    const int ci = ci0 + k, co = co0 + tx;
    float v = (ci < cin && co < cout) ? wt[(size_t)ci * cout + co] : 0.f;
    tile[k][tx] = v;
    if (w_ck && ci < cin && co < cout) w_ck[((size_t)tap * cin + ci) * cout + co] = (half_t)v;
  }
  __syncthreads();
  if (w_kc) {
    for (int k = ty; k < 32; k += 8) {
      const int co = co0 + k, ci = ci0 + tx;
      if (ci < cin && co < cout) w_kc[((size_t)tap * cout + co) * cin + ci] = (half_t)tile[tx][k];
    }
  }
}

// The same re-pack for MANY layers in one launch (after the optimiser step: 14 launches of ~7 us per VGG step, 62 per
// ResNet-50 step otherwise).  items[i] = one layer; block_first[i] = first block of layer i (block_first[n] = grid):
// a block finds its layer by bisection, then its (tap, ci tile, co tile) inside it.
struct PackItem {
  const float* w;
  half_t* w_kc;
  half_t* w_ck;
  int taps, cin, cout, tiles_ci, tiles_co;
  int ld_ck;                 // row length of w_ck: cout, or 32 for the fuse heads' zero-padded [cin][32] copy
};

__global__ void pack_weights_batch_kernel(const PackItem* __restrict__ items, const int* __restrict__ block_first, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n;                              // block_first[lo] <= blockIdx.x < block_first[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= block_first[mid]) lo = mid;
    else hi = mid;
  }
  const PackItem it = items[lo];
  int b = (int)blockIdx.x - block_first[lo];
  const int cot = b % it.tiles_co;
  b /= it.tiles_co;
  const int cit = b % it.tiles_ci;
  const int tap = b / it.tiles_ci;
  const int ci0 = cit * 32, co0 = cot * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* wt = it.w + (size_t)tap * it.cin * it.cout;
  for (int k = ty; k < 32; k += 8) {
    const int ci = ci0 + k, co = co0 + tx;
    const float v = (ci < it.cin && co < it.cout) ? wt[(size_t)ci * it.cout + co] : 0.f;
    tile[k][tx] = v;
    if (it.w_ck && ci < it.cin && co < it.cout) it.w_ck[((size_t)tap * it.cin + ci) * it.ld_ck + co] = (half_t)v;
  }
  __syncthreads();
  if (it.w_kc) {
    for (int k = ty; k < 32; k += 8) {
      const int co = co0 + k, ci = ci0 + tx;
      if (ci < it.cin && co < it.cout) it.w_kc[((size_t)tap * it.cout + co) * it.cin + ci] = (half_t)tile[tx][k];
    }
  }
}

// head weights f32 [cin][cout<=32] -> w_kc32 f16 [32][cin], w_ck32 f16 [cin][32] (zero padded)
__global__ void pack_small_kernel(const float* __restrict__ w, int cin, int cout,
                                  half_t* __restrict__ w_kc32, half_t* __restrict__ w_ck32) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cin * 32) return;
  const int ci = i >> 5, co = i & 31;
  const float v = co < cout ? w[ci * cout + co] : 0.f;
  w_ck32[i] = (half_t)v;
  w_kc32[(size_t)co * cin + ci] = (half_t)v;
}

__global__ void scale_kernel(float* __restrict__ x, long long n, float s) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    x[i] *= s;
}

__global__ void fill_kernel(float* __restrict__ x, long long n, float v) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    x[i] = v;
}

// scale * sum(x^2): slim.l2_regularizer(s)(w) = s * tf.nn.l2_loss(w) = s * sum(w^2) / 2 over the
// regularised variables (the REGULARIZATION_LOSSES term of `total_loss`, multigpu_train.py:36).
// Two launches, f64 partial per block and a fixed-order final sum: bitwise reproducible.
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void sumsq_partial_kernel(const float* __restrict__ x, long long n, double* __restrict__ partial) {
  __shared__ double sh[4];
  double acc = 0.0;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x4[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = x[(n4 << 2) + threadIdx.x];
    acc += (double)v * v;
  }
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void sumsq_final_kernel(const double* __restrict__ partial, int g, double scale, float* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < g; i += 256) acc += partial[i];
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[0] = (float)(scale * t);
}

unsigned sumsq_grid(long long n) {
  long long b = (n / 4 + 255) / 256;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (unsigned)b;
}

unsigned ogrid(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int ocr_adam_step(void* w, const void* g, void* m, void* v, void* ema, int64_t n,
                             int64_t n_regularized, float lr_t, float beta1, float beta2, float eps,
                             float weight_decay, float inv_loss_scale, float ema_decay,
                             void* stream) {
  OCR_CHECK_ARG(w && g && m && v && n > 0 && n_regularized >= 0 && n_regularized <= n);
  AdamP a{lr_t, beta1, beta2, eps, weight_decay, inv_loss_scale, ema_decay, n, n_regularized};
  hipLaunchKernelGGL(adam_kernel, dim3(ogrid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     static_cast<float*>(w), static_cast<const float*>(g), static_cast<float*>(m),
                     static_cast<float*>(v), static_cast<float*>(ema));
  return ocr_launch_status();
}

extern "C" int ocr_momentum_step(void* w, const void* g, void* accum, void* ema, int64_t n,
                                 int64_t n_regularized, float lr, float momentum, float weight_decay,
                                 float inv_loss_scale, float ema_decay, void* stream) {
  OCR_CHECK_ARG(w && g && accum && n > 0 && n_regularized >= 0 && n_regularized <= n);
  MomP a{lr, momentum, weight_decay, inv_loss_scale, ema_decay, n, n_regularized};
  hipLaunchKernelGGL(momentum_kernel, dim3(ogrid(n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, static_cast<float*>(w),
                     static_cast<const float*>(g), static_cast<float*>(accum),
                     static_cast<float*>(ema));
  return ocr_launch_status();
}

extern "C" int ocr_pack_weights_f16(const void* w_hwio_f32, int taps, int cin, int cout, void* w_kc,
                                    void* w_ck, void* stream) {
  OCR_CHECK_ARG(w_hwio_f32 && (w_kc || w_ck) && taps > 0 && cin > 0 && cout > 0);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(ocr_cdiv(cout, 32), ocr_cdiv(cin, 32), taps),
                     dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(w_hwio_f32), taps, cin, cout,
                     static_cast<half_t*>(w_kc), static_cast<half_t*>(w_ck));
  return ocr_launch_status();
}

extern "C" size_t ocr_pack_weights_batch_table_bytes(int n) { return (size_t)n * sizeof(PackItem) + (size_t)(n + 1) * sizeof(int); }

extern "C" int ocr_pack_weights_batch_table(int n, const void* const* w_hwio_f32, const int* taps, const int* cin,
                                            const int* cout, void* const* w_kc, void* const* w_ck, const int* ld_ck,
                                            void* table_host, int* grid_out) {
  OCR_CHECK_ARG(n > 0 && w_hwio_f32 && taps && cin && cout && w_kc && w_ck && table_host && grid_out);
  PackItem* items = static_cast<PackItem*>(table_host);
  int* first = reinterpret_cast<int*>(items + n);
  int g = 0;
  for (int i = 0; i < n; ++i) {
    OCR_CHECK_ARG(w_hwio_f32[i] && (w_kc[i] || w_ck[i]) && taps[i] > 0 && cin[i] > 0 && cout[i] > 0);
    items[i] = PackItem{static_cast<const float*>(w_hwio_f32[i]), static_cast<half_t*>(w_kc[i]),
                        static_cast<half_t*>(w_ck[i]), taps[i], cin[i], cout[i], ocr_cdiv(cin[i], 32), ocr_cdiv(cout[i], 32),
                        (ld_ck && ld_ck[i] > 0) ? ld_ck[i] : cout[i]};
    OCR_CHECK_ARG(items[i].ld_ck >= cout[i]);
    first[i] = g;
    g += taps[i] * items[i].tiles_ci * items[i].tiles_co;
  }
  first[n] = g;
  *grid_out = g;
  return OCR_OK;
}

extern "C" int ocr_pack_weights_batch_f16(const void* table_dev, int n, int grid, void* stream) {
  OCR_CHECK_ARG(table_dev && n > 0 && grid > 0);
  const PackItem* items = static_cast<const PackItem*>(table_dev);
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     items, reinterpret_cast<const int*>(items + n), n);
  return ocr_launch_status();
}

extern "C" int ocr_pack_weights_small_f16(const void* w_f32, int cin, int cout, void* w_kc32,
                                          void* w_ck32, void* stream) {
  OCR_CHECK_ARG(w_f32 && w_kc32 && w_ck32 && cin > 0 && cout > 0 && cout <= 32);
  hipLaunchKernelGGL(pack_small_kernel, dim3(ocr_cdiv(cin * 32, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(w_f32), cin, cout,
                     static_cast<half_t*>(w_kc32), static_cast<half_t*>(w_ck32));
  return ocr_launch_status();
}

extern "C" size_t ocr_sum_squares_workspace(int64_t n) { return (size_t)sumsq_grid(n) * sizeof(double); }

extern "C" int ocr_sum_squares_f32(const void* x, int64_t n, float scale, void* out_f32, void* workspace,
                                   size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(x && out_f32 && workspace && n > 0 && ((uintptr_t)x & 15) == 0);
  const unsigned g = sumsq_grid(n);
  if (ws_bytes < (size_t)g * sizeof(double)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(g), dim3(256), 0, st, static_cast<const float*>(x),
                     (long long)n, static_cast<double*>(workspace));
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, static_cast<const double*>(workspace),
                     (int)g, (double)scale, static_cast<float*>(out_f32));
  return ocr_launch_status();
}

extern "C" int ocr_scale_f32(void* x, int64_t n, float s, void* stream) {
  OCR_CHECK_ARG(x && n > 0);
  hipLaunchKernelGGL(scale_kernel, dim3(ogrid(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<float*>(x), (long long)n, s);
  return ocr_launch_status();
}

extern "C" int ocr_fill_f32(void* x, int64_t n, float value, void* stream) {
  OCR_CHECK_ARG(x && n > 0);
  hipLaunchKernelGGL(fill_kernel, dim3(ogrid(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<float*>(x), (long long)n, value);
  return ocr_launch_status();
}
