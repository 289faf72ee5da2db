// Shared device/host helpers for the gfx950 kernels behind include/ocr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ocr_hip.h"

// 16-bit storage type of activations / packed weights / activation gradients.  The library is built
// twice from the same sources: libocr_hip.so (IEEE half, the default product path) and
// libocr_hip_bf16.so (-DOCR_BF16: bfloat16 storage + bf16 MFMA, BASELINE config "ResNet-50 bf16").
// Entry points keep their `_f16` names in both: the suffix names the 16-bit storage slot.
#ifdef OCR_BF16
typedef __bf16 half_t;
typedef __bf16 half2_t __attribute__((ext_vector_type(2)));
typedef __bf16 half4_t __attribute__((ext_vector_type(4)));
typedef __bf16 half8_t __attribute__((ext_vector_type(8)));
#define OCR_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define OCR_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define OCR_STORAGE_NAME "bf16"
#else
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
#define OCR_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define OCR_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define OCR_STORAGE_NAME "f16"
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define OCR_CHECK_ARG(cond)                 \
  do {                                      \
    if (!(cond)) return OCR_ERR_INVALID_ARG; \
  } while (0)

#define OCR_CHECK_SHAPE(cond)                 \
  do {                                        \
    if (!(cond)) return OCR_ERR_UNSUPPORTED;  \
  } while (0)

static inline int ocr_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? OCR_OK : OCR_ERR_HIP;
}

static inline int ocr_cdiv(int a, int b) { return (a + b - 1) / b; }

// wave64 all-lane sum through DPP-free shuffles (width 64).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// out[i] = scale * sum_{s<S} partial[s*elems + i]: one wave per output element (64 row lanes),
// fixed summation order -> reproducible.  Launch with sum_rows_grid(elems) blocks of 256.
static __global__ void ocr_sum_rows_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                           int elems, int S, float scale) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= elems) return;
  float a = 0.f;
  for (int s = lane; s < S; s += 64) a += partial[(size_t)s * elems + i];
  a = wave_sum(a);
  if (lane == 0) out[i] = a * scale;
}
static inline unsigned sum_rows_grid(int elems) { return (unsigned)((elems + 3) / 4); }
// The same sum with the result row split over two destinations (elements [0, n_a) -> out_a, the rest ->
// out_b; out_b may be null when n_a == elems, a null out_a drops the first part): the row is written where
// it is wanted instead of being copied there by two more launches.
static __global__ void ocr_sum_rows_split_kernel(const float* __restrict__ partial, float* __restrict__ out_a,
                                                 int n_a, float* __restrict__ out_b, int elems, int S,
                                                 float scale) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= elems) return;
  float a = 0.f;
  for (int s = lane; s < S; s += 64) a += partial[(size_t)s * elems + i];
  a = wave_sum(a);
  if (lane == 0) {
    if (i < n_a) { if (out_a) out_a[i] = a * scale; }
    else if (out_b) out_b[i - n_a] = a * scale;
  }
}
