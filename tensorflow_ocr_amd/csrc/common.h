// Shared device/host helpers for the gfx950 kernels behind include/ocr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ocr_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
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
