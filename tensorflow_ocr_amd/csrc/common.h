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
// a float z rounds (to nearest even) to a POSITIVE 16-bit value iff z > OCR_RELU_TIE: half the smallest subnormal
// (the kernels run with denormals preserved).  The fused BN-backward reductions test the ReLU mask of the STORED
// activation this way instead of converting to 16 bits and back (two conversions per element in epilogues that
// are bound by instruction issue).
#define OCR_RELU_TIE 0x1p-134f
#else
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
#define OCR_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define OCR_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define OCR_STORAGE_NAME "f16"
#define OCR_RELU_TIE 0x1p-25f
#endif
// ---- in-kernel clock stamps: DIAGNOSTIC BUILD ONLY (libocr_hip_diag.so, -DOCR_DIAG_CLOCK) --------------
// MI355X_MICROARCH.md "DVFS give-back" item 6: the clock a kernel really runs at is
// d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped once around its main loop.  The stamps go to a
// buffer of their own that nothing else reads; in the product libraries none of this is compiled.
#ifdef OCR_DIAG_CLOCK
#define OCR_DIAG_SLOTS 4096
struct OcrDiagStamp { unsigned long long t0, r0; };
#define OCR_DIAG_DECLARE(name) static __device__ unsigned long long name[OCR_DIAG_SLOTS][2]; \
  static __device__ unsigned long long name##_wg[OCR_DIAG_SLOTS][2];
// whole-workgroup stamp (kernel entry -> end of the epilogue's instruction stream of wave 0): with the main-loop stamp
// it splits a tile's cycles into main loop and prologue + epilogue
#define OCR_DIAG_WG_BEGIN() const unsigned long long diag_wg0_ = __builtin_amdgcn_s_memtime();
#define OCR_DIAG_WG_END(name)                                         \
  {                                                                   \
    const unsigned long long tw_ = __builtin_amdgcn_s_memtime();      \
    __builtin_amdgcn_s_waitcnt(0xC07F);                               \
    if (threadIdx.x == 0) {                                           \
      name##_wg[blockIdx.x % OCR_DIAG_SLOTS][0] = tw_ - diag_wg0_;    \
      name##_wg[blockIdx.x % OCR_DIAG_SLOTS][1] = diag_.t0 - diag_wg0_;   /* entry -> main loop: the prologue */ \
    }                                                                 \
  }
#define OCR_DIAG_BEGIN()                                              \
  OcrDiagStamp diag_;                                                 \
  diag_.t0 = __builtin_amdgcn_s_memtime();                            \
  diag_.r0 = __builtin_amdgcn_s_memrealtime();                        \
  __builtin_amdgcn_s_waitcnt(0xC07F);
#define OCR_DIAG_END(name)                                            \
  {                                                                   \
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();      \
    const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();  \
    __builtin_amdgcn_s_waitcnt(0xC07F);                               \
    if (threadIdx.x == 0) {                                           \
      name[blockIdx.x % OCR_DIAG_SLOTS][0] = t1_ - diag_.t0;          \
      name[blockIdx.x % OCR_DIAG_SLOTS][1] = r1_ - diag_.r0;          \
    }                                                                 \
  }
#define OCR_DIAG_READER(fn, name)                                                        \
  extern "C" int fn(void* host_out, int slots) {                                         \
    if (!host_out || slots <= 0 || slots > OCR_DIAG_SLOTS) return OCR_ERR_INVALID_ARG;   \
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(name), (size_t)slots * 16) == hipSuccess ? OCR_OK : OCR_ERR_HIP; \
  }                                                                                      \
  extern "C" int fn##_wg(void* host_out, int slots) {                                    \
    if (!host_out || slots <= 0 || slots > OCR_DIAG_SLOTS) return OCR_ERR_INVALID_ARG;   \
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(name##_wg), (size_t)slots * 16) == hipSuccess ? OCR_OK : OCR_ERR_HIP; \
  }
#else
#define OCR_DIAG_DECLARE(name)
#define OCR_DIAG_WG_BEGIN()
#define OCR_DIAG_WG_END(name)
#define OCR_DIAG_BEGIN()
#define OCR_DIAG_END(name)
#define OCR_DIAG_READER(fn, name)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// In-place accumulate on an accumulator-file (AGPR) quad.  With all 256 accumulator registers live hipcc's
// builtin form picks a destination different from the C operand and shuffles quads through
// v_accvgpr_read/write around every MFMA (968 such moves per 1152 MFMAs in the first build of
// conv3x3_w4_kernel); the asm form pins D = C.  The accumulators are read again only in the epilogue, far beyond
// the MFMA -> accvgpr_read hazard window (an explicit s_nop block precedes it anyway).
__device__ __forceinline__ void mfma16_acc(f32x4& c, const half8_t& a, const half8_t& b) {
#ifdef OCR_BF16
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#else
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#endif
}

// The LAST MFMA statement of a loop body (and of the kernel) carries its own drain: hipcc does not know that
// the asm MFMAs wrote the accumulator quads a few cycles ago, and the copies it places at a loop exit
// (v_accvgpr_read / v_accvgpr_mov re-shuffling the quads for the epilogue, spills) may sit directly behind
// the last asm statement — they did once the epilogue grew, and the quads of the last two MFMAs came out
// stale.  With the wait states inside the statement nothing can get between the MFMA and its drain.
__device__ __forceinline__ void mfma16_acc_drain(f32x4& c, const half8_t& a, const half8_t& b) {
#ifdef OCR_BF16
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" : "+a"(c) : "v"(a), "v"(b));
#else
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" : "+a"(c) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ void mfma16_acc_v_drain(f32x4& c, const half8_t& a, const half8_t& b) {
#ifdef OCR_BF16
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(c) : "v"(a), "v"(b));
#else
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(c) : "v"(a), "v"(b));
#endif
}

// the same with the accumulator quad in the vector-register half (kernels with more than 256 accumulators)
__device__ __forceinline__ void mfma16_acc_v(f32x4& c, const half8_t& a, const half8_t& b) {
#ifdef OCR_BF16
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#else
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
}

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
