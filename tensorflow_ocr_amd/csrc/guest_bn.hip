// GUEST kernels: the batch-norm backward APPLY passes in a register footprint that fits beside a matrix-core kernel.
//
// The weight-gradient kernel wgrad3_kernel<9,128> allocates 200 + 256 = 456 of a SIMD's 512 registers per lane (one wave
// per SIMD, one workgroup per CU): 56 registers per lane stay free on every SIMD of the chip for its whole launch, and a
// wave that needs no more than that is placed beside it (measured, scripts/coresidency_probe.hip: an HBM-streaming
// guest keeps its stand-alone rate beside a synthetic 456-register MFMA host and costs it nothing; beside the real
// wgrad3 the pair runs at 1.36x the serial rate, profiles/r05_coresidency_probe.json, r05_guest_probe.json).  The apply pass of layer L-1
// (reference: the gradient of slim.batch_norm + ReLU, nets/vgg.py:14-39 under nets/model_vgg_16.py:144) depends only on
// the input gradient of layer L, the weight gradient of layer L on neither: the recorded step runs the two side by
// side on two streams (train.TrainStep._replay).  bn_relu_bwd_kernel<1> needs 146 registers and could never be such a
// guest; these need <= 56:
//   * the apply step as the affine map dy = A*dz + B*y + C per channel (ocr_bn_bwd_coefficients; A = the forward scale),
//     dz = da * [fma(y, A, shift) rounds to a positive 16-bit value]: four coefficient rows instead of six;
//   * buffer addressing (one 32-bit offset register; the range check drops the ragged tail, no bounds branches);
//   * four channels per thread (8-byte accesses), three units (8 B of y + 8 B of da per lane each) in flight.
// The grid is one workgroup per CU when the pass runs beside hosts (guest_grid below).
#include "common.h"

namespace {

// Registers: hipcc ignores amdgpu_num_vgpr, so the budget (56 per lane: 512 - the 456 of wgrad3_kernel<9,128>) is met by
// construction — a thread owns FOUR channels (8-byte accesses: a wave still moves whole 128-byte lines, 512 B per
// instruction), which halves the coefficient rows a thread keeps (4 x 4 registers), and U units in flight (2 + 2
// registers each).  tests/test_host_cpu.py::test_guest_kernels_fit_beside_the_weight_gradient compiles this file with
// -Rpass-analysis=kernel-resource-usage and fails when a kernel here exceeds the budget or touches LDS / scratch.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
// Non-temporal policy (aux = 2) on the guests' loads (bit 0) and stores (bit 1): every byte is touched once, and the host
// beside them lives on what its L2 keeps.  Measured per variant inside one gpurun call (libraries built with
// -DOCR_GUEST_NT=0/1/2/3): 19.14-19.20 / 19.10-19.14 / 19.13 / 19.06-19.11 ms per step; on another box 18.87-18.90 -> 18.79-18.82.
#ifndef OCR_GUEST_NT
#define OCR_GUEST_NT 3
#endif
constexpr int kGuestAuxLd = (OCR_GUEST_NT & 1) ? 2 : 0, kGuestAuxSt = (OCR_GUEST_NT & 2) ? 2 : 0;
__device__ __forceinline__ half4_t ld8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(half4_t, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, kGuestAuxLd));
}
__device__ __forceinline__ void st8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, half4_t v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)soff, kGuestAuxSt);
}

// (The guest's waves at s_setprio 3 were measured — they issue a handful of instructions per hundred of the host's and are
// otherwise served after the older, host, wave of their SIMD: no difference, removed.)

// dy = A*dz + B*y + C, dz = relu ? da * [fma(y, A, S) > tie] : da.  Thread = one 4-channel chunk (fixed for the
// thread's lifetime: its coefficients stay in registers) x a strided set of pixels, U of them in flight.
// (A work-queue form — workgroups drawing 96 KB pieces from a counter — was built and measured: 25 % slower alone, the
// piece boundary drains the loads in flight; with the grid sized to what is RESIDENT, below, the static split is even.)
template <bool RELU, int U>
__global__ __launch_bounds__(256) void bn_apply_affine_kernel(
    const half_t* __restrict__ y, const half_t* __restrict__ da, const float* __restrict__ cA,
    const float* __restrict__ cS, const float* __restrict__ cB, const float* __restrict__ cC, unsigned units,
    int c, half_t* __restrict__ dy) {
  const int chunks = c >> 2, lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const size_t bytes = (size_t)units * c * 2;
  const __amdgpu_buffer_rsrc_t ry = rsrc_of(y, bytes), rg = rsrc_of(da, bytes), ro = rsrc_of(dy, bytes);
  float A[4], S[4], B[4], C[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    A[e] = cA[ch * 4 + e];
    S[e] = RELU ? cS[ch * 4 + e] : 0.f;
    B[e] = cB[ch * 4 + e];
    C[e] = cC[ch * 4 + e];
  }
  const unsigned row = (unsigned)c * 2;
  const unsigned stride = gridDim.x * (unsigned)lanes;           // units between a thread's successive visits
  const unsigned sbytes = stride * row;
  unsigned u = blockIdx.x * (unsigned)lanes + (unsigned)ul;
  unsigned off = u * row + (unsigned)ch * 8;
  // offsets past the tensor read zeros and their stores are dropped by the range check: no bounds branches
  for (; u < units; u += U * stride, off += U * sbytes) {
    half4_t yv[U], gv[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      yv[k] = ld8(ry, off, k * sbytes);
      gv[k] = ld8(rg, off, k * sbytes);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yf = (float)yv[k][e];
        float dz = (float)gv[k][e];
        if (RELU) dz = __builtin_fmaf(yf, A[e], S[e]) > OCR_RELU_TIE ? dz : 0.f;
        o[e] = (half_t)__builtin_fmaf(A[e], dz, __builtin_fmaf(B[e], yf, C[e]));
      }
      st8(ro, off, k * sbytes, o);
    }
  }
}

// The same for a layer whose only reader is its 2x2/2 max-pool (conv1_2, conv2_2): the pooled gradient is routed to
// the window's first maximum (bits 0-1 of the forward's index byte; bit 2 = the pooled activation was positive) and
// dy = A*dz + B*y + C at all four positions.  Even h and w only (the host keeps the general kernel for ragged maps).
// Two pooled units (2 x 4 full-resolution loads) in flight per thread.
template <bool RELU>
__global__ __launch_bounds__(256) void bn_pool_apply_affine_kernel(
    const half_t* __restrict__ y, const half_t* __restrict__ da_pool, const unsigned char* __restrict__ argmax,
    const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC, int n, int h, int w,
    int c, half_t* __restrict__ dy) {
  const int chunks = c >> 2, lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const int oh = h >> 1, ow = w >> 1;
  const unsigned units = (unsigned)n * oh * ow;
  const size_t full = (size_t)n * h * w * c * 2, pooled = (size_t)units * c * 2;
  const __amdgpu_buffer_rsrc_t ry = rsrc_of(y, full), ro = rsrc_of(dy, full), rg = rsrc_of(da_pool, pooled),
                               ra = rsrc_of(argmax, pooled / 2);
  float A[4], B[4], C[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    A[e] = cA[ch * 4 + e];
    B[e] = cB[ch * 4 + e];
    C[e] = cC[ch * 4 + e];
  }
  const unsigned row = (unsigned)c * 2;
  const unsigned below = (unsigned)w * row;                     // one full-resolution row down
  const unsigned stride = gridDim.x * (unsigned)lanes;
  for (unsigned u0 = blockIdx.x * (unsigned)lanes + (unsigned)ul; u0 < units; u0 += 2 * stride) {
    unsigned foff[2];
    half4_t gp[2], yv[2][4];
    unsigned am[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned u = u0 + j * stride;                       // (past the end: every access falls out of range)
      const unsigned ox = u % (unsigned)ow, t = u / (unsigned)ow;       // t = img * oh + oy
      const unsigned poff = u < units ? u * row + (unsigned)ch * 8 : 0x80000000u;
      foff[j] = u < units ? ((t * 2u) * (unsigned)w + ox * 2u) * row + (unsigned)ch * 8 : 0x80000000u;
      gp[j] = ld8(rg, poff, 0);
      am[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)(poff >> 1), 0, 0);
      yv[j][0] = ld8(ry, foff[j], 0);
      yv[j][1] = ld8(ry, foff[j], row);
      yv[j][2] = ld8(ry, foff[j], below);
      yv[j][3] = ld8(ry, foff[j], below + row);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (unsigned k = 0; k < 4; ++k) {
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned byte = (am[j] >> (8 * e)) & (RELU ? 7u : 3u);
          const float dz = byte == (RELU ? (k | 4u) : k) ? (float)gp[j][e] : 0.f;
          o[e] = (half_t)__builtin_fmaf(A[e], dz, __builtin_fmaf(B[e], (float)yv[j][k][e], C[e]));
        }
        st8(ro, foff[j], (k & 1 ? row : 0) + (k & 2 ? below : 0), o);
      }
    }
  }
}

// ... and for a pooled layer that is ALSO an end point (conv3_3 / conv4_3 of nets/vgg.py:22-30: the fuse heads read the
// full-resolution activation, nets/model_vgg_16.py:160-172): the activation's gradient is the heads' full-resolution
// contribution plus the pooled gradient routed to each window's first maximum (bits 0-1 of the forward's index byte),
// masked by the ReLU of the position itself:  dz = (da_full + [first max] * da_pool) * [fma(y, A, S) > tie].
// One pooled unit (4 positions x (y, da_full) + the pooled gradient + the index) in flight per thread.
template <bool RELU>
__global__ __launch_bounds__(256) void bn_poolfull_apply_affine_kernel(
    const half_t* __restrict__ y, const half_t* __restrict__ da_full, const half_t* __restrict__ da_pool,
    const unsigned char* __restrict__ argmax, const float* __restrict__ cA, const float* __restrict__ cS,
    const float* __restrict__ cB, const float* __restrict__ cC, int n, int h, int w, int c, half_t* __restrict__ dy) {
  const int chunks = c >> 2, lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const int oh = h >> 1, ow = w >> 1;
  const unsigned units = (unsigned)n * oh * ow;
  const size_t full = (size_t)n * h * w * c * 2, pooled = (size_t)units * c * 2;
  const __amdgpu_buffer_rsrc_t ry = rsrc_of(y, full), rf = rsrc_of(da_full, full), ro = rsrc_of(dy, full),
                               rg = rsrc_of(da_pool, pooled), ra = rsrc_of(argmax, pooled / 2);
  float A[4], S[4], B[4], C[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    A[e] = cA[ch * 4 + e];
    S[e] = RELU ? cS[ch * 4 + e] : 0.f;
    B[e] = cB[ch * 4 + e];
    C[e] = cC[ch * 4 + e];
  }
  const unsigned row = (unsigned)c * 2;
  const unsigned below = (unsigned)w * row;
  const unsigned stride = gridDim.x * (unsigned)lanes;
  for (unsigned u = blockIdx.x * (unsigned)lanes + (unsigned)ul; u < units; u += stride) {
    const unsigned ox = u % (unsigned)ow, t = u / (unsigned)ow;
    const unsigned poff = u * row + (unsigned)ch * 8;
    const unsigned foff = ((t * 2u) * (unsigned)w + ox * 2u) * row + (unsigned)ch * 8;
    const half4_t gp = ld8(rg, poff, 0);
    const unsigned am = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)(poff >> 1), 0, 0);
    half4_t yv[4], gv[4];
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
      const unsigned so = (k & 1 ? row : 0) + (k & 2 ? below : 0);
      yv[k] = ld8(ry, foff, so);
      gv[k] = ld8(rf, foff, so);
    }
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yf = (float)yv[k][e];
        float g = ((am >> (8 * e)) & 3u) == k ? (float)gp[e] : 0.f;       // (the general kernel's order: routed + full)
        g += (float)gv[k][e];
        const float dz = RELU ? (__builtin_fmaf(yf, A[e], S[e]) > OCR_RELU_TIE ? g : 0.f) : g;
        o[e] = (half_t)__builtin_fmaf(A[e], dz, __builtin_fmaf(B[e], yf, C[e]));
      }
      st8(ro, foff, (k & 1 ? row : 0) + (k & 2 ? below : 0), o);
    }
  }
}

// The REDUCTION pass of an end-point layer's batch-norm backward (conv3_3 / conv4_3 / conv5_3 / fc7: several consumers
// contributed to the activation's gradient, so no convolution's epilogue could sum it) in the same footprint: per thread
// (sum dz, sum dz*y) of its four channels over its pixels, converted to (sum dz, sum dz*xhat) at the end and written as the
// thread's own partial row — no LDS, no cross-lane step; ocr_bn_bwd_coefficients folds the rows in a fixed order.
// bn_relu_bwd_kernel<0> (166 registers, one 16-byte load in flight, an LDS fold) ran these passes at 1.8-3.3 TB/s on
// the critical path between two input-gradient convolutions; this one runs beside held-back weight gradients.
template <bool RELU, int U>
__global__ __launch_bounds__(256) void bn_reduce_rows_kernel(
    const half_t* __restrict__ y, const half_t* __restrict__ da, const float* __restrict__ cA,
    const float* __restrict__ cS, const float* __restrict__ cMu, const float* __restrict__ cIs, unsigned units, int c,
    float* __restrict__ partial) {
  const int chunks = c >> 2, lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const size_t bytes = (size_t)units * c * 2;
  const __amdgpu_buffer_rsrc_t ry = rsrc_of(y, bytes), rg = rsrc_of(da, bytes);
  float A[4], S[4], s[4], q[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    A[e] = cA[ch * 4 + e];
    S[e] = RELU ? cS[ch * 4 + e] : 0.f;
    s[e] = 0.f;
    q[e] = 0.f;
  }
  const unsigned row = (unsigned)c * 2;
  const unsigned stride = gridDim.x * (unsigned)lanes;
  const unsigned sbytes = stride * row;
  unsigned u = blockIdx.x * (unsigned)lanes + (unsigned)ul;
  unsigned off = u * row + (unsigned)ch * 8;
  for (; u < units; u += U * stride, off += U * sbytes) {       // (offsets past the tensor read zeros: dz = 0)
    half4_t yv[U], gv[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      yv[k] = ld8(ry, off, k * sbytes);
      gv[k] = ld8(rg, off, k * sbytes);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      // unit by unit: hipcc converts every unit's eight values up front otherwise (24 more registers live: 64 in all);
      // the empty asm statements keep their order, so unit k's values are not touched before unit k-1's sums exist
      u32x2 ry2 = __builtin_bit_cast(u32x2, yv[k]), rg2 = __builtin_bit_cast(u32x2, gv[k]);
      asm volatile("" : "+v"(ry2), "+v"(rg2));
      const half4_t y4 = __builtin_bit_cast(half4_t, ry2), g4 = __builtin_bit_cast(half4_t, rg2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yf = (float)y4[e];
        float dz = (float)g4[e];
        if (RELU) dz = __builtin_fmaf(yf, A[e], S[e]) > OCR_RELU_TIE ? dz : 0.f;
        s[e] += dz;
        q[e] = __builtin_fmaf(dz, yf, q[e]);
      }
      asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    }
  }
  float* const out = partial + ((size_t)(blockIdx.x * (unsigned)lanes + (unsigned)ul) * 2) * c + ch * 4;
  f32x4 so, qo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    so[e] = s[e];
    qo[e] = (q[e] - cMu[ch * 4 + e] * s[e]) * cIs[ch * 4 + e];      // sum dz*xhat = invstd * (sum dz*y - mean * sum dz)
  }
  *reinterpret_cast<f32x4*>(out) = so;
  *reinterpret_cast<f32x4*>(out + c) = qo;
}

// ... of a pooled end point (conv3_3 / conv4_3): dz = (da_full + [first max] * da_pool) * [ReLU mask of the position],
// the forward's index byte as in bn_poolfull_apply_affine_kernel.  One pooled unit in flight per thread.
template <bool RELU>
__global__ __launch_bounds__(256) void bn_poolfull_reduce_rows_kernel(
    const half_t* __restrict__ y, const half_t* __restrict__ da_full, const half_t* __restrict__ da_pool,
    const unsigned char* __restrict__ argmax, const float* __restrict__ cA, const float* __restrict__ cS,
    const float* __restrict__ cMu, const float* __restrict__ cIs, int n, int h, int w, int c,
    float* __restrict__ partial) {
  const int chunks = c >> 2, lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const int oh = h >> 1, ow = w >> 1;
  const unsigned units = (unsigned)n * oh * ow;
  const size_t full = (size_t)n * h * w * c * 2, pooled = (size_t)units * c * 2;
  const __amdgpu_buffer_rsrc_t ry = rsrc_of(y, full), rf = rsrc_of(da_full, full), rg = rsrc_of(da_pool, pooled),
                               ra = rsrc_of(argmax, pooled / 2);
  float A[4], S[4], s[4], q[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    A[e] = cA[ch * 4 + e];
    S[e] = RELU ? cS[ch * 4 + e] : 0.f;
    s[e] = 0.f;
    q[e] = 0.f;
  }
  const unsigned row = (unsigned)c * 2;
  const unsigned below = (unsigned)w * row;
  const unsigned stride = gridDim.x * (unsigned)lanes;
  for (unsigned u = blockIdx.x * (unsigned)lanes + (unsigned)ul; u < units; u += stride) {
    const unsigned ox = u % (unsigned)ow, t = u / (unsigned)ow;
    const unsigned poff = u * row + (unsigned)ch * 8;
    const unsigned foff = ((t * 2u) * (unsigned)w + ox * 2u) * row + (unsigned)ch * 8;
    const half4_t gp = ld8(rg, poff, 0);
    const unsigned am = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, (int)(poff >> 1), 0, 0);
    half4_t yv[4], gv[4];
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
      const unsigned so = (k & 1 ? row : 0) + (k & 2 ? below : 0);
      yv[k] = ld8(ry, foff, so);
      gv[k] = ld8(rf, foff, so);
    }
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
      // position by position (see bn_reduce_rows_kernel; the index word too: its sixteen selections would all be made up front)
      u32x2 ry2 = __builtin_bit_cast(u32x2, yv[k]), rg2 = __builtin_bit_cast(u32x2, gv[k]);
      unsigned amk = am;
      asm volatile("" : "+v"(ry2), "+v"(rg2), "+v"(amk));
      const half4_t y4 = __builtin_bit_cast(half4_t, ry2), g4 = __builtin_bit_cast(half4_t, rg2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yf = (float)y4[e];
        float g = ((amk >> (8 * e)) & 3u) == k ? (float)gp[e] : 0.f;      // (the general kernel's order: routed + full)
        g += (float)g4[e];
        const float dz = RELU ? (__builtin_fmaf(yf, A[e], S[e]) > OCR_RELU_TIE ? g : 0.f) : g;
        s[e] += dz;
        q[e] = __builtin_fmaf(dz, yf, q[e]);
      }
      asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    }
  }
  float* const out = partial + ((size_t)(blockIdx.x * (unsigned)lanes + (unsigned)ul) * 2) * c + ch * 4;
  f32x4 so, qo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    so[e] = s[e];
    qo[e] = (q[e] - cMu[ch * 4 + e] * s[e]) * cIs[ch * 4 + e];
  }
  *reinterpret_cast<f32x4*>(out) = so;
  *reinterpret_cast<f32x4*>(out + c) = qo;
}

bool pow2g(int v) { return v > 0 && (v & (v - 1)) == 0; }

// grid: `max_workgroups` when the caller names one (the recorded step asks for 256 = ONE per CU for a pass it runs
// beside weight gradients: that is what is resident beside a 456-register host, so the static split stays even, and a
// second host finds its registers free when the first has gone — more workgroups would fill the CUs the moment a host
// retires and keep the next host out until the whole pass has finished: measured, 427 -> 735 us for that host);
// otherwise 4 per CU, which is what the pass needs to reach the HBM rate alone
unsigned guest_grid(size_t units, size_t per_wg, int max_workgroups) {
  size_t b = (units + per_wg - 1) / per_wg;
  const size_t lim = max_workgroups > 0 ? (size_t)max_workgroups : (size_t)1024;
  if (b > lim) b = lim;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

// Apply step of the batch-norm (+ReLU) backward as an affine map of (dz, y): the second half of
// ocr_bn_relu_bwd_apply_f16 with the coefficients from ocr_bn_bwd_coefficients (coef_a = the forward scale).
// One launch, <= 56 registers per lane, no LDS: a guest beside the weight-gradient kernels.
extern "C" int ocr_bn_relu_bwd_apply_affine_f16(const void* y, const void* da, const void* scale, const void* shift,
                                                const void* coef_b, const void* coef_c, int n, int h, int w, int c,
                                                int relu, void* dy, int max_workgroups, void* stream) {
  OCR_CHECK_ARG(y && da && scale && shift && coef_b && coef_c && dy && n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2g(c / 4) && c / 4 <= 256);
  const size_t units = (size_t)n * h * w;
  OCR_CHECK_SHAPE(units * (size_t)c * 2 < (1ull << 31));   // 32-bit buffer offsets, 0x80000000 = out of range
  const int lanes = 256 / (c / 4);
  constexpr int U = 3;
  const unsigned grid = guest_grid(units, (size_t)lanes * U, max_workgroups);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (relu)
    hipLaunchKernelGGL((bn_apply_affine_kernel<true, U>), dim3(grid), dim3(256), 0, st, static_cast<const half_t*>(y),
                       static_cast<const half_t*>(da), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<const float*>(coef_b), static_cast<const float*>(coef_c), (unsigned)units, c,
                       static_cast<half_t*>(dy));
  else
    hipLaunchKernelGGL((bn_apply_affine_kernel<false, U>), dim3(grid), dim3(256), 0, st, static_cast<const half_t*>(y),
                       static_cast<const half_t*>(da), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<const float*>(coef_b), static_cast<const float*>(coef_c), (unsigned)units, c,
                       static_cast<half_t*>(dy));
  return ocr_launch_status();
}

// ... and of ocr_bn_relu_pool_bwd_idx_apply_f16 (pooled layers, stored first-max index).  OCR_ERR_UNSUPPORTED for odd
// h / w: the caller keeps the general kernel for those.
extern "C" int ocr_bn_relu_pool_bwd_idx_apply_affine_f16(const void* y, const void* argmax_u8, const void* da_pool,
                                                         const void* coef_a, const void* coef_b, const void* coef_c,
                                                         int n, int h, int w, int c, int relu, void* dy, int max_workgroups,
                                                         void* stream) {
  OCR_CHECK_ARG(y && argmax_u8 && da_pool && coef_a && coef_b && coef_c && dy && n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2g(c / 4) && c / 4 <= 256 && h % 2 == 0 && w % 2 == 0);
  const size_t full = (size_t)n * h * w * c * 2;
  OCR_CHECK_SHAPE(full < (1ull << 31));
  const int lanes = 256 / (c / 4);
  const unsigned grid = guest_grid((size_t)n * (h / 2) * (w / 2), (size_t)lanes * 2, max_workgroups);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (relu)
    hipLaunchKernelGGL(bn_pool_apply_affine_kernel<true>, dim3(grid), dim3(256), 0, st, static_cast<const half_t*>(y),
                       static_cast<const half_t*>(da_pool), static_cast<const unsigned char*>(argmax_u8),
                       static_cast<const float*>(coef_a), static_cast<const float*>(coef_b),
                       static_cast<const float*>(coef_c), n, h, w, c, static_cast<half_t*>(dy));
  else
    hipLaunchKernelGGL(bn_pool_apply_affine_kernel<false>, dim3(grid), dim3(256), 0, st, static_cast<const half_t*>(y),
                       static_cast<const half_t*>(da_pool), static_cast<const unsigned char*>(argmax_u8),
                       static_cast<const float*>(coef_a), static_cast<const float*>(coef_b),
                       static_cast<const float*>(coef_c), n, h, w, c, static_cast<half_t*>(dy));
  return ocr_launch_status();
}

// ... and of ocr_bn_relu_bwd_f16(pool = 2, da_full given) for pooled end-point layers, with the forward's first-max index
// (ocr_bn_relu_pool_idx_f16 with a_full): coefficients from ocr_bn_relu_bwd_reduce_f16(da_pool given).  Even h, w.
extern "C" int ocr_bn_relu_poolfull_bwd_apply_affine_f16(const void* y, const void* da_full, const void* da_pool,
                                                         const void* argmax_u8, const void* scale, const void* shift,
                                                         const void* coef_b, const void* coef_c, int n, int h, int w, int c,
                                                         int relu, void* dy, int max_workgroups, void* stream) {
  OCR_CHECK_ARG(y && da_full && da_pool && argmax_u8 && scale && shift && coef_b && coef_c && dy && n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2g(c / 4) && c / 4 <= 256 && h % 2 == 0 && w % 2 == 0);
  const size_t full = (size_t)n * h * w * c * 2;
  OCR_CHECK_SHAPE(full < (1ull << 31));
  const int lanes = 256 / (c / 4);
  const unsigned grid = guest_grid((size_t)n * (h / 2) * (w / 2), (size_t)lanes, max_workgroups);
  hipStream_t st = static_cast<hipStream_t>(stream);
  OCR_CHECK_SHAPE(relu != 0);          // (the ReLU-less instantiation needs 58 registers: no net builds a pooled end point without ReLU)
  hipLaunchKernelGGL(bn_poolfull_apply_affine_kernel<true>, dim3(grid), dim3(256), 0, st, static_cast<const half_t*>(y),
                     static_cast<const half_t*>(da_full), static_cast<const half_t*>(da_pool),
                     static_cast<const unsigned char*>(argmax_u8), static_cast<const float*>(scale),
                     static_cast<const float*>(shift), static_cast<const float*>(coef_b), static_cast<const float*>(coef_c),
                     n, h, w, c, static_cast<half_t*>(dy));
  return ocr_launch_status();
}

// Reduction pass of an end-point layer's batch-norm backward as a guest (bn_reduce_rows_kernel): partial rows
// [ocr_bn_relu_bwd_reduce_rows_count(...)][2][c] f32 = (sum dz, sum dz*xhat) per thread lane, for
// ocr_bn_bwd_coefficients.  dz = (da_full + [first max] * da_pool) * [fma(y, scale, shift) rounds to a positive value];
// da_pool / argmax_u8 (both or neither): the layer's 2x2/2 max-pool gradient and the forward's index
// (ocr_bn_relu_pool_idx_f16 with a_full), even h and w.  The grid never exceeds one workgroup per CU (what is resident
// beside a weight-gradient host), so the row count does not depend on where the recorded step places the launch;
// max_workgroups (0 = 256) can only lower it and must then be passed to the row count too.
static unsigned reduce_rows_grid(int n, int h, int w, int c, int pooled, int max_workgroups, int* lanes_out) {
  const int lanes = 256 / (c / 4);
  const size_t units = pooled ? (size_t)n * (h / 2) * (w / 2) : (size_t)n * h * w;
  const int cap = max_workgroups > 0 && max_workgroups < 256 ? max_workgroups : 256;
  *lanes_out = lanes;
  return guest_grid(units, (size_t)lanes * (pooled ? 1 : 4), cap);
}
extern "C" int ocr_bn_relu_bwd_reduce_rows_count(int n, int h, int w, int c, int pooled, int max_workgroups) {
  if (n <= 0 || h <= 0 || w <= 0 || c % 8 != 0 || !pow2g(c / 4) || c / 4 > 256) return -1;
  int lanes;
  return (int)reduce_rows_grid(n, h, w, c, pooled, max_workgroups, &lanes) * lanes;
}
extern "C" int ocr_bn_relu_bwd_reduce_rows_f16(const void* y, const void* da_full, const void* da_pool, const void* argmax_u8,
                                               const void* scale, const void* shift, const void* save_mean,
                                               const void* save_invstd, int n, int h, int w, int c, int relu, void* partial,
                                               int max_workgroups, void* stream) {
  OCR_CHECK_ARG(y && da_full && scale && shift && save_mean && save_invstd && partial && n > 0 && h > 0 && w > 0);
  OCR_CHECK_ARG((da_pool == nullptr) == (argmax_u8 == nullptr));
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2g(c / 4) && c / 4 <= 256);
  OCR_CHECK_SHAPE((size_t)n * h * w * c * 2 < (1ull << 31));
  const int pooled = da_pool != nullptr;
  OCR_CHECK_SHAPE(!pooled || (h % 2 == 0 && w % 2 == 0 && relu != 0));
  int lanes;
  const unsigned grid = reduce_rows_grid(n, h, w, c, pooled, max_workgroups, &lanes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const half_t* yp = static_cast<const half_t*>(y);
  const half_t* gf = static_cast<const half_t*>(da_full);
  const float *sc = static_cast<const float*>(scale), *sh = static_cast<const float*>(shift),
              *mu = static_cast<const float*>(save_mean), *is = static_cast<const float*>(save_invstd);
  float* out = static_cast<float*>(partial);
  if (pooled)
    hipLaunchKernelGGL(bn_poolfull_reduce_rows_kernel<true>, dim3(grid), dim3(256), 0, st, yp, gf,
                       static_cast<const half_t*>(da_pool), static_cast<const unsigned char*>(argmax_u8), sc, sh, mu, is,
                       n, h, w, c, out);
  else if (relu)
    hipLaunchKernelGGL((bn_reduce_rows_kernel<true, 4>), dim3(grid), dim3(256), 0, st, yp, gf, sc, sh, mu, is,
                       (unsigned)((size_t)n * h * w), c, out);
  else
    hipLaunchKernelGGL((bn_reduce_rows_kernel<false, 4>), dim3(grid), dim3(256), 0, st, yp, gf, sc, sh, mu, is,
                       (unsigned)((size_t)n * h * w), c, out);
  return ocr_launch_status();
}
