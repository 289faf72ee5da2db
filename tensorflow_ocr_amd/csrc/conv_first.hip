// First VGG convolution (conv1_1, cin = 3; reference nets/vgg.py:14) and its
// weight gradient.  The image is kept as [n,h,w,4] f16 (RGB minus mean, 4th
// channel zero: ocr_prep_images_f16) so a pixel is one 8-byte word.
//
// Forward: per kernel row ky one MFMA k-step of 16 = 4 pixels (kx = 0..3, the
// 4th with zero weights) x 4 channels, i.e. a lane's 8 k-values are two adjacent
// pixels = 16 contiguous bytes of the LDS halo tile.  K is padded 27 -> 48, which
// is irrelevant: the layer is bound by its 64-channel output stream.
//
// Weight gradient: 27 x cout outputs, K = all pixels.  A VALU kernel: one lane
// per cout, all 27 (ky,kx,c) sums in registers, the halo pixel (4 channels) is a
// wave-uniform 8-byte LDS broadcast.  Partials per workgroup -> fixed-order sum.
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int HW = 36;  // halo row pitch in pixels (34 used + 2 so kx=3 stays in range)

struct FirstP {
  int n, h, w, cout, tiles_x, tiles_y, m_tiles, flags;
};

__device__ __forceinline__ void load_halo4(const half_t* __restrict__ x4, char* halo, int img, int h,
                                           int w, int iy0, int ix0) {
  for (int i = threadIdx.x; i < 10 * HW; i += 256) {
    const int hy = i / HW, hx = i - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    u32x2 v = {0u, 0u};
    if (iy >= 0 && iy < h && ix >= 0 && ix < w)
      v = *reinterpret_cast<const u32x2*>(x4 + (((size_t)img * h + iy) * w + ix) * 4);
    *reinterpret_cast<u32x2*>(halo + i * 8) = v;
  }
}

// The layer's arithmetic, shared by every kernel that evaluates it (the forward, its batch-norm-apply form and the
// weight gradient that recomputes y instead of reading it): the SAME MFMA sequence on the same operand layout gives
// the same f32 accumulators, hence bit-identical 16-bit y wherever it is evaluated.  Wave w owns tile rows 2w, 2w+1.
__device__ __forceinline__ void first_weights(const half_t* __restrict__ wf, int cout, int co0, int lane,
                                              half8_t (&a)[3][2]) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      a[ky][i] = *reinterpret_cast<const half8_t*>(wf + ((size_t)(ky * cout + co0 + i * 32 + r)) * 16 + 8 * hh);
}

__device__ __forceinline__ void first_mfma(const char* halo, const half8_t (&a)[3][2], int wave, int lane,
                                           f32x16 (&acc)[2][2]) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ty = wave * 2 + t;
      const char* src = halo + ((ty + ky) * HW + r + 2 * hh) * 8;
      half4_t lo = *reinterpret_cast<const half4_t*>(src);
      half4_t hi = *reinterpret_cast<const half4_t*>(src + 8);
      half8_t b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[i][t] = OCR_MFMA_32x32x16(a[ky][i], b, acc[i][t], 0, 0, 0);
    }
  }
}

// BNAPPLY (conv + batch norm + ReLU nets, second pass of conv1_1): the statistics exist (the first pass), so the
// convolution is evaluated AGAIN — 67 MB of image instead of the 1 GiB of y the element-wise pass would read — and
// the tile leaves as the activation a = relu(scale * y16 + shift), y16 = the 16-bit value the first pass stored.
struct FirstAct {
  const float *scale, *shift;
  int relu;
};

template <bool BNAPPLY>
__global__ __launch_bounds__(256) void conv_first_kernel(FirstP p, const half_t* __restrict__ x4,
                                                         const half_t* __restrict__ wf,
                                                         const float* __restrict__ bias,
                                                         half_t* __restrict__ y,
                                                         float* __restrict__ stats, FirstAct act) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_tiles = p.cout / 64;
  const int nt = blockIdx.x % n_tiles;
  const int mt = blockIdx.x / n_tiles;
  const int txi = mt % p.tiles_x;
  const int tmp = mt / p.tiles_x;
  const int tyi = tmp % p.tiles_y;
  const int img = tmp / p.tiles_y;
  const int co0 = nt * 64;

  half8_t a[3][2];
  first_weights(wf, p.cout, co0, lane, a);
  load_halo4(x4, halo, img, p.h, p.w, tyi * TILE_H - 1, txi * TILE_W - 1);
  __syncthreads();
  f32x16 acc[2][2];
  first_mfma(halo, a, wave, lane, acc);
  __syncthreads();
  if constexpr (BNAPPLY) {
    constexpr int OSTR = 64 * 2 + 16;                 // conv_epilogue's staging layout
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = i * 32 + q * 8 + hh * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(act.scale + co0 + col);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(act.shift + co0 + col);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int px = (wave * 2 + t) * 32 + r;
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y16 = (float)(half_t)acc[i][t][q * 4 + e];
            float f = __builtin_fmaf(y16, sc[e], sh[e]);
            if (act.relu) f = f > 0.f ? f : 0.f;
            o[e] = (half_t)f;
          }
          *reinterpret_cast<half4_t*>(smem + px * OSTR + col * 2) = o;
        }
      }
    conv_epilogue_store<64, 256>(smem, 0, y, nullptr, img, tyi, txi, mt, co0, p.h, p.w, p.cout, nullptr);
  } else {
    conv_epilogue<64, 2, 2, 1>(acc, smem, p.flags, bias, y, stats, img, tyi, txi, mt, co0, p.h, p.w,
                               p.cout, 0, wave, true);
  }
}

// ---- weight gradient -------------------------------------------------------
// MFMA form: M = 32 patch rows (27 used: (ky*3+kx)*3+c), N = 64 couts, K = pixels.  Per 8x32
// tile the 256 patch rows are built in LDS from the halo ([px][32] f16, row stride padded to
// 128 B so the transposing reads are conflict-free) next to the dy tile; both MFMA operands come
// from ds_read_b64_tr_b16.  The kernel is bound by streaming dy (128 B per pixel).
typedef short short4v_f __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) short4v_f* lds_s4_ptr_f;

__device__ __forceinline__ half8_t tr_pair_f(const char* base, int second_off) {
  short4v_f lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr_f)(base));
  short4v_f hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr_f)(base + second_off));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8_t, v);
}

constexpr int PSTRF = 64;        // bytes per im2col row (32 f16): 16 dwords -> tr reads conflict-free
constexpr int DSTRF = 64 * 2 + 64;  // bytes per dy row

// BNAPPLY: `dy` is the gradient of the layer's ACTIVATION a = relu(bn(y)) and the batch-norm backward apply
//   dy = A*dz + B*y + C,  dz = da * [relu(bn(y)) > 0]     (coefficients: ocr_bn_bwd_coefficients)
// is computed while the tile is staged — conv1_1 has no input gradient, so the weight gradient is the ONLY reader
// of dy and the separate apply pass (read y, read da, write dy: 3 GiB at 32 x 512^2) plus this kernel's read of dy
// become one read of y and da.
// RECOMPUTE: y is not read either — the tile's y16 is evaluated from the halo the kernel stages anyway (first_mfma:
// 12 MFMAs per wave and tile) into the dy staging rows, where each thread then replaces its own eight chunks by dy.
// The kernel is bound by its HBM streams: one (da) instead of two.
struct FirstBn {
  const half_t* y;
  const float *A, *B, *C, *shift;
  int relu;
  const half_t* wf;        // RECOMPUTE: the forward's packed weights [3][cout][16]
};

template <bool BNAPPLY, bool RECOMPUTE = false>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(FirstP p, const half_t* __restrict__ x4,
                                                               const half_t* __restrict__ dy, FirstBn bn,
                                                               float* __restrict__ partial) {
  static_assert(!RECOMPUTE || BNAPPLY, "y is only needed by the batch-norm apply");
  __shared__ __attribute__((aligned(16))) char halo[10 * HW * 8];
  __shared__ __attribute__((aligned(16))) char patch[256 * PSTRF];
  __shared__ __attribute__((aligned(16))) char dyt[256 * DSTRF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int q = li >> 2, pp = li & 3, hh = g >> 1, gc = g & 1;
  const int co0 = blockIdx.y * 64;
  const int cow = wave & 1, kw = wave >> 1;   // 2 co tiles x 2 K halves
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int a_lane = (8 * hh + q) * PSTRF + (16 * gc + 4 * pp) * 2;
  const int b_lane = (8 * hh + q) * DSTRF + (cow * 32 + 16 * gc + 4 * pp) * 2;

  // the dy tile (32 KB, the stream that bounds this kernel) of tile t+1 is fetched into registers
  // while tile t's patch rows are built and multiplied
  u32x4 dreg[8];
  u32x4 yreg[BNAPPLY && !RECOMPUTE ? 8 : 1];
  u32x4 yrec[RECOMPUTE ? 8 : 1];
  constexpr int YSTRF = 64 * 2 + 16;
  half8_t wa[RECOMPUTE ? 3 : 1][2];
  if constexpr (RECOMPUTE) first_weights(bn.wf, p.cout, co0, lane, wa);
  float cA[8], cB[8], cC[8], cS[8];
  if (BNAPPLY) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = co0 + (tid & 7) * 8 + e;         // this thread's 16-byte chunk of every staged row
      cA[e] = bn.A[cc]; cB[e] = bn.B[cc]; cC[e] = bn.C[cc]; cS[e] = bn.shift[cc];
    }
  }
  auto load_dy = [&](int mt) {
    const int txi = mt % p.tiles_x;
    const int tmp = mt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int img = tmp / p.tiles_y;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 256 + tid;
      const int px = idx >> 3, c = idx & 7;
      const int oy = tyi * TILE_H + (px >> 5), ox = txi * TILE_W + (px & 31);
      dreg[u] = u32x4{0u, 0u, 0u, 0u};
      if (BNAPPLY && !RECOMPUTE) yreg[u] = u32x4{0u, 0u, 0u, 0u};
      if (oy < p.h && ox < p.w) {
        const size_t off = (((size_t)img * p.h + oy) * p.w + ox) * p.cout + co0 + c * 8;
        dreg[u] = *reinterpret_cast<const u32x4*>(dy + off);
        if (BNAPPLY && !RECOMPUTE) yreg[u] = *reinterpret_cast<const u32x4*>(bn.y + off);
      }
    }
  };
  // (tiles are whole multiples of the image here or zero padded: a pixel outside the image stages dy = 0)
  auto staged = [&](int u, bool inside, u32x4 yv) -> u32x4 {
    if (!BNAPPLY) return dreg[u];
    if (!inside) return u32x4{0u, 0u, 0u, 0u};
    const half8_t g8 = __builtin_bit_cast(half8_t, dreg[u]);
    const half8_t y8 = __builtin_bit_cast(half8_t, yv);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float yf = (float)y8[e];
      const bool pass = !bn.relu || __builtin_fmaf(yf, cA[e], cS[e]) > OCR_RELU_TIE;   // the stored activation is positive
      const float dz = pass ? (float)g8[e] : 0.f;
      o[e] = (half_t)__builtin_fmaf(cA[e], dz, __builtin_fmaf(cB[e], yf, cC[e]));
    }
    return __builtin_bit_cast(u32x4, o);
  };
  if ((int)blockIdx.x < p.m_tiles) load_dy(blockIdx.x);
  for (int mt = blockIdx.x; mt < p.m_tiles; mt += gridDim.x) {
    const int txi = mt % p.tiles_x;
    const int tmp = mt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int img = tmp / p.tiles_y;
    __syncthreads();
    load_halo4(x4, halo, img, p.h, p.w, tyi * TILE_H - 1, txi * TILE_W - 1);
    if constexpr (RECOMPUTE) {
      __syncthreads();
      f32x16 yacc[2][2];
      first_mfma(halo, wa, wave, lane, yacc);
      const int r = lane & 31, h5 = lane >> 5;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)yacc[i][t][qq * 4 + e];
            // (a layout of its own inside the dy staging area: 144-byte rows, the conflict-free pitch of conv_epilogue)
            *reinterpret_cast<half4_t*>(dyt + ((wave * 2 + t) * 32 + r) * YSTRF + (i * 32 + qq * 8 + h5 * 4) * 2) = o;
          }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = u * 256 + tid;
        yrec[u] = *reinterpret_cast<const u32x4*>(dyt + (idx >> 3) * YSTRF + (idx & 7) * 16);
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 256 + tid;
      const int px = idx >> 3;
      const bool inside = tyi * TILE_H + (px >> 5) < p.h && txi * TILE_W + (px & 31) < p.w;
      u32x4 yv = u32x4{0u, 0u, 0u, 0u};
      if constexpr (RECOMPUTE) yv = yrec[u];
      else if constexpr (BNAPPLY) yv = yreg[u];
      *reinterpret_cast<u32x4*>(dyt + px * DSTRF + (idx & 7) * 16) = staged(u, inside, yv);
    }
    if (mt + (int)gridDim.x < p.m_tiles) load_dy(mt + gridDim.x);
    __syncthreads();
    {  // im2col row of this thread's pixel: 27 values, k = (ky*3+kx)*3 + c, zero padded to 32
      const int ty = tid >> 5, tx = tid & 31;
      half_t row[32];
#pragma unroll
      for (int k = 27; k < 32; ++k) row[k] = (half_t)0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          half4_t xv = *reinterpret_cast<const half4_t*>(halo + ((ty + ky) * HW + tx + kx) * 8);
#pragma unroll
          for (int c = 0; c < 3; ++c) row[(ky * 3 + kx) * 3 + c] = xv[c];
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        half8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = row[k * 8 + e];
        *reinterpret_cast<half8_t*>(patch + tid * PSTRF + k * 16) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k0 = (kw * 8 + s) * 16;     // this wave's 16-pixel k-step
      half8_t a = tr_pair_f(patch + a_lane + k0 * PSTRF, 4 * PSTRF);
      half8_t b = tr_pair_f(dyt + b_lane + k0 * DSTRF, 4 * DSTRF);
      acc = OCR_MFMA_32x32x16(a, b, acc, 0, 0, 0);
    }
  }
  // partial[blk][kw][27][cout]: the two K halves are separate partial rows
  const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h2;
    if (row < 27)
      partial[(((size_t)blockIdx.x * 2 + kw) * 27 + row) * p.cout + co0 + cow * 32 + r] = acc[e];
  }
}

// ---- batch-norm statistics of conv1_1 WITHOUT evaluating the convolution (round 5) ---------------------------------
// y_c(p) = w_c . v(p) with v(p) the 27-value patch of pixel p (zero outside the image), so over all pixels
//   sum y_c = w_c . m,   sum y_c^2 = w_c^T M w_c,   m = sum_p v(p),  M = sum_p v(p) v(p)^T
// — 28 x 28 numbers that depend on the IMAGES only.  The statistics pass evaluated the 64-channel convolution just to
// round, sum and square 537 M outputs (VALU-bound: 262 us at 32 x 512^2, plus 30 us to finalise its 32 768 partial
// rows); M is one 32x32x16 MFMA per 16 pixels with the SAME fragment as both operands (the im2col rows the weight
// gradient builds, column 27 = 1: row 27 of M is m, M[27][27] the pixel count).  f32 accumulation over a workgroup's
// ~8 K pixels, f64 across workgroups and in the quadratic form; the sums are those of the f32 conv outputs (the reference's
// fused batch norm: nets/vgg.py:14 under slim.batch_norm), not of their 16-bit roundings: NOT bit-identical to the
// evaluating pass.  Accuracy against float64 on the same operands, measured: variance within 1e-5 relative on centred or
// uncorrelated images, and within 7.6e-6 on the worst case built for it — 0..255 images with no mean subtracted, smooth,
// every filter zero-sum, so that w^T M w is the difference of terms 3.9e3 times its size
// (tests/test_gpu_layers.py::test_first_conv_moments_on_uncentred_correlated_images_and_zero_sum_filters; the evaluating
// pass is at 1.0e-5 there: 2 048 pixels per f32 accumulation in that test, 8 192 at the headline batch).
constexpr int FM_WGS = 1024;
__global__ __launch_bounds__(256) void first_moments_kernel(FirstP p, const half_t* __restrict__ x4,
                                                            float* __restrict__ slab) {
  __shared__ __attribute__((aligned(16))) char halo[10 * HW * 8];
  __shared__ __attribute__((aligned(16))) char patch[256 * PSTRF];
  __shared__ float red[4][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int q = li >> 2, pp = li & 3, hh = g >> 1, gc = g & 1;
  const int a_lane = (8 * hh + q) * PSTRF + (16 * gc + 4 * pp) * 2;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int mt = blockIdx.x; mt < p.m_tiles; mt += gridDim.x) {
    const int txi = mt % p.tiles_x;
    const int tmp = mt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int img = tmp / p.tiles_y;
    __syncthreads();
    load_halo4(x4, halo, img, p.h, p.w, tyi * TILE_H - 1, txi * TILE_W - 1);
    __syncthreads();
    {  // im2col row of this thread's pixel: 27 values, k = (ky*3+kx)*3 + c, then the constant 1; all zero outside the image
      const int ty = tid >> 5, tx = tid & 31;
      const bool inside = tyi * TILE_H + ty < p.h && txi * TILE_W + tx < p.w;
      half_t row[32];
#pragma unroll
      for (int k = 27; k < 32; ++k) row[k] = (half_t)0.f;
      row[27] = inside ? (half_t)1.f : (half_t)0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          half4_t xv = *reinterpret_cast<const half4_t*>(halo + ((ty + ky) * HW + tx + kx) * 8);
#pragma unroll
          for (int c = 0; c < 3; ++c) row[(ky * 3 + kx) * 3 + c] = inside ? xv[c] : (half_t)0.f;
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        half8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = row[k * 8 + e];
        *reinterpret_cast<half8_t*>(patch + tid * PSTRF + k * 16) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k0 = (wave * 4 + s) * 16;   // this wave's 16-pixel k-step
      half8_t a = tr_pair_f(patch + a_lane + k0 * PSTRF, 4 * PSTRF);
      acc = OCR_MFMA_32x32x16(a, a, acc, 0, 0, 0);
    }
  }
  // the four waves' quarters meet in LDS (fixed order), one [32][32] block per workgroup
  const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * h2) * 32 + r] = acc[e];
  __syncthreads();
  for (int i = tid; i < 1024; i += 256)
    slab[(size_t)blockIdx.x * 1024 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// slab [blocks][1024] -> stage [32][1024] f64 (fixed order)
__global__ __launch_bounds__(256) void first_moments_stage_kernel(const float* __restrict__ slab, int blocks,
                                                                  double* __restrict__ stage) {
  const int per = (blocks + 31) / 32;
  const int b0 = blockIdx.x * per, b1 = b0 + per < blocks ? b0 + per : blocks;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    double a = 0.0;
    for (int b = b0; b < b1; ++b) a += (double)slab[(size_t)b * 1024 + i];
    stage[(size_t)blockIdx.x * 1024 + i] = a;
  }
}
// stage -> M; per channel (sum y, sum y^2) with the PACKED (16-bit) weights the convolution multiplies by -> one partial
// row [2][cout] for ocr_bn_finalize (T = 1)
__global__ __launch_bounds__(256) void first_moments_finish_kernel(const double* __restrict__ stage,
                                                                   const half_t* __restrict__ wf, int cout,
                                                                   float* __restrict__ row, double* __restrict__ m_out) {
  __shared__ double M[32 * 32];
  for (int i = threadIdx.x; i < 1024; i += 256) {
    double a = 0.0;
    for (int b = 0; b < 32; ++b) a += stage[(size_t)b * 1024 + i];
    M[i] = a;
    if (m_out != nullptr) m_out[i] = a;       // kept for the weight gradient (first_wgrad_sums_kernel)
  }
  __syncthreads();
  // thread = (channel of a group of 64, one of four row lanes): rows i = lane, lane + 4, ... of the quadratic form, the
  // lanes' parts added in lane order (one thread per channel walked all 27 x 27 products alone: 39 us on one workgroup)
  __shared__ double part[2][4][64];
  for (int c0 = 0; c0 < cout; c0 += 64) {
    const int co = c0 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    double s1 = 0.0, s2 = 0.0;
    if (co < cout) {
      double wv[27];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int c = 0; c < 3; ++c) wv[(ky * 3 + kx) * 3 + c] = (double)(float)wf[((size_t)(ky * cout + co)) * 16 + kx * 4 + c];
      for (int i = rl; i < 27; i += 4) {
        s1 += wv[i] * M[27 * 32 + i];
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 27; ++j) t += M[i * 32 + j] * wv[j];
        s2 += wv[i] * t;
      }
    }
    part[0][rl][threadIdx.x & 63] = s1;
    part[1][rl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (threadIdx.x < 128) {
      const int k = threadIdx.x >> 6, cc = threadIdx.x & 63;
      if (c0 + cc < cout)
        row[k * cout + c0 + cc] = (float)(((part[k][0][cc] + part[k][1][cc]) + part[k][2][cc]) + part[k][3][cc]);
    }
    __syncthreads();
  }
}

// conv1_1's weight gradient from sums (round 5, second half).  With dy = A dz + B y + C per channel (the batch-norm
// backward apply: ocr_bn_bwd_coefficients), V the 27-value patches, y = V W:
//   dW = V^T dy = A .* (V^T dz) + B .* (V^T V) W + C .* (V^T 1) = A .* S1 + B .* (M W) + C .* m
// S1 arrives as one [32 slots][64] block per workgroup of conv1_2's input-gradient launch (conv_c64_persist_kernel,
// epilogue mode 7; slot ky*10 + kx*3 + c), M (rows / columns k = (ky*3+kx)*3 + c, row 27 = m) from
// first_moments_finish_kernel; W = the packed 16-bit weights the forward multiplies by.  One block per k: thread =
// (one of 4 block lanes, channel), blocks summed in index order per lane (eight loads in flight), lanes in lane order.
__global__ __launch_bounds__(256) void first_wgrad_sums_kernel(const float* __restrict__ s1, int blocks,
                                                               const double* __restrict__ M,
                                                               const half_t* __restrict__ wf,
                                                               const float* __restrict__ A, const float* __restrict__ B,
                                                               const float* __restrict__ C, float* __restrict__ dw) {
  __shared__ double red[2][4][64];
  __shared__ double Mrow[32];
  const int k = blockIdx.x;                      // (ky*3 + kx)*3 + c
  const int ky = k / 9, kx = (k / 3) % 3, ch = k % 3;
  const int slot = ky * 10 + kx * 3 + ch;
  const int c = threadIdx.x & 63, bl = threadIdx.x >> 6;
  if (threadIdx.x < 32) Mrow[threadIdx.x] = M[k * 32 + threadIdx.x];
  // this lane's share of (M W)[k][c]: columns j = bl, bl + 4, ... (the loads go out before the block sums')
  float wj[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int j = bl + 4 * u;
    const int jy = j / 9, jx = (j / 3) % 3, jc = j % 3;
    wj[u] = j < 27 ? (float)wf[((size_t)(jy * 64 + c)) * 16 + jx * 4 + jc] : 0.f;
  }
  double acc = 0.0;
  for (int b0 = bl; b0 < blocks; b0 += 64) {     // sixteen loads in flight, added in block order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int b = b0 + 4 * u;
      v[u] = b < blocks ? s1[((size_t)b * 32 + slot) * 64 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (double)v[u];
  }
  __syncthreads();
  double mw = 0.0;
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int j = bl + 4 * u;
    if (j < 27) mw += Mrow[j] * (double)wj[u];
  }
  red[0][bl][c] = acc;
  red[1][bl][c] = mw;
  __syncthreads();
  if (threadIdx.x < 64) {
    const double S1 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    const double MW = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    dw[k * 64 + c] = (float)((double)A[c] * S1 + (double)B[c] * MW + (double)C[c] * M[27 * 32 + k]);
  }
}

__global__ void first_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                    int elems, int blocks) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float a = 0.f;
  for (int b = 0; b < blocks; ++b) a += partial[(size_t)b * elems + i];
  dw[i] = a;
}

// HWIO f32 [3,3,3,cout] -> [3][cout][16] f16 (k = kx*4 + c)
__global__ void pack_first_kernel(const float* __restrict__ w, half_t* __restrict__ out, int cout) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * cout * 16) return;
  const int k = i & 15, co = (i >> 4) % cout, ky = (i >> 4) / cout;
  const int kx = k >> 2, c = k & 3;
  float v = (kx < 3 && c < 3) ? w[((ky * 3 + kx) * 3 + c) * cout + co] : 0.f;
  out[i] = (half_t)v;
}

// two workgroups per CU (~70 KB of LDS each) cover each other's barrier stalls: 32 x 512^2 0.31 -> 0.21 ms,
// the time the dy stream alone takes from HBM
int wgrad_blocks(int m_tiles) { return m_tiles < 512 ? m_tiles : 512; }

int fill(FirstP* p, int n, int h, int w, int cout, int flags) {
  OCR_CHECK_ARG(n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(cout % 64 == 0);
  p->n = n; p->h = h; p->w = w; p->cout = cout; p->flags = flags;
  p->tiles_x = ocr_cdiv(w, TILE_W);
  p->tiles_y = ocr_cdiv(h, TILE_H);
  p->m_tiles = n * p->tiles_x * p->tiles_y;
  return OCR_OK;
}

}  // namespace

extern "C" int ocr_conv2d_first_num_mtiles(int n, int h, int w) {
  return n * ocr_cdiv(w, TILE_W) * ocr_cdiv(h, TILE_H);
}

extern "C" int ocr_pack_weights_first_f16(const void* w_hwio_f32, int cout, void* w_first,
                                          void* stream) {
  OCR_CHECK_ARG(w_hwio_f32 && w_first && cout > 0);
  hipLaunchKernelGGL(pack_first_kernel, dim3(ocr_cdiv(3 * cout * 16, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(w_hwio_f32),
                     static_cast<half_t*>(w_first), cout);
  return ocr_launch_status();
}

extern "C" int ocr_conv2d_first_f16(int n, int h, int w, int cout, const void* x4,
                                    const void* w_first, const void* bias, int flags, void* y,
                                    void* stats, void* stream) {
  FirstP p;
  int rc = fill(&p, n, h, w, cout, flags);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && w_first);
  OCR_CHECK_ARG(y || (flags & OCR_CONV_STATS));          // y == NULL: a statistics-only launch (nothing is stored)
  OCR_CHECK_ARG(!(flags & OCR_CONV_BIAS) || bias);
  OCR_CHECK_ARG(!(flags & OCR_CONV_STATS) || stats);
  const size_t lds = conv_epilogue_lds(64);
  hipLaunchKernelGGL(conv_first_kernel<false>, dim3((unsigned)(p.m_tiles * (cout / 64))), dim3(256), lds,
                     static_cast<hipStream_t>(stream), p, static_cast<const half_t*>(x4),
                     static_cast<const half_t*>(w_first), static_cast<const float*>(bias),
                     static_cast<half_t*>(y), static_cast<float*>(stats), FirstAct{});
  return ocr_launch_status();
}

extern "C" int ocr_conv2d_first_bn_relu_f16(int n, int h, int w, int cout, const void* x4, const void* w_first,
                                            const void* scale, const void* shift, int relu, void* a,
                                            void* stream) {
  FirstP p;
  int rc = fill(&p, n, h, w, cout, 0);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && w_first && scale && shift && a);
  const size_t lds = conv_epilogue_lds(64);
  hipLaunchKernelGGL(conv_first_kernel<true>, dim3((unsigned)(p.m_tiles * (cout / 64))), dim3(256), lds,
                     static_cast<hipStream_t>(stream), p, static_cast<const half_t*>(x4),
                     static_cast<const half_t*>(w_first), (const float*)nullptr, static_cast<half_t*>(a),
                     (float*)nullptr, FirstAct{static_cast<const float*>(scale), static_cast<const float*>(shift), relu});
  return ocr_launch_status();
}

extern "C" size_t ocr_conv2d_first_wgrad_workspace(int n, int h, int w, int cout) {
  const int mt = ocr_conv2d_first_num_mtiles(n, h, w);
  return (size_t)wgrad_blocks(mt) * 2 * 27 * cout * sizeof(float);
}

extern "C" int ocr_conv2d_first_wgrad_f16(int n, int h, int w, int cout, const void* x4,
                                          const void* dy, void* dw, void* workspace,
                                          size_t ws_bytes, void* stream) {
  FirstP p;
  int rc = fill(&p, n, h, w, cout, 0);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && dy && dw && workspace);
  if (ws_bytes < ocr_conv2d_first_wgrad_workspace(n, h, w, cout)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = wgrad_blocks(p.m_tiles);
  hipLaunchKernelGGL(conv_first_wgrad_kernel<false>, dim3(blocks, cout / 64), dim3(256), 0, st, p,
                     static_cast<const half_t*>(x4), static_cast<const half_t*>(dy), FirstBn{},
                     static_cast<float*>(workspace));
  const int elems = 27 * cout;
  hipLaunchKernelGGL(ocr_sum_rows_kernel, dim3(sum_rows_grid(elems)), dim3(256), 0, st,
                     static_cast<const float*>(workspace), static_cast<float*>(dw), elems, blocks * 2, 1.f);
  return ocr_launch_status();
}

extern "C" int ocr_conv2d_first_wgrad_bn_f16(int n, int h, int w, int cout, const void* x4, const void* da,
                                             const void* bn_y, const void* w_first, const void* bn_shift,
                                             const void* coef_a, const void* coef_b, const void* coef_c, int relu,
                                             void* dw, void* workspace, size_t ws_bytes, void* stream) {
  FirstP p;
  int rc = fill(&p, n, h, w, cout, 0);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && da && (bn_y || w_first) && bn_shift && coef_a && coef_b && coef_c && dw && workspace);
  if (ws_bytes < ocr_conv2d_first_wgrad_workspace(n, h, w, cout)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = wgrad_blocks(p.m_tiles);
  FirstBn bn{static_cast<const half_t*>(bn_y), static_cast<const float*>(coef_a), static_cast<const float*>(coef_b),
             static_cast<const float*>(coef_c), static_cast<const float*>(bn_shift), relu,
             static_cast<const half_t*>(w_first)};
  if (w_first != nullptr)
    hipLaunchKernelGGL((conv_first_wgrad_kernel<true, true>), dim3(blocks, cout / 64), dim3(256), 0, st, p,
                       static_cast<const half_t*>(x4), static_cast<const half_t*>(da), bn,
                       static_cast<float*>(workspace));
  else
    hipLaunchKernelGGL((conv_first_wgrad_kernel<true, false>), dim3(blocks, cout / 64), dim3(256), 0, st, p,
                       static_cast<const half_t*>(x4), static_cast<const half_t*>(da), bn,
                       static_cast<float*>(workspace));
  const int elems = 27 * cout;
  hipLaunchKernelGGL(ocr_sum_rows_kernel, dim3(sum_rows_grid(elems)), dim3(256), 0, st,
                     static_cast<const float*>(workspace), static_cast<float*>(dw), elems, blocks * 2, 1.f);
  return ocr_launch_status();
}

// Batch-norm statistics of conv1_1 from the image moments (first_moments_kernel above): row [2][cout] = (sum y, sum y^2)
// over the n*h*w outputs, for ocr_bn_finalize with T = 1.  workspace: ocr_conv2d_first_moments_workspace() bytes.
extern "C" size_t ocr_conv2d_first_moments_workspace(void) { return (size_t)FM_WGS * 1024 * sizeof(float) + 32 * 1024 * sizeof(double); }
extern "C" int ocr_conv2d_first_moments_keep_f16(int n, int h, int w, int cout, const void* x4, const void* w_first,
                                                 void* stats_row, void* moments_f64, void* workspace, size_t ws_bytes,
                                                 void* stream);
extern "C" int ocr_conv2d_first_moments_f16(int n, int h, int w, int cout, const void* x4, const void* w_first,
                                            void* stats_row, void* workspace, size_t ws_bytes, void* stream) {
  return ocr_conv2d_first_moments_keep_f16(n, h, w, cout, x4, w_first, stats_row, nullptr, workspace, ws_bytes, stream);
}
// ... and with the moments themselves kept: moments_f64 [32][32] doubles (rows / columns k = (ky*3+kx)*3 + c, row 27 =
// the patch sums, [27][27] = the pixel count), what ocr_conv2d_first_wgrad_sums_f32 needs in the backward pass.
extern "C" int ocr_conv2d_first_moments_keep_f16(int n, int h, int w, int cout, const void* x4, const void* w_first,
                                                 void* stats_row, void* moments_f64, void* workspace, size_t ws_bytes,
                                                 void* stream) {
  FirstP p;
  int rc = fill(&p, n, h, w, cout, 0);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && w_first && stats_row && workspace);
  if (ws_bytes < ocr_conv2d_first_moments_workspace() || ((uintptr_t)workspace & 7)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = p.m_tiles < FM_WGS ? p.m_tiles : FM_WGS;
  float* slab = static_cast<float*>(workspace);
  double* stage = reinterpret_cast<double*>(static_cast<char*>(workspace) + (size_t)FM_WGS * 1024 * sizeof(float));
  hipLaunchKernelGGL(first_moments_kernel, dim3(blocks), dim3(256), 0, st, p, static_cast<const half_t*>(x4), slab);
  hipLaunchKernelGGL(first_moments_stage_kernel, dim3(32), dim3(256), 0, st, slab, blocks, stage);
  OCR_CHECK_ARG(((uintptr_t)moments_f64 & 7) == 0);
  hipLaunchKernelGGL(first_moments_finish_kernel, dim3(1), dim3(256), 0, st, stage, static_cast<const half_t*>(w_first),
                     cout, static_cast<float*>(stats_row), static_cast<double*>(moments_f64));
  return ocr_launch_status();
}

// conv1_1's weight gradient dw [3,3,3,64] f32 from the S1 blocks of ocr_conv2d_bnred_first_wgrad_f16 ([blocks][32][64]
// f32), the kept moments and the batch-norm backward coefficients (ocr_bn_bwd_coefficients): first_wgrad_sums_kernel.
extern "C" int ocr_conv2d_first_wgrad_sums_f32(const void* s1_blocks, int blocks, const void* moments_f64,
                                               const void* w_first, int cout, const void* coef_a, const void* coef_b,
                                               const void* coef_c, void* dw, void* stream) {
  OCR_CHECK_ARG(s1_blocks && blocks > 0 && moments_f64 && w_first && coef_a && coef_b && coef_c && dw);
  OCR_CHECK_SHAPE(cout == 64);
  hipLaunchKernelGGL(first_wgrad_sums_kernel, dim3(27), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(s1_blocks), blocks, static_cast<const double*>(moments_f64),
                     static_cast<const half_t*>(w_first), static_cast<const float*>(coef_a),
                     static_cast<const float*>(coef_b), static_cast<const float*>(coef_c), static_cast<float*>(dw));
  return ocr_launch_status();
}
