// ResNet-v1 root block convolution: 7x7, stride 2, cin = 3 -> 64 with the explicit (3,3)
// padding of resnet_utils.conv2d_same (reference nets/resnet_v1.py:193,
// nets/resnet_utils.py:111-122), and its weight gradient.  Same scheme as conv_first.hip: the
// image is [n,h,w,4] f16 (one pixel = 8 bytes); per kernel row ky the K dimension is 8 pixels
// (kx = 0..7, the 8th with zero weights) x 4 channels = two MFMA k-steps, and a lane's 8
// k-values are two adjacent input pixels = 16 contiguous bytes of the LDS halo tile.
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int SHW = 72;            // halo row pitch in pixels: (32-1)*2 + 8 = 70 used
constexpr int SHH = 21;            // (8-1)*2 + 7

struct StemP {
  int n, h, w, oh, ow, cout, tiles_x, tiles_y, m_tiles, flags;
};

// All of a thread's halo pixels are requested before the first is stored (an out-of-image position reads pixel 0 and is
// replaced by zero): as a load -> ds_write loop hipcc kept one request in flight per thread, six round trips per tile.
constexpr int kHaloPerThread = (SHH * SHW + 255) / 256;
__device__ __forceinline__ void load_halo_stem(const half_t* __restrict__ x4, char* halo, int img,
                                               int h, int w, int iy0, int ix0) {
  u32x2 v[kHaloPerThread];
  bool inside[kHaloPerThread];
#pragma unroll
  for (int j = 0; j < kHaloPerThread; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int hy = i / SHW, hx = i - hy * SHW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    inside[j] = i < SHH * SHW && iy >= 0 && iy < h && ix >= 0 && ix < w;
    const size_t off = inside[j] ? (((size_t)img * h + iy) * w + ix) * 4 : 0;
    v[j] = *reinterpret_cast<const u32x2*>(x4 + off);
  }
#pragma unroll
  for (int j = 0; j < kHaloPerThread; ++j) {
    const int i = threadIdx.x + 256 * j;
    if (i < SHH * SHW) *reinterpret_cast<u32x2*>(halo + i * 8) = inside[j] ? v[j] : u32x2{0u, 0u};
  }
}

// w_stem [7][2][cout][16] f16: k = (kx - 4*khalf)*4 + c, zero for kx = 7 or c = 3
__global__ __launch_bounds__(256) void conv_stem_kernel(StemP p, const half_t* __restrict__ x4,
                                                        const half_t* __restrict__ ws,
                                                        const float* __restrict__ bias,
                                                        half_t* __restrict__ y,
                                                        float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int n_tiles = p.cout / 64;
  const int nt = blockIdx.x % n_tiles;
  const int mt = blockIdx.x / n_tiles;
  const int txi = mt % p.tiles_x;
  const int tmp = mt / p.tiles_x;
  const int tyi = tmp % p.tiles_y;
  const int img = tmp / p.tiles_y;
  const int co0 = nt * 64;

  // the 28 weight fragments of this lane (7 kernel rows x 2 halves x 2 cout tiles, 112 registers) are requested FIRST and
  // land under the halo fetch: loaded where they are used, each MFMA pair waited for its own fragment (L2 latency x 14)
  half8_t wa[7][2][2];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        wa[ky][kh][i] = *reinterpret_cast<const half8_t*>(
            ws + ((size_t)((ky * 2 + kh) * p.cout + co0 + i * 32 + r)) * 16 + 8 * hh);
  load_halo_stem(x4, halo, img, p.h, p.w, tyi * TILE_H * 2 - 3, txi * TILE_W * 2 - 3);
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;

#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ty = wave * 2 + t;
        const char* src = halo + ((2 * ty + ky) * SHW + 2 * r + 4 * kh + 2 * hh) * 8;
        half4_t lo = *reinterpret_cast<const half4_t*>(src);
        half4_t hi = *reinterpret_cast<const half4_t*>(src + 8);
        half8_t b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][t] = OCR_MFMA_32x32x16(wa[ky][kh][i], b, acc[i][t], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  conv_epilogue<64, 2, 2, 1>(acc, smem, p.flags, bias, y, stats, img, tyi, txi, mt, co0, p.oh, p.ow,
                             p.cout, 0, wave, true);
}

// ---- weight gradient: per ky a [256 px][32] im2col slab (kx*4 + c) in LDS, both MFMA operands
// through ds_read_b64_tr_b16; waves = 2 cout tiles x 2 halves of the tile's pixels.
typedef short short4v_s __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) short4v_s* lds_s4_ptr_s;

__device__ __forceinline__ half8_t tr_pair_s(const char* base, int second_off) {
  short4v_s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr_s)(base));
  short4v_s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr_s)(base + second_off));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8_t, v);
}

constexpr int PSTRS = 64;             // bytes per im2col row (32 f16)
constexpr int DSTRS = 64 * 2 + 64;    // bytes per dy row

// BNAPPLY: as conv_first_wgrad_kernel<true> (conv_first.hip) — the root convolution has no input gradient, its weight
// gradient is the only reader of dy, so the BN-backward apply dy = A*dz + B*y + C, dz = da * [relu(bn(y)) > 0], is
// computed while the dy tile is staged and the apply pass (3 x 839 MB at 64 x 640^2) and the dy tensor disappear.
struct StemBn {
  const half_t* y;
  const float *A, *B, *C, *shift;
  int relu;
};

template <bool BNAPPLY>
__global__ __launch_bounds__(256, 2) void conv_stem_wgrad_kernel(StemP p, const half_t* __restrict__ x4,
                                                              const half_t* __restrict__ dy, StemBn bn,
                                                              float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) char halo[SHH * SHW * 8];
  __shared__ __attribute__((aligned(16))) char patch[256 * PSTRS];
  __shared__ __attribute__((aligned(16))) char dyt[256 * DSTRS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int q = li >> 2, pp = li & 3, hh = g >> 1, gc = g & 1;
  const int co0 = blockIdx.y * 64;
  const int cow = wave & 1, kw = wave >> 1;
  f32x16 acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  const int a_lane = (8 * hh + q) * PSTRS + (16 * gc + 4 * pp) * 2;
  const int b_lane = (8 * hh + q) * DSTRS + (cow * 32 + 16 * gc + 4 * pp) * 2;
  float cA[8], cB[8], cC[8], cS[8];
  if (BNAPPLY) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = co0 + (tid & 7) * 8 + e;         // this thread's 16-byte chunk of every staged row
      cA[e] = bn.A[cc]; cB[e] = bn.B[cc]; cC[e] = bn.C[cc]; cS[e] = bn.shift[cc];
    }
  }

  for (int mt = blockIdx.x; mt < p.m_tiles; mt += gridDim.x) {
    const int txi = mt % p.tiles_x;
    const int tmp = mt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int img = tmp / p.tiles_y;
    __syncthreads();
    // the tile's dy (and y) rows in two batches of four rows per thread, the halo's requests behind the first: every
    // request of a batch goes out before the first is used (a position outside the map reads element 0 and is stored
    // as zero) — under `if (inside)` hipcc waited for each dy / y pair before requesting the next: 8 + 6 round trips per
    // tile, which is what this kernel's time was.  (All sixteen at once cost the second resident workgroup: 276 registers.)
#pragma unroll
    for (int ub = 0; ub < 8; ub += 4) {
      u32x4 gv[4];
      half8_t yv8[4];
      bool okv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = (ub + k) * 256 + tid;
        const int px = idx >> 3, c = idx & 7;
        const int oy = tyi * TILE_H + (px >> 5), ox = txi * TILE_W + (px & 31);
        okv[k] = oy < p.oh && ox < p.ow;
        const size_t off = okv[k] ? (((size_t)img * p.oh + oy) * p.ow + ox) * p.cout + co0 + c * 8 : 0;
        gv[k] = *reinterpret_cast<const u32x4*>(dy + off);
        if (BNAPPLY) yv8[k] = *reinterpret_cast<const half8_t*>(bn.y + off);
      }
      if (ub == 0) load_halo_stem(x4, halo, img, p.h, p.w, tyi * TILE_H * 2 - 3, txi * TILE_W * 2 - 3);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = (ub + k) * 256 + tid;
        const int px = idx >> 3, c = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (okv[k]) {
          v = gv[k];
          if (BNAPPLY) {
            const half8_t g8 = __builtin_bit_cast(half8_t, v);
            const half8_t y8 = yv8[k];
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float yf = (float)y8[e];
              const bool pass = !bn.relu || __builtin_fmaf(yf, cA[e], cS[e]) > OCR_RELU_TIE;
              const float dz = pass ? (float)g8[e] : 0.f;
              o[e] = (half_t)__builtin_fmaf(cA[e], dz, __builtin_fmaf(cB[e], yf, cC[e]));
            }
            v = __builtin_bit_cast(u32x4, o);
          }
        }
        *reinterpret_cast<u32x4*>(dyt + px * DSTRS + c * 16) = v;
      }
    }
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      __syncthreads();   // halo/dy staged (ky = 0) or previous slab consumed
      {
        const int ty = tid >> 5, tx = tid & 31;
        const char* src = halo + ((2 * ty + ky) * SHW + 2 * tx) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          *reinterpret_cast<u32x2*>(patch + tid * PSTRS + k * 8) = *reinterpret_cast<const u32x2*>(src + k * 8);
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k0 = (kw * 8 + s) * 16;
        half8_t a = tr_pair_s(patch + a_lane + k0 * PSTRS, 4 * PSTRS);
        half8_t b = tr_pair_s(dyt + b_lane + k0 * DSTRS, 4 * DSTRS);
        acc[ky] = OCR_MFMA_32x32x16(a, b, acc[ky], 0, 0, 0);
      }
    }
  }
  // partial[blk][kw][147][cout]; row = (ky*7 + kx)*3 + c for kx < 7, c < 3
  const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h2;    // kx*4 + c
      const int kx = row >> 2, c = row & 3;
      if (kx < 7 && c < 3)
        partial[(((size_t)blockIdx.x * 2 + kw) * 147 + (ky * 7 + kx) * 3 + c) * p.cout + co0 + cow * 32 + r] =
            acc[ky][e];
    }
}

// HWIO f32 [7,7,3,cout] -> [7][2][cout][16] f16
__global__ void pack_stem_kernel(const float* __restrict__ w, half_t* __restrict__ out, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 14 * cout * 16) return;
  const int k = i & 15, co = (i >> 4) % cout, kk = (i >> 4) / cout;
  const int ky = kk >> 1, kh = kk & 1;
  const int kx = 4 * kh + (k >> 2), c = k & 3;
  const float v = (kx < 7 && c < 3) ? w[((ky * 7 + kx) * 3 + c) * cout + co] : 0.f;
  out[i] = (half_t)v;
}

// two workgroups per CU (76 KB of LDS each): a tile is a chain of 14 barriers with 8 MFMAs between them, so a
// second resident workgroup fills the stalls (64 x 640^2: 1.09 -> 0.67 ms; 768 / 1024 blocks: 0.85 / 0.74 ms)
int stem_blocks(int m_tiles) { return m_tiles < 512 ? m_tiles : 512; }

int fill(StemP* p, int n, int h, int w, int cout, int flags) {
  OCR_CHECK_ARG(n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(cout % 64 == 0);
  p->n = n; p->h = h; p->w = w; p->cout = cout; p->flags = flags;
  p->oh = (h - 1) / 2 + 1;          // pad (3,3), kernel 7, stride 2, VALID
  p->ow = (w - 1) / 2 + 1;
  p->tiles_x = ocr_cdiv(p->ow, TILE_W);
  p->tiles_y = ocr_cdiv(p->oh, TILE_H);
  p->m_tiles = n * p->tiles_x * p->tiles_y;
  return OCR_OK;
}

}  // namespace

extern "C" int ocr_conv2d_stem_num_mtiles(int n, int h, int w) {
  return n * ocr_cdiv((w - 1) / 2 + 1, TILE_W) * ocr_cdiv((h - 1) / 2 + 1, TILE_H);
}

extern "C" int ocr_pack_weights_stem_f16(const void* w_hwio_f32, int cout, void* w_stem, void* stream) {
  OCR_CHECK_ARG(w_hwio_f32 && w_stem && cout > 0);
  hipLaunchKernelGGL(pack_stem_kernel, dim3(ocr_cdiv(14 * cout * 16, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(w_hwio_f32),
                     static_cast<half_t*>(w_stem), cout);
  return ocr_launch_status();
}

extern "C" int ocr_conv2d_stem_f16(int n, int h, int w, int cout, const void* x4, const void* w_stem,
                                   const void* bias, int flags, void* y, void* stats, void* stream) {
  StemP p;
  int rc = fill(&p, n, h, w, cout, flags);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && w_stem && y);
  OCR_CHECK_ARG(!(flags & OCR_CONV_BIAS) || bias);
  OCR_CHECK_ARG(!(flags & OCR_CONV_STATS) || stats);
  const size_t lds = conv_epilogue_lds(64);
  hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)(p.m_tiles * (cout / 64))), dim3(256), lds,
                     static_cast<hipStream_t>(stream), p, static_cast<const half_t*>(x4),
                     static_cast<const half_t*>(w_stem), static_cast<const float*>(bias),
                     static_cast<half_t*>(y), static_cast<float*>(stats));
  return ocr_launch_status();
}

extern "C" size_t ocr_conv2d_stem_wgrad_workspace(int n, int h, int w, int cout) {
  return (size_t)stem_blocks(ocr_conv2d_stem_num_mtiles(n, h, w)) * 2 * 147 * cout * sizeof(float);
}

extern "C" int ocr_conv2d_stem_wgrad_f16(int n, int h, int w, int cout, const void* x4, const void* dy,
                                         void* dw, void* workspace, size_t ws_bytes, void* stream) {
  StemP p;
  int rc = fill(&p, n, h, w, cout, 0);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && dy && dw && workspace);
  if (ws_bytes < ocr_conv2d_stem_wgrad_workspace(n, h, w, cout)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = stem_blocks(p.m_tiles);
  hipLaunchKernelGGL(conv_stem_wgrad_kernel<false>, dim3(blocks, cout / 64), dim3(256), 0, st, p,
                     static_cast<const half_t*>(x4), static_cast<const half_t*>(dy), StemBn{},
                     static_cast<float*>(workspace));
  const int elems = 147 * cout;
  hipLaunchKernelGGL(ocr_sum_rows_kernel, dim3(sum_rows_grid(elems)), dim3(256), 0, st,
                     static_cast<const float*>(workspace), static_cast<float*>(dw), elems, blocks * 2, 1.f);
  return ocr_launch_status();
}

extern "C" int ocr_conv2d_stem_wgrad_bn_f16(int n, int h, int w, int cout, const void* x4, const void* da,
                                            const void* bn_y, const void* bn_shift, const void* coef_a,
                                            const void* coef_b, const void* coef_c, int relu, void* dw,
                                            void* workspace, size_t ws_bytes, void* stream) {
  StemP p;
  int rc = fill(&p, n, h, w, cout, 0);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x4 && da && bn_y && bn_shift && coef_a && coef_b && coef_c && dw && workspace);
  if (ws_bytes < ocr_conv2d_stem_wgrad_workspace(n, h, w, cout)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = stem_blocks(p.m_tiles);
  StemBn bn{static_cast<const half_t*>(bn_y), static_cast<const float*>(coef_a), static_cast<const float*>(coef_b),
            static_cast<const float*>(coef_c), static_cast<const float*>(bn_shift), relu};
  hipLaunchKernelGGL(conv_stem_wgrad_kernel<true>, dim3(blocks, cout / 64), dim3(256), 0, st, p,
                     static_cast<const half_t*>(x4), static_cast<const half_t*>(da), bn, static_cast<float*>(workspace));
  const int elems = 147 * cout;
  hipLaunchKernelGGL(ocr_sum_rows_kernel, dim3(sum_rows_grid(elems)), dim3(256), 0, st,
                     static_cast<const float*>(workspace), static_cast<float*>(dw), elems, blocks * 2, 1.f);
  return ocr_launch_status();
}
