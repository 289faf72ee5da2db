// Weight gradient of the NHWC convolution on gfx950 MFMA.
//
//   dw[ky,kx,ci,co] = sum_{n,oy,ox} x[n, oy*s+ky*d-pt, ox*s+kx*d-pl, ci] * dy[n,oy,ox,co]
//
// (tf.gradients of slim.conv2d w.r.t. its weights; reference call site
// multigpu_train.py:129 `opt.compute_gradients`.)
//
// GEMM view: M = ci, N = co, K = output pixels.  Both operands live in HBM as
// [pixel][channel], i.e. K-major, so the MFMA fragments (8 consecutive K per
// lane) are produced with gfx950's transposing LDS read ds_read_b64_tr_b16
// straight from NHWC tiles — no transposed copy of the activations exists.
//
// One workgroup (4 waves) owns a 64(ci) x 64(co) x (kh*kw taps) block of dw and
// a contiguous range of 8x32-pixel tiles (split-K over pixels).  Per tile it
// stages the x halo tile and the dy tile once in LDS and sweeps all taps from
// shifted addresses, keeping one f32 32x32 accumulator per tap in registers.
// Partial blocks go to a [split][tap][ci][co] f32 slab; a second kernel sums the
// slabs in fixed order, so the result is bitwise reproducible (no atomics).
#include "common.h"
#include <stdlib.h>

namespace ocr_detail {   // conv_wgrad_pw.hip: GEMM-tiled path for 1x1 convolutions
int wgrad_pw_splits(const ocr_conv_desc* d);
int wgrad_pw_launch(const ocr_conv_desc* d, const void* x, const void* dy, void* slab, hipStream_t st);
bool wgrad_pw_is_256x256(const ocr_conv_desc* d);     // the 256 x 256-tile instantiation (the one whose register room is tabulated)
}

namespace {

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

struct WgP {
  int n, h, w, cin, oh, ow, cout, kh, kw, stride, dil, pt, pl;
  int tiles_x, tiles_y, m_tiles, HT, WT, splits, tiles_per_split, nci, nco;
};

OCR_DIAG_DECLARE(ocr_diag_wgrad)

constexpr int TILE_H = 8;
constexpr int TILE_W = 32;
constexpr int CIB = 64;
constexpr int COB = 64;
constexpr int XSTR = CIB * 2 + 16;  // bytes per halo pixel
constexpr int DSTR = COB * 2 + 16;  // bytes per dy pixel

__device__ __forceinline__ half8_t tr_pair(const char* base, int second_off) {
  // two transposing reads: k = 0..3 and k = 4..7 of this lane's fragment
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base));
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + second_off));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8_t, v);
}

template <int MAXTAPS>
__global__ __launch_bounds__(512) void wgrad_kernel(WgP p, const half_t* __restrict__ x,
                                                    const half_t* __restrict__ dy,
                                                    float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int halo_px = p.HT * p.WT;
  char* xh = smem;
  char* dyt = smem + halo_px * XSTR;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // 8 waves = 2 (pixel halves of every tile) x 2 (ci) x 2 (co); the two pixel groups leave as
  // separate slab rows
  const int kgrp = wave >> 2;
  const int ciw = (wave >> 1) & 1, cow = wave & 1;
  const int li = lane & 15, g = lane >> 4;
  const int q = li >> 2, pp = li & 3, hh = g >> 1, gc = g & 1;

  int bid = blockIdx.x;
  const int cob = bid % p.nco;
  bid /= p.nco;
  const int cib = bid % p.nci;
  const int split = bid / p.nci;
  const int ci0 = cib * CIB, co0 = cob * COB;
  const int ntaps = p.kh * p.kw;

  f32x16 acc[MAXTAPS];
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int a_lane = ((8 * hh + q) * p.stride) * XSTR + (ciw * 32 + 16 * gc + 4 * pp) * 2;
  const int a_half = 4 * p.stride * XSTR;
  const int b_lane = (8 * hh + q) * DSTR + (cow * 32 + 16 * gc + 4 * pp) * 2;
  const int b_half = 4 * DSTR;

  const int mt_begin = split * p.tiles_per_split;
  int mt_end = mt_begin + p.tiles_per_split;
  if (mt_end > p.m_tiles) mt_end = p.m_tiles;

  for (int mt = mt_begin; mt < mt_end; ++mt) {
    const int txi = mt % p.tiles_x;
    int tmp = mt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int img = tmp / p.tiles_y;
    const int iy0 = tyi * TILE_H * p.stride - p.pt;
    const int ix0 = txi * TILE_W * p.stride - p.pl;
    __syncthreads();
    {  // x halo tile, channels ci0..ci0+63
      const int total = halo_px * (CIB / 8);
      const half_t* xb = x + (size_t)img * p.h * p.w * p.cin + ci0;
      for (int base = 0; base < total; base += 512 * 4) {
        u32x4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * 512 + tid;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          off[u] = -1;
          if (idx < total) {
            int hp = idx >> 3, c = idx & 7;
            int hy = hp / p.WT, hx = hp - hy * p.WT;
            int iy = iy0 + hy, ix = ix0 + hx;
            off[u] = hp * XSTR + c * 16;
            if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w)
              v[u] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * p.w + ix) * p.cin + c * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (off[u] >= 0) *reinterpret_cast<u32x4*>(xh + off[u]) = v[u];
      }
    }
    {  // dy tile [256][64], zero outside the output
      const half_t* db = dy + (size_t)img * p.oh * p.ow * p.cout + co0;
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int idx = u * 512 + tid;
        int px = idx >> 3, c = idx & 7;
        int oy = tyi * TILE_H + (px >> 5), ox = txi * TILE_W + (px & 31);
        v[u] = u32x4{0u, 0u, 0u, 0u};
        if (oy < p.oh && ox < p.ow)
          v[u] = *reinterpret_cast<const u32x4*>(db + ((size_t)oy * p.ow + ox) * p.cout + c * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int idx = u * 512 + tid;
        int px = idx >> 3, c = idx & 7;
        *reinterpret_cast<u32x4*>(dyt + px * DSTR + c * 16) = v[u];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int s = kgrp * 8; s < kgrp * 8 + 8; ++s) {
      const int ty = s >> 1, tx0 = (s & 1) * 16;
      half8_t b = tr_pair(dyt + b_lane + (ty * 32 + tx0) * DSTR, b_half);
      const char* abase = xh + a_lane + ((ty * p.stride) * p.WT + tx0 * p.stride) * XSTR;
#pragma unroll
      for (int t = 0; t < MAXTAPS; ++t) {
        if (t < ntaps) {
          const int ky = t / p.kw, kx = t - ky * p.kw;
          half8_t a = tr_pair(abase + ((ky * p.dil) * p.WT + kx * p.dil) * XSTR, a_half);
          acc[t] = OCR_MFMA_32x32x16(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // partial block -> slab[split][tap][ci][co]
  const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t) {
    if (t < ntaps) {
      float* dst = slab + (((size_t)(split * 2 + kgrp) * ntaps + t) * p.cin + ci0 + ciw * 32) * p.cout +
                   co0 + cow * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h2;
        dst[(size_t)row * p.cout] = acc[t][e];
      }
    }
  }
}

// dw = sum over splits of slab[split], in a FIXED order (bitwise reproducible).  SL lanes share one
// output (each sums splits sl, sl+SL, ... in order, then a fixed-order LDS tree): small weight tensors
// with hundreds of splits (conv1_2: 9216 float4 outputs x 512 slabs) would otherwise run on a few
// dozen workgroups of 512 dependent loads each.
template <int SL>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                          size_t elems4, int splits) {
  constexpr int OUTS = 256 / SL;
  __shared__ f32x4 red[256];
  const int sl = threadIdx.x / OUTS, o = threadIdx.x % OUTS;
  const size_t i = (size_t)blockIdx.x * OUTS + o;
  const f32x4* s = reinterpret_cast<const f32x4*>(slab);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < elems4) {
    // four slabs in flight per lane, added in slab order
    for (int k0 = sl; k0 < splits; k0 += 4 * SL) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * SL;
        v[u] = k < splits ? s[(size_t)k * elems4 + i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) a += v[u];
    }
  }
  if constexpr (SL > 1) {
    red[threadIdx.x] = a;
    __syncthreads();
#pragma unroll
    for (int step = SL / 2; step >= 1; step >>= 1) {
      if (sl < step) red[threadIdx.x] += red[threadIdx.x + step * OUTS];
      __syncthreads();
    }
    a = red[threadIdx.x];
  }
  if (sl == 0 && i < elems4) reinterpret_cast<f32x4*>(dw)[i] = a;
}

static void launch_slab_reduce(const float* slab, float* dw, size_t elems4, int splits, hipStream_t st) {
  // enough workgroups to fill the chip, but never more split lanes than slabs
  int sl = 1;
  while (sl < 64 && sl * 4 <= splits && (elems4 * sl) / 256 < 2048) sl *= 4;
  const unsigned grid = (unsigned)((elems4 * sl + 255) / 256);
  switch (sl) {
    case 1: hipLaunchKernelGGL(slab_reduce_kernel<1>, dim3(grid), dim3(256), 0, st, slab, dw, elems4, splits); break;
    case 4: hipLaunchKernelGGL(slab_reduce_kernel<4>, dim3(grid), dim3(256), 0, st, slab, dw, elems4, splits); break;
    case 16: hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3(grid), dim3(256), 0, st, slab, dw, elems4, splits); break;
    default: hipLaunchKernelGGL(slab_reduce_kernel<64>, dim3(grid), dim3(256), 0, st, slab, dw, elems4, splits); break;
  }
}

// ---------------------------------------------------------------------------
// v2: register-blocked, software-pipelined variant (used whenever its LDS fits).
//
//   block per workgroup : 64 ci x COB co x all taps   (COB = 128, or 64 when cout % 128 != 0)
//   wave tile           : 32 ci x (COB/2) co x taps   -> each shifted x fragment (2 tr-reads)
//                         feeds COB/64 MFMAs, each dy fragment is shared by all taps:
//                         22 LDS reads per 18 MFMAs instead of 20 per 9
//   pixel tile          : 4 x 32 output pixels, DOUBLE-buffered in LDS; the global loads of
//                         tile t+1 are issued before the MFMAs of tile t and written to the
//                         other buffer afterwards (one barrier per tile)
// ---------------------------------------------------------------------------
constexpr int T2_H = 4;

struct Wg2P {
  int n, h, w, cin, oh, ow, cout, kh, kw, stride, dil, pt, pl;
  int tiles_x, tiles_y, m_tiles, HT, WT, splits, tiles_per_split, nci, nco, xcd_swizzle;
};

// v2 uses v_mfma_f32_16x16x32_f16 with the K (pixel) order permuted identically for both operands:
// register j of lane group g holds pixel column 4g+j (j < 4) or 16+4g+(j-4) of the 32-pixel tile
// row, so the two 32-lane halves of a transposing read touch 8 consecutive pixels each.  With a
// pixel stride of 32*odd bytes those are 8 distinct 32-byte bank slots: conflict-free for every
// tap shift.
constexpr int X2STR = CIB * 2 + 32;   // 160 B

template <int COB2, int MAXTAPS>
__global__ __launch_bounds__(512) void wgrad2_kernel(Wg2P p, const half_t* __restrict__ x,
                                                          const half_t* __restrict__ dy,
                                                          float* __restrict__ slab) {
  // 8 waves: 2 (ci) x COB2/32 (co) x KG (pixel halves); every wave owns a 32x32 block per tap.
  // COB2 = 64 splits the tile's 8 k-steps over KG = 2 wave groups whose partial blocks leave as
  // two separate slab rows (split index 2*split + kgrp).
  constexpr int NT = 512;
  constexpr int NWCO = COB2 / 32;
  constexpr int KG = 8 / (2 * NWCO);
  constexpr int DSTR2 = COB2 * 2 + 32;     // bytes per dy pixel in LDS (32*odd)
  constexpr int DCH = COB2 / 8;            // 16-byte chunks per dy pixel
  constexpr int NDY = 128 * DCH / NT;      // dy loads per thread per tile
  constexpr int XPP = NT / 8;              // halo pixels covered per pass
  constexpr int NXMAX = (224 + XPP - 1) / XPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int halo_px = p.HT * p.WT;
  const int halo_bytes = halo_px * X2STR;
  const int stage_bytes = halo_bytes + 128 * DSTR2;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kgrp = wave / (2 * NWCO);
  const int ciw = (wave / NWCO) & 1, cow = wave % NWCO;
  const int li = lane & 15, g = lane >> 4;
  const int q = li >> 2, pp = li & 3;

  int bid = blockIdx.x;
  if (p.xcd_swizzle) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);   // see conv_igemm.hip
  const int cob = bid % p.nco;
  bid /= p.nco;
  const int cib = bid % p.nci;
  const int split = bid / p.nci;
  const int ci0 = cib * CIB, co0 = cob * COB2;
  const int ntaps = p.kh * p.kw;

  f32x4 acc[MAXTAPS][2][2];      // [tap][ci half][co half] 16x16 blocks of the wave's 32x32
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][i >> 1][i & 1][e] = 0.f;

  const int a_lane = ((4 * g + q) * p.stride) * X2STR + (ciw * 32 + 4 * pp) * 2;
  const int a_half = 16 * p.stride * X2STR;
  const int b_lane = (4 * g + q) * DSTR2 + (cow * 32 + 4 * pp) * 2;
  const int b_half = 16 * DSTR2;

  // this thread's halo chunks (fixed across tiles): packed (hy << 16 | hx), -1 = none
  int xpos[NXMAX];
  const int xc = tid & 7;
#pragma unroll
  for (int u = 0; u < NXMAX; ++u) {
    const int hp = u * XPP + (tid >> 3);
    xpos[u] = -1;
    if (hp < halo_px) {
      const int hy = hp / p.WT;
      xpos[u] = (hy << 16) | (hp - hy * p.WT);
    }
  }

  const int mt_begin = split * p.tiles_per_split;
  int mt_end = mt_begin + p.tiles_per_split;
  if (mt_end > p.m_tiles) mt_end = p.m_tiles;

  u32x4 xr[NXMAX], dr[NDY];
  auto load_tile = [&](int mt) {
    const int txi = mt % p.tiles_x;
    const int tmp = mt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int img = tmp / p.tiles_y;
    const int iy0 = tyi * T2_H * p.stride - p.pt;
    const int ix0 = txi * TILE_W * p.stride - p.pl;
    const half_t* xb = x + (size_t)img * p.h * p.w * p.cin + ci0 + xc * 8;
#pragma unroll
    for (int u = 0; u < NXMAX; ++u) {
      xr[u] = u32x4{0u, 0u, 0u, 0u};
      if (xpos[u] >= 0) {
        const int iy = iy0 + (xpos[u] >> 16), ix = ix0 + (xpos[u] & 0xffff);
        if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w && ci0 + xc * 8 < p.cin)
          xr[u] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * p.w + ix) * p.cin);
      }
    }
    const half_t* db = dy + (size_t)img * p.oh * p.ow * p.cout + co0;
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int idx = u * NT + tid;
      const int px = idx / DCH, c = idx % DCH;
      const int oy = tyi * T2_H + (px >> 5), ox = txi * TILE_W + (px & 31);
      dr[u] = u32x4{0u, 0u, 0u, 0u};
      if (oy < p.oh && ox < p.ow && co0 + c * 8 < p.cout)
        dr[u] = *reinterpret_cast<const u32x4*>(db + ((size_t)oy * p.ow + ox) * p.cout + c * 8);
    }
  };
  auto store_tile = [&](int buf) {
    char* xh = smem + buf * stage_bytes;
    char* dyt = xh + halo_bytes;
#pragma unroll
    for (int u = 0; u < NXMAX; ++u)
      if (xpos[u] >= 0)
        *reinterpret_cast<u32x4*>(xh + (u * XPP + (tid >> 3)) * X2STR + xc * 16) = xr[u];
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int idx = u * NT + tid;
      *reinterpret_cast<u32x4*>(dyt + (idx / DCH) * DSTR2 + (idx % DCH) * 16) = dr[u];
    }
  };

  OCR_DIAG_BEGIN()
  if (mt_begin < mt_end) {
    load_tile(mt_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int mt = mt_begin; mt < mt_end; ++mt) {
    const int buf = (mt - mt_begin) & 1;
    const bool more = mt + 1 < mt_end;
    if (more) load_tile(mt + 1);
    const char* xh = smem + buf * stage_bytes;
    const char* dyt = xh + halo_bytes;
#pragma unroll 1
    for (int ty = kgrp * (T2_H / KG); ty < (kgrp + 1) * (T2_H / KG); ++ty) {   // one 32-pixel row = one k32 step
      half8_t b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = tr_pair(dyt + b_lane + ty * 32 * DSTR2 + j * 32, b_half);
      const char* abase = xh + a_lane + ((ty * p.stride) * p.WT) * X2STR;
#pragma unroll
      for (int t = 0; t < MAXTAPS; ++t) {
        if (t < ntaps) {
          const int ky = t / p.kw, kx = t - ky * p.kw;
          const char* at = abase + ((ky * p.dil) * p.WT + kx * p.dil) * X2STR;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            half8_t a = tr_pair(at + i * 32, a_half);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[t][i][j] = OCR_MFMA_16x16x32(a, b[j], acc[t][i][j], 0, 0, 0);
          }
        }
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  OCR_DIAG_END(ocr_diag_wgrad)
  // D blocks: lane (li, g) holds rows (ci) 4g..4g+3 of column (co) li
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t) {
    if (t < ntaps) {
      float* dst = slab + (((size_t)(split * KG + kgrp) * ntaps + t) * p.cin + ci0 + ciw * 32 + 4 * g) * p.cout +
                   co0 + cow * 32 + li;
      if (ci0 + ciw * 32 < p.cin && co0 + cow * 32 < p.cout) {   // 32-wide partial blocks
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)(i * 16 + e) * p.cout + j * 16] = acc[t][i][j][e];
      }
    }
  }
}


// ---------------------------------------------------------------------------
// v3: the same tiling as v2 (64 ci x 128 co x taps per workgroup, 4 x 32-pixel tiles double-buffered in
// LDS, permuted-K transposing reads) on FOUR waves, one per SIMD, with the whole register file: wave tile
// 32 ci x 64 co x taps = 288 accumulator registers.  v2's eight waves own 32 x 32 x taps each and issue
// 40 transposing reads per 36 MFMAs (every shifted x fragment feeds 2 MFMAs): 142 B/clk of LDS reads
// per CU, MFMA pipe busy 43 % (profiles/r02_pmc_mfma.json).  Here a shifted x fragment feeds 4 MFMAs and
// a dy fragment 18: 44 reads per 72 MFMAs, i.e. 0.55x the LDS bytes per MFMA.  Accumulators are pinned
// in place (mfma16_acc); the staging registers of the next tile (60 per lane) fit beside them.
template <int N> struct WgIC { static constexpr int value = N; };
template <typename F>
__device__ __forceinline__ void static_for4(F&& f) {
  f(WgIC<0>{});
  f(WgIC<1>{});
  f(WgIC<2>{});
  f(WgIC<3>{});
}

template <int MAXTAPS, int COB2>
__global__ __launch_bounds__(256) void wgrad3_kernel(Wg2P p, const half_t* __restrict__ x,
                                                      const half_t* __restrict__ dy,
                                                      float* __restrict__ slab) {
  constexpr int NT = 256;
  static_assert(COB2 == 128 || COB2 == 64, "cout block");
  constexpr int NJ = COB2 / 32;             // 16-cout blocks per wave: wave tile 32 ci x (COB2 / 2) co
  constexpr int DSTR2 = COB2 * 2 + 32;
  constexpr int DCH = COB2 / 8;
  constexpr int NDY = 128 * DCH / NT;      // 8 | 4 passes over the 128 dy pixels
  constexpr int PPP = NT / DCH;            // 16 | 32 pixels per pass
  constexpr int XPP = NT / 8;              // 32 halo pixels per pass
  constexpr int NXMAX = (224 + XPP - 1) / XPP;   // 7
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int halo_px = p.HT * p.WT;
  const int halo_bytes = halo_px * X2STR;
  const int stage_bytes = halo_bytes + 128 * DSTR2;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ciw = wave >> 1, cow = wave & 1;       // 32-ci half, cout half
  const int li = lane & 15, g = lane >> 4;
  const int q = li >> 2, pp = li & 3;

  int bid = blockIdx.x;
  if (p.xcd_swizzle) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  const int cob = bid % p.nco;
  bid /= p.nco;
  const int cib = bid % p.nci;
  const int split = bid / p.nci;
  const int ci0 = cib * CIB, co0 = cob * COB2;
  const int ntaps = p.kh * p.kw;

  f32x4 acc[MAXTAPS][2][NJ];     // [tap][ci half of 16][16-cout block]
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_lane = ((4 * g + q) * p.stride) * X2STR + (ciw * 32 + 4 * pp) * 2;
  const int a_half = 16 * p.stride * X2STR;
  const int b_lane = (4 * g + q) * DSTR2 + (cow * (COB2 / 2) + 4 * pp) * 2;
  const int b_half = 16 * DSTR2;

  const int mt_begin = split * p.tiles_per_split;
  int mt_end = mt_begin + p.tiles_per_split;
  if (mt_end > p.m_tiles) mt_end = p.m_tiles;

  // Tile staging: 15 PIECES per tile (7 passes over the 6 x 34-pixel x halo, 8 over the 128 dy pixels), each
  // one 16-byte buffer load per lane and, a tile later, one 16-byte LDS store.  With one wave per SIMD every
  // instruction between two MFMAs costs issue cycles unless it is (nearly) alone in the gap, so the pieces
  // are built to be SHORT and are handed out one per MFMA gap by the main loop:
  //   * zero padding by the buffer range check with a PER-IMAGE descriptor: rows above / below the image
  //     fall outside [0, image bytes) on their own (negative offsets wrap), only the column needs a compare;
  //   * per-lane constants (column, relative byte offset) are set up once; the tile's origin is scalar and
  //     advances incrementally (no divisions in the loop);
  //   * a piece is two halves — offset (add, compare, select, add) | load — issued in consecutive gaps.
  // Both tensors are < 2 GiB (checked by the launcher).  Measured (ablation, conv4_2): staging issued as
  // one block per tile cost 2,800 of 8,700 cycles per tile.
  constexpr unsigned OOB = 0xfffffff0u;
  constexpr int WT_ = 34, HALO_PX = 6 * WT_;             // 3x3, stride 1: 6 x 34 halo pixels per 4 x 32 tile
  const int xc = tid & 7;
  int hxv[NXMAX];                                        // halo column of piece u (huge = never valid)
  unsigned relx[NXMAX];                                  // byte offset relative to the tile's halo origin
#pragma unroll
  for (int u = 0; u < NXMAX; ++u) {
    const int hp = u * XPP + (tid >> 3);
    const int hy = hp / WT_, hx = hp - hy * WT_;
    hxv[u] = (hp < HALO_PX && ci0 + xc * 8 < p.cin) ? hx : 0x40000000;
    relx[u] = (unsigned)(((hy * p.w + hx) * p.cin + ci0 + xc * 8) * 2);
  }
  const int dc = tid % DCH, dprow = tid / DCH;           // dy piece u: pixel u*PPP + dprow = row (u*PPP)>>5, column (u*PPP & 31) + dprow
  int colv[2];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) colv[h2] = (co0 + dc * 8 < p.cout) ? (h2 * 16) % 32 + dprow : 0x40000000;
  const unsigned reld = (unsigned)((dprow * p.cout + co0 + dc * 8) * 2);

  // the tile whose pieces are being loaded: scalar state
  int l_mt = mt_begin;
  int l_txi = l_mt % p.tiles_x, l_tyi = (l_mt / p.tiles_x) % p.tiles_y, l_img = l_mt / (p.tiles_x * p.tiles_y);
  int s_ix0 = 0, s_orgx = 0, s_ox0 = 0, s_orgd = 0;
  __amdgpu_buffer_rsrc_t xrs, drs;
  auto load_setup = [&]() {
    s_ix0 = l_txi * TILE_W - p.pl;
    s_orgx = ((l_tyi * T2_H - p.pt) * p.w + s_ix0) * p.cin * 2;
    s_ox0 = l_txi * TILE_W;
    s_orgd = ((l_tyi * T2_H) * p.ow + s_ox0) * p.cout * 2;
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(x) + (size_t)l_img * p.h * p.w * p.cin, 0,
                                            p.h * p.w * p.cin * 2, 0x00020000);
    drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(dy) + (size_t)l_img * p.oh * p.ow * p.cout, 0,
                                            p.oh * p.ow * p.cout * 2, 0x00020000);
  };
  auto load_advance = [&]() {                            // next tile of this split; stays on the last one
    if (l_mt + 1 < mt_end) {
      ++l_mt;
      if (++l_txi == p.tiles_x) {
        l_txi = 0;
        if (++l_tyi == p.tiles_y) {
          l_tyi = 0;
          ++l_img;
        }
      }
    }
    load_setup();
  };
  u32x4 xr[NXMAX], dr[NDY];
  unsigned offp = 0;
  auto load_half = [&](int hh) {                          // hh = 2 * piece + (0: offset | 1: load)
    const int k = hh >> 1;
    if (k >= NXMAX + NDY) return;
    if (k < NXMAX) {
      if (!(hh & 1)) {
        const int ix = hxv[k] + s_ix0;
        offp = (unsigned)ix < (unsigned)p.w ? relx[k] + (unsigned)s_orgx : OOB;
      } else {
        xr[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, offp, 0, 0));
      }
    } else {
      const int u = k - NXMAX;
      if (!(hh & 1)) {
        const int ox = colv[((u * PPP) >> 4) & 1] + s_ox0;
        offp = (unsigned)ox < (unsigned)p.ow ? reld + (unsigned)(s_orgd + (((u * PPP) >> 5) * p.ow + ((u * PPP) & 31)) * p.cout * 2) : OOB;
      } else {
        dr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, offp, 0, 0));
      }
    }
  };
  auto store_piece = [&](int buf, int k) {
    char* xh = smem + buf * stage_bytes;
    if (k < NXMAX) {
      if (k * XPP + XPP <= HALO_PX || k * XPP + (tid >> 3) < HALO_PX)
        *reinterpret_cast<u32x4*>(xh + (k * XPP + (tid >> 3)) * X2STR + xc * 16) = xr[k];
    } else if (k < NXMAX + NDY) {
      const int u = k - NXMAX;
      *reinterpret_cast<u32x4*>(xh + halo_bytes + (u * PPP + dprow) * DSTR2 + dc * 16) = dr[u];
    }
  };
  auto load_tile = [&]() {
#pragma unroll
    for (int hh = 0; hh < 2 * (NXMAX + NDY); ++hh) load_half(hh);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NXMAX + NDY; ++k) store_piece(buf, k);
  };

  // Software pipeline over k-steps (one 32-pixel tile row = 72 MFMAs per wave), fully unrolled per tile:
  //   step ty:   the 8 dy fragments and the first x fragment pair of step ty+1 are read behind this step's
  //              taps, the x fragment pair of tap t+1 behind tap t's MFMAs (LDS latency under MFMAs: with
  //              one wave per SIMD nothing else would cover it);
  //   ty == 1:   barrier (every wave is done with the other buffer); the next tile goes registers -> LDS, one
  //              piece per free MFMA gap, and the loads of the tile after it start behind them;
  //   ty == 2:   barrier (next tile visible), so step 3 can prefetch the next tile's first fragments; the
  //              rest of the loads (each has more than two steps to land).  Past the split's last tile the
  //              loads re-fetch that tile and the stores fill the unused buffer: no branches in the loop.
  // 3x3, stride 1, dilation 1 only (all LDS offsets are immediates); other shapes run v2.
  OCR_DIAG_BEGIN()
  if (mt_begin < mt_end) {
    load_setup();
    load_tile();
    store_tile(0);
    load_advance();
    load_tile();
  }
  __syncthreads();
#ifndef WG3_ABL
#define WG3_ABL 0      // dev ablations: 1 no tile staging in the loop, 2 no barriers, 4 no x fragment reads, 8 no dy fragment reads
#endif
  // Fragments are kept as their two transposing reads (k = 0..3 | 4..7 of the lane's eight), so that the loop
  // can issue ONE LDS read behind each MFMA: with one wave per SIMD an instruction between two MFMAs costs
  // its issue cycles unless it is alone in the gap (measured: 4 reads + a wait behind every 4th MFMA cost
  // 6.4 cycles per read, 19 % of the loop).  x fragment pairs: slot = running tap index & 3 (36 taps per
  // tile), fetched two taps ahead; dy fragments: two sets, the next step's fetched behind taps 4 and 5.
  typedef short short8v __attribute__((ext_vector_type(8)));
  short4v al[4][2], ah[4][2], bl[2][NJ], bh[2][NJ];
  auto rd = [&](const char* a) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a)); };
  auto frag = [&](const short4v& lo, const short4v& hi) {
    short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(half8_t, v);
  };
  // read k (0..2*NJ-1) of the dy fragments of step ty: fragment j = k >> 1, half k & 1
  auto read_b1 = [&](int set, const char* dyt, int ty, int k) {
    if constexpr ((WG3_ABL & 8) != 0) return;
    const char* a = dyt + b_lane + ty * 32 * DSTR2 + (k >> 1) * 32 + ((k & 1) ? b_half : 0);
    if (k & 1) bh[set][k >> 1] = rd(a);
    else bl[set][k >> 1] = rd(a);
  };
  // read k (0..3) of the x fragment pair of tap t of step ty: fragment i = k >> 1, half k & 1
  auto read_a1 = [&](int slot, const char* xh, int ty, int t, int k) {
    if constexpr ((WG3_ABL & 4) != 0) return;
    const int ky = t / 3, kx = t - ky * 3;
    const char* a = xh + a_lane + ((ty + ky) * WT_ + kx) * X2STR + (k >> 1) * 32 + ((k & 1) ? a_half : 0);
    if (k & 1) ah[slot][k >> 1] = rd(a);
    else al[slot][k >> 1] = rd(a);
  };
  if (mt_begin < mt_end) {
#pragma unroll
    for (int k = 0; k < 2 * NJ; ++k) read_b1(0, smem + halo_bytes, 0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      read_a1(0, smem, 0, 0, k);
      read_a1(1, smem, 0, 1, k);
    }
  }
  for (int mt = mt_begin; mt < mt_end; ++mt) {
    const int buf = (mt - mt_begin) & 1;
    const char* xh = smem + buf * stage_bytes;
    const char* dyt = xh + halo_bytes;
    const char* xh_n = smem + (buf ^ 1) * stage_bytes;
    static_for4([&](auto TY) {
      constexpr int ty = decltype(TY)::value;
      constexpr int S = ty & 1;                        // dy fragment set of this step
      if constexpr (ty == 1) {
        if constexpr (!(WG3_ABL & 2)) __builtin_amdgcn_s_barrier();                  // all reads of the other buffer are complete
      }
      if constexpr (ty == 1) load_advance();
      if constexpr (ty == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes of the next tile are done
        if constexpr (!(WG3_ABL & 2)) __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int t2 = t + 2;
        const int cur = (ty * 9 + t) & 3, nxt = (ty * 9 + t + 2) & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const half8_t fa = frag(al[cur][i], ah[cur][i]), fb = frag(bl[S][j], bh[S][j]);
            // 9 * 2 * NJ accumulator quads: taps 0..7 in the accumulator file, tap 8 in VGPRs
            if (t < 8) mfma16_acc(acc[t][i][j], fa, fb);
            else if (ty == 3 && i == 1 && j == NJ - 1) mfma16_acc_v_drain(acc[t][i][j], fa, fb);   // last MFMA of the loop body
            else mfma16_acc_v(acc[t][i][j], fa, fb);
            const int m = i * NJ + j;                  // gap behind this MFMA: 2 * NJ per tap
            // (a) the x fragment pair two taps ahead: four reads, one per gap (with NJ = 2 every gap of a tap)
            if (m < 4) {
              if (t2 < 9) read_a1(nxt, xh, ty, t2, m);
              else if (ty < 3) read_a1(nxt, xh, ty + 1, t2 - 9, m);
              else read_a1(nxt, xh_n, 0, t2 - 9, m);   // next tile (garbage past the last one: unused)
            }
            // (b) the next step's dy fragments (2 * NJ reads) and the staging pieces go into the gaps (a) leaves
            // free — NJ = 4: gaps 4..7 of every tap; NJ = 2: as a second instruction in all four gaps
            constexpr int FG = 4;                      // such slots per tap
            const int fm = NJ == 4 ? m - 4 : m;
            if (fm >= 0) {
              constexpr int NBT = 2 * NJ / FG;         // taps whose slots carry the dy reads: 4,5 | 4
              if (t >= 4 && t < 4 + NBT) {
                const int k = (t - 4) * FG + fm;
                if (ty < 3) read_b1(S ^ 1, dyt, ty + 1, k);
                else read_b1(S ^ 1, xh_n + halo_bytes, 0, k);
              } else if constexpr (!(WG3_ABL & 1)) {
                // free slot f of the step ((9 - NBT) * 4 of them): step 1 = the stores, then the first load
                // halves; step 2 = the remaining load halves
                constexpr int NPC = NXMAX + NDY, NSL = (9 - NBT) * FG;
                const int f = (t < 4 ? t : t - NBT) * FG + fm;
                if constexpr (ty == 1) {
                  if (f < NPC) store_piece(buf ^ 1, f);
                  else load_half(f - NPC);
                }
                if constexpr (ty == 2) load_half(f + NSL - NPC);
              }
            }
          }
      }
    });
  }
  OCR_DIAG_END(ocr_diag_wgrad)

  // D blocks: lane (li, g) holds rows (ci) 4g..4g+3 of column (co) li
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t) {
    if (t < ntaps) {
      float* dst = slab + (((size_t)split * ntaps + t) * p.cin + ci0 + ciw * 32 + 4 * g) * p.cout +
                   co0 + cow * (COB2 / 2) + li;
      if (ci0 + ciw * 32 < p.cin && co0 + cow * (COB2 / 2) < p.cout) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)(i * 16 + e) * p.cout + j * 16] = acc[t][i][j][e];
      }
    }
  }
}

// returns OCR_OK when v2 applies (and fills p / cob), OCR_ERR_UNSUPPORTED otherwise
int fill2(const ocr_conv_desc* d, Wg2P* p, int* cob) {
  if (d->cin % 32 || d->cout % 32 || d->kh * d->kw > 9) return OCR_ERR_UNSUPPORTED;
  *cob = (d->cout % 128 == 0) ? 128 : 64;
  p->n = d->n; p->h = d->h; p->w = d->w; p->cin = d->cin;
  p->oh = d->oh; p->ow = d->ow; p->cout = d->cout;
  p->kh = d->kh; p->kw = d->kw; p->stride = d->stride; p->dil = d->dilation;
  p->pt = d->pad_top; p->pl = d->pad_left;
  p->tiles_x = ocr_cdiv(d->ow, TILE_W);
  p->tiles_y = ocr_cdiv(d->oh, T2_H);
  p->m_tiles = d->n * p->tiles_x * p->tiles_y;
  p->HT = (T2_H - 1) * d->stride + (d->kh - 1) * d->dilation + 1;
  p->WT = (TILE_W - 1) * d->stride + (d->kw - 1) * d->dilation + 1;
  if (p->HT * p->WT > 7 * 32) return OCR_ERR_UNSUPPORTED;
  const size_t stage = (size_t)p->HT * p->WT * X2STR + 128 * (*cob * 2 + 32);
  if (2 * stage > 160 * 1024) return OCR_ERR_UNSUPPORTED;
  p->nci = ocr_cdiv(d->cin, CIB);
  p->nco = ocr_cdiv(d->cout, *cob);
  const int blocks = p->nci * p->nco;
  int want = ocr_cdiv(256, blocks);            // one resident workgroup per CU
  if (want > p->m_tiles) want = p->m_tiles;
  if (want < 1) want = 1;
  p->tiles_per_split = ocr_cdiv(p->m_tiles, want);
  p->splits = ocr_cdiv(p->m_tiles, p->tiles_per_split);
  return OCR_OK;
}

int fill(const ocr_conv_desc* d, WgP* p) {
  OCR_CHECK_ARG(d != nullptr);
  OCR_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->oh > 0 && d->ow > 0);
  OCR_CHECK_SHAPE(d->cin % 64 == 0 && d->cout % 64 == 0);
  OCR_CHECK_SHAPE(d->kh * d->kw <= 9);
  p->n = d->n; p->h = d->h; p->w = d->w; p->cin = d->cin;
  p->oh = d->oh; p->ow = d->ow; p->cout = d->cout;
  p->kh = d->kh; p->kw = d->kw; p->stride = d->stride; p->dil = d->dilation;
  p->pt = d->pad_top; p->pl = d->pad_left;
  p->tiles_x = ocr_cdiv(d->ow, TILE_W);
  p->tiles_y = ocr_cdiv(d->oh, TILE_H);
  p->m_tiles = d->n * p->tiles_x * p->tiles_y;
  p->HT = (TILE_H - 1) * d->stride + (d->kh - 1) * d->dilation + 1;
  p->WT = (TILE_W - 1) * d->stride + (d->kw - 1) * d->dilation + 1;
  p->nci = d->cin / CIB;
  p->nco = d->cout / COB;
  const int blocks = p->nci * p->nco;
  int want = ocr_cdiv(1024, blocks);           // ~4 workgroups per CU in total
  if (want > p->m_tiles) want = p->m_tiles;
  if (want < 1) want = 1;
  p->tiles_per_split = ocr_cdiv(p->m_tiles, want);
  p->splits = ocr_cdiv(p->m_tiles, p->tiles_per_split);
  size_t lds = (size_t)p->HT * p->WT * XSTR + 256 * DSTR;
  if (lds > 160 * 1024) return OCR_ERR_UNSUPPORTED;
  return OCR_OK;
}

}  // namespace

OCR_DIAG_READER(ocr_diag_read_wgrad, ocr_diag_wgrad)

template <typename K, typename P>
static int launch_wg(K kern, const P& p, unsigned grid, size_t lds, const void* x, const void* dy,
                     void* ws, hipStream_t st, unsigned threads = 256) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return OCR_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, p, static_cast<const half_t*>(x),
                     static_cast<const half_t*>(dy), static_cast<float*>(ws));
  return ocr_launch_status();
}

// Which kernel family computes the weight gradient of `d`, with its plan — the ONE place this is decided: the launch
// (wgrad_slabs), the slab count the reduction sums (wgrad_slab_count) and the register room a guest finds beside the
// kernel (ocr_conv2d_wgrad_guest_room) all read it.
struct WgradSel {
  enum Family { NONE, PW, TAP3, TAP2, GENERIC } family = NONE;
  int splits = 0;       // slabs written
  int cob = 0;          // TAP3 / TAP2: cout block (64 | 128)
  Wg2P p2{};            // TAP3 / TAP2
  WgP p{};              // GENERIC
};
static WgradSel wgrad_select(const ocr_conv_desc* d) {
  WgradSel s;
  if ((s.splits = ocr_detail::wgrad_pw_splits(d)) > 0) {
    s.family = WgradSel::PW;
    return s;
  }
  if (fill2(d, &s.p2, &s.cob) == OCR_OK) {
    static const int v3 = [] { const char* e = getenv("OCR_WGRAD3"); return e ? atoi(e) : 1; }();   // 0: the wgrad2 family (tests)
    const bool small = (size_t)d->n * d->h * d->w * d->cin < (1u << 30) && (size_t)d->n * d->oh * d->ow * d->cout < (1u << 30);   // < 2 GiB each
    const bool use3 = d->kh * d->kw == 9 && v3 && small && d->cin % 64 == 0 && d->kw == 3 && d->stride == 1 && d->dilation == 1;
    s.family = use3 ? WgradSel::TAP3 : WgradSel::TAP2;
    s.splits = s.p2.splits * (s.cob == 64 && !use3 ? 2 : 1);       // (wgrad2<64> splits K once more inside the workgroup)
    return s;
  }
  if (fill(d, &s.p) == OCR_OK) {
    s.family = WgradSel::GENERIC;
    s.splits = s.p.splits * 2;
  }
  return s;
}

extern "C" size_t ocr_conv2d_wgrad_workspace(const ocr_conv_desc* d) {
  if (!d) return 0;
  return (size_t)wgrad_select(d).splits * d->kh * d->kw * d->cin * d->cout * sizeof(float);
}

// The partial slabs [splits][kh][kw][cin][cout] of one weight gradient (every kernel family); *splits_out = their number.
static int wgrad_slabs(const ocr_conv_desc* d, const void* x, const void* dy, void* workspace, size_t ws_bytes,
                       void* stream, int* splits_out) {
  OCR_CHECK_ARG(d != nullptr);
  OCR_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->oh > 0 && d->ow > 0);
  OCR_CHECK_ARG(x && dy && workspace);
  const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ntaps = d->kh * d->kw;
  WgradSel sel = wgrad_select(d);
  if (sel.family == WgradSel::NONE) return OCR_ERR_UNSUPPORTED;
  const int splits = sel.splits;
  if (ws_bytes < (size_t)splits * elems * sizeof(float)) return OCR_ERR_WORKSPACE;
  int rc;
  if (sel.family == WgradSel::PW) {
    rc = ocr_detail::wgrad_pw_launch(d, x, dy, workspace, st);
  } else if (sel.family == WgradSel::TAP3 || sel.family == WgradSel::TAP2) {
    Wg2P& p2 = sel.p2;
    const int cob = sel.cob;
    const size_t lds = 2 * ((size_t)p2.HT * p2.WT * X2STR + 128 * (cob * 2 + 32));
    const unsigned grid = (unsigned)(p2.splits * p2.nci * p2.nco);
    p2.xcd_swizzle = grid % 8 == 0;           // XCD-aware workgroup order (conv_igemm.hip)
    if (sel.family == WgradSel::TAP3)
      rc = cob == 128 ? launch_wg(wgrad3_kernel<9, 128>, p2, grid, lds, x, dy, workspace, st, 256)
                      : launch_wg(wgrad3_kernel<9, 64>, p2, grid, lds, x, dy, workspace, st, 256);
    else if (cob == 128)
      rc = ntaps == 1 ? launch_wg(wgrad2_kernel<128, 1>, p2, grid, lds, x, dy, workspace, st, 512)
                      : launch_wg(wgrad2_kernel<128, 9>, p2, grid, lds, x, dy, workspace, st, 512);
    else
      rc = ntaps == 1 ? launch_wg(wgrad2_kernel<64, 1>, p2, grid, lds, x, dy, workspace, st, 512)
                      : launch_wg(wgrad2_kernel<64, 9>, p2, grid, lds, x, dy, workspace, st, 512);
  } else {
    const WgP& p = sel.p;
    const size_t lds = (size_t)p.HT * p.WT * XSTR + 256 * DSTR;
    const unsigned grid = (unsigned)(p.splits * p.nci * p.nco);
    rc = ntaps == 1 ? launch_wg(wgrad_kernel<1>, p, grid, lds, x, dy, workspace, st, 512)
                    : launch_wg(wgrad_kernel<9>, p, grid, lds, x, dy, workspace, st, 512);
  }
  if (rc != OCR_OK) return rc;
  *splits_out = splits;
  return ocr_launch_status();
}

extern "C" int ocr_conv2d_wgrad_f16(const ocr_conv_desc* d, const void* x, const void* dy,
                                    void* dw, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(dw != nullptr);
  int splits = 0;
  const int rc = wgrad_slabs(d, x, dy, workspace, ws_bytes, stream, &splits);
  if (rc != OCR_OK) return rc;
  const size_t elems4 = (size_t)d->kh * d->kw * d->cin * d->cout / 4;
  launch_slab_reduce(static_cast<const float*>(workspace), static_cast<float*>(dw), elems4, splits,
                     static_cast<hipStream_t>(stream));
  return ocr_launch_status();
}

// The two halves of ocr_conv2d_wgrad_f16 as calls of their own: the recorded step issues the slab kernel as the HOST of
// a guest pass and the (bandwidth- and cache-sensitive: 13 us alone, 60-370 us beside an HBM-streaming guest) slab sum
// behind the join.  `workspace` must then be the launch's own until the reduce has run.
extern "C" int ocr_conv2d_wgrad_slabs_f16(const ocr_conv_desc* d, const void* x, const void* dy, void* workspace,
                                          size_t ws_bytes, void* stream) {
  int splits = 0;
  return wgrad_slabs(d, x, dy, workspace, ws_bytes, stream, &splits);
}

// the number of slabs wgrad_slabs writes for `d` (the same selection, nothing launched); <= 0: unsupported
static int wgrad_slab_count(const ocr_conv_desc* d) { return wgrad_select(d).splits; }

// Registers per lane a SIMD has LEFT beside the resident workgroup(s) of the kernel ocr_conv2d_wgrad_slabs_f16 selects for
// `d` — what a guest wave may use (csrc/guest_bn.hip needs 56) — or 0 where the kernel fills the file / is not tabulated.
// From -Rpass-analysis=kernel-resource-usage (allocation granule 8, AGPRs behind the VGPRs rounded to 4):
//   wgrad3_kernel<9,128>: 198 + 256 -> 456, one wave per SIMD                       -> 56
//   wgrad3_kernel<9,64>:  134 + 128 -> 264, one wave per SIMD (106 KB LDS: one WG)   -> 248
//   wgrad_pw_kernel<256,256,2>: 212 -> 216, two waves per SIMD (one WG of 512)       -> 80
// tests/test_host_cpu.py::test_guest_kernels_fit_beside_the_weight_gradient re-derives the first from the compiler.
extern "C" int ocr_conv2d_wgrad_guest_room(const ocr_conv_desc* d) {
  if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh <= 0 || d->ow <= 0) return 0;
  const WgradSel sel = wgrad_select(d);
  if (sel.family == WgradSel::PW) return ocr_detail::wgrad_pw_is_256x256(d) ? 80 : 0;
  if (sel.family != WgradSel::TAP3) return 0;
  return sel.cob == 128 ? 56 : 248;
}

extern "C" int ocr_conv2d_wgrad_reduce_f32(const ocr_conv_desc* d, const void* workspace, void* dw, void* stream) {
  OCR_CHECK_ARG(d && workspace && dw);
  OCR_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->oh > 0 && d->ow > 0);
  const int splits = wgrad_slab_count(d);
  if (splits <= 0) return OCR_ERR_UNSUPPORTED;
  const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
  launch_slab_reduce(static_cast<const float*>(workspace), static_cast<float*>(dw), elems / 4, splits,
                     static_cast<hipStream_t>(stream));
  return ocr_launch_status();
}
