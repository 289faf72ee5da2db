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

namespace {

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

struct WgP {
  int n, h, w, cin, oh, ow, cout, kh, kw, stride, dil, pt, pl;
  int tiles_x, tiles_y, m_tiles, HT, WT, splits, tiles_per_split, nci, nco;
};

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
__global__ __launch_bounds__(256) void wgrad_kernel(WgP p, const half_t* __restrict__ x,
                                                    const half_t* __restrict__ dy,
                                                    float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int halo_px = p.HT * p.WT;
  char* xh = smem;
  char* dyt = smem + halo_px * XSTR;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ciw = wave >> 1, cow = wave & 1;
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
      for (int base = 0; base < total; base += 256 * 4) {
        u32x4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * 256 + tid;
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
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int idx = u * 256 + tid;
        int px = idx >> 3, c = idx & 7;
        int oy = tyi * TILE_H + (px >> 5), ox = txi * TILE_W + (px & 31);
        v[u] = u32x4{0u, 0u, 0u, 0u};
        if (oy < p.oh && ox < p.ow)
          v[u] = *reinterpret_cast<const u32x4*>(db + ((size_t)oy * p.ow + ox) * p.cout + c * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int idx = u * 256 + tid;
        int px = idx >> 3, c = idx & 7;
        *reinterpret_cast<u32x4*>(dyt + px * DSTR + c * 16) = v[u];
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {
      const int ty = s >> 1, tx0 = (s & 1) * 16;
      half8_t b = tr_pair(dyt + b_lane + (ty * 32 + tx0) * DSTR, b_half);
      const char* abase = xh + a_lane + ((ty * p.stride) * p.WT + tx0 * p.stride) * XSTR;
#pragma unroll
      for (int t = 0; t < MAXTAPS; ++t) {
        if (t < ntaps) {
          const int ky = t / p.kw, kx = t - ky * p.kw;
          half8_t a = tr_pair(abase + ((ky * p.dil) * p.WT + kx * p.dil) * XSTR, a_half);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // partial block -> slab[split][tap][ci][co]
  const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t) {
    if (t < ntaps) {
      float* dst = slab + (((size_t)split * ntaps + t) * p.cin + ci0 + ciw * 32) * p.cout + co0 +
                   cow * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h2;
        dst[(size_t)row * p.cout] = acc[t][e];
      }
    }
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                   size_t elems4, int splits) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems4) return;
  const f32x4* s = reinterpret_cast<const f32x4*>(slab);
  f32x4 a = s[i];
  for (int k = 1; k < splits; ++k) {
    f32x4 b = s[(size_t)k * elems4 + i];
    a += b;
  }
  reinterpret_cast<f32x4*>(dw)[i] = a;
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

extern "C" size_t ocr_conv2d_wgrad_workspace(const ocr_conv_desc* d) {
  WgP p;
  if (fill(d, &p) != OCR_OK) return 0;
  return (size_t)p.splits * d->kh * d->kw * d->cin * d->cout * sizeof(float);
}

extern "C" int ocr_conv2d_wgrad_f16(const ocr_conv_desc* d, const void* x, const void* dy,
                                    void* dw, void* workspace, size_t ws_bytes, void* stream) {
  WgP p;
  int rc = fill(d, &p);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x && dy && dw && workspace);
  const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
  if (ws_bytes < (size_t)p.splits * elems * sizeof(float)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)p.HT * p.WT * XSTR + 256 * DSTR;
  const int ntaps = d->kh * d->kw;
  dim3 grid((unsigned)(p.splits * p.nci * p.nco));
  auto go = [&](auto kern) -> int {
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        return OCR_ERR_HIP;
      configured = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p, static_cast<const half_t*>(x),
                       static_cast<const half_t*>(dy), static_cast<float*>(workspace));
    return ocr_launch_status();
  };
  rc = (ntaps == 1) ? go(wgrad_kernel<1>) : go(wgrad_kernel<9>);
  if (rc != OCR_OK) return rc;
  const size_t elems4 = elems / 4;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((elems4 + 255) / 256)), dim3(256), 0, st,
                     static_cast<const float*>(workspace), static_cast<float*>(dw), elems4,
                     p.splits);
  return ocr_launch_status();
}
