// Weight gradient of 1x1 (pointwise) convolutions -- and, tap by tap, of dilated k x k convolutions
// (fc6, nets/vgg.py:36) -- on gfx950 MFMA: a K-major x K-major GEMM
//
//   dw[ci,co] = sum_{n,oy,ox} x[n, oy*s, ox*s, ci] * dy[n,oy,ox,co]          (M = ci, N = co, K = pixels)
//
// (reference call sites: the 1x1 bottleneck / shortcut convs of nets/resnet_v1.py:97-105 through
// resnet_utils, fc7 of nets/vgg.py:37, the feature-merge convs of nets/model_vgg_16.py:111-131;
// gradient taken by `opt.compute_gradients`, multigpu_train.py:129.)
//
// With a single tap there is no shifted-window reuse to exploit, so the workgroup tile is a plain
// GEMM block: CIB x COB outputs (up to 256 x 256), 8 waves as NWCI x (8/NWCI), each wave holding
// (CIB/NWCI) x (COB*NWCI/8) outputs as 16x16 accumulators of v_mfma_f32_16x16x32_f16.  Both
// operands are [pixel][channel] in HBM, i.e. K-major; the MFMA fragments (8 K values per lane) come
// from ds_read_b64_tr_b16 on NHWC tiles staged in LDS, with the K order permuted identically for
// both operands (register j of lane group g holds pixel 4g+j, j<4, and 16+4g+(j-4) otherwise) so
// that the two 32-lane halves of a transposing read touch pixel rows 0..7 / 8..15: with a row
// stride of 32*odd bytes those are 8 distinct 32-byte bank slots (conflict-free).
// Pixel stages of 64 (2 x 32 output pixels) are double-buffered: the global loads of stage t+1 are
// issued before the MFMAs of stage t.  Split-K partial blocks go to a [split][ci][co] f32 slab that
// the caller reduces in fixed order (bitwise reproducible, no atomics).
#include "common.h"

namespace {

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

__device__ __forceinline__ half8_t tr_pair16(const char* base, int second_off) {
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base));
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + second_off));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8_t, v);
}

struct PwP {
  int n, h, w, cin, oh, ow, cout, stride, pt, pl, kh, kw, dil;
  int tiles_x, tiles_y, m_tiles, splits, tiles_per_split, nci, nco, xcd_swizzle;
};

constexpr int PXS = 64;      // pixels per stage: 2 rows x 32 columns of the output map
constexpr int PW_ROWS = 2;

template <int CIB, int COB, int NWCI>
__global__ __launch_bounds__(512) void wgrad_pw_kernel(PwP p, const half_t* __restrict__ x,
                                                       const half_t* __restrict__ dy,
                                                       float* __restrict__ slab) {
  constexpr int NT = 512;
  constexpr int NWCO = 8 / NWCI;
  constexpr int WCI = CIB / NWCI, WCO = COB / NWCO;     // wave tile
  constexpr int AI = WCI / 16, AJ = WCO / 16;           // 16x16 accumulator blocks
  constexpr int XS = CIB * 2 + 32, DS = COB * 2 + 32;   // LDS row strides (32*odd bytes)
  constexpr int XCH = CIB / 8, DCH = COB / 8;           // 16-byte chunks per pixel
  constexpr int NX = PXS * XCH / NT, ND = PXS * DCH / NT;
  constexpr int STAGE = PXS * (XS + DS);
  static_assert(NX >= 1 && ND >= 1 && AI >= 1 && AJ >= 1, "tile too small for 512 threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wci = wave % NWCI, wco = wave / NWCI;
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;

  // kh*kw > 1 (dilated convs whose halo is too large for the tap-sweeping kernel): every tap is its
  // own pointwise GEMM on a window of x shifted by the tap offset -- one workgroup per (tap, block)
  int bid = blockIdx.x;
  // XCD-aware order (conv_igemm.hip): the nci x nco blocks of one pixel range (split) run on ONE XCD, whose L2 serves
  // the x rows to the nco blocks that share them and the dy rows to the nci blocks — dealt round-robin, each of the
  // eight L2s fetched both (PMC: 1.5-2x the operand bytes on the 256 x 256-tile launches of ResNet's stages 3-4)
  if (p.xcd_swizzle) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  const int ntaps = p.kh * p.kw;
  const int tap = bid % ntaps;
  bid /= ntaps;
  const int cob = bid % p.nco;
  bid /= p.nco;
  const int cib = bid % p.nci;
  const int split = bid / p.nci;
  const int ci0 = cib * CIB, co0 = cob * COB;
  const int tky = tap / p.kw, tkx = tap - tky * p.kw;
  const int pt = p.pt - tky * p.dil, pl = p.pl - tkx * p.dil;

  f32x4 acc[AI][AJ];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < AJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int a_lane = (4 * g + q) * XS + (wci * WCI + 4 * pp) * 2;
  const int b_lane = (4 * g + q) * DS + (wco * WCO + 4 * pp) * 2;

  const int mt_begin = split * p.tiles_per_split;
  int mt_end = mt_begin + p.tiles_per_split;
  if (mt_end > p.m_tiles) mt_end = p.m_tiles;

  // Stage loads are BUFFER loads with the range check doing the zero padding (an offset beyond the descriptor reads
  // as zero): with ordinary loads under `if (inside)` hipcc branches around every one of the eight loads and waits
  // for each before the next (conv_wgrad.hip: wgrad3_kernel).  Both tensors are < 2 GiB (checked by pw_plan).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(x), 0, (int)((size_t)p.n * p.h * p.w * p.cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(dy), 0, (int)((size_t)p.n * p.oh * p.ow * p.cout * 2), 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  u32x4 xr[NX], dr[ND];
  auto load_tile = [&](int mt) {
    const int txi = mt % p.tiles_x;
    const int tmp = mt / p.tiles_x;
    const int tyi = tmp % p.tiles_y;
    const int img = tmp / p.tiles_y;
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int idx = u * NT + tid;
      const int px = idx / XCH, c = idx % XCH;
      const int oy = tyi * PW_ROWS + (px >> 5), ox = txi * 32 + (px & 31);
      const int iy = oy * p.stride - pt, ix = ox * p.stride - pl;
      const bool ok = (oy < p.oh) & (ox < p.ow) & ((unsigned)iy < (unsigned)p.h) & ((unsigned)ix < (unsigned)p.w);
      const unsigned off = ok ? (unsigned)((((img * p.h + iy) * p.w + ix) * p.cin + ci0 + c * 8) * 2) : OOB;
      xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int idx = u * NT + tid;
      const int px = idx / DCH, c = idx % DCH;
      const int oy = tyi * PW_ROWS + (px >> 5), ox = txi * 32 + (px & 31);
      const bool ok = (oy < p.oh) & (ox < p.ow);
      const unsigned off = ok ? (unsigned)((((img * p.oh + oy) * p.ow + ox) * p.cout + co0 + c * 8) * 2) : OOB;
      dr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
    char* xs = smem + buf * STAGE;
    char* ds = xs + PXS * XS;
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int idx = u * NT + tid;
      *reinterpret_cast<u32x4*>(xs + (idx / XCH) * XS + (idx % XCH) * 16) = xr[u];
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int idx = u * NT + tid;
      *reinterpret_cast<u32x4*>(ds + (idx / DCH) * DS + (idx % DCH) * 16) = dr[u];
    }
  };

  if (mt_begin < mt_end) {
    load_tile(mt_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int mt = mt_begin; mt < mt_end; ++mt) {
    const int buf = (mt - mt_begin) & 1;
    const bool more = mt + 1 < mt_end;
    if (more) load_tile(mt + 1);
    const char* xs = smem + buf * STAGE;
    const char* ds = xs + PXS * XS;
#pragma unroll
    for (int s = 0; s < PXS / 32; ++s) {
      half8_t a[AI], b[AJ];
#pragma unroll
      for (int i = 0; i < AI; ++i) a[i] = tr_pair16(xs + a_lane + s * 32 * XS + i * 32, 16 * XS);
#pragma unroll
      for (int j = 0; j < AJ; ++j) b[j] = tr_pair16(ds + b_lane + s * 32 * DS + j * 32, 16 * DS);
#pragma unroll
      for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < AJ; ++j)
          acc[i][j] = OCR_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // D block: lane (li, g) holds rows (ci) 4g..4g+3 of column (co) li
  float* dst = slab + (((size_t)split * ntaps + tap) * p.cin + ci0 + wci * WCI + 4 * g) * p.cout + co0 +
               wco * WCO + li;
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < AJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(size_t)(i * 16 + e) * p.cout + j * 16] = acc[i][j][e];
}

struct PwCfg { int cib, cob; };

bool pw_plan(const ocr_conv_desc* d, PwP* p, PwCfg* c) {
  if (d->cin % 64 || d->cout % 64) return false;
  if ((size_t)d->n * d->h * d->w * d->cin >= (1u << 30) || (size_t)d->n * d->oh * d->ow * d->cout >= (1u << 30))
    return false;                                   // 32-bit buffer offsets (the tap-sweeping kernels take these)
  // 1x1 convs, and dilated k x k convs (taps far apart: no halo reuse worth a tap-sweeping tile)
  if (!(d->kh * d->kw == 1 || (d->dilation > 1 && d->kh * d->kw <= 9))) return false;
  c->cib = d->cin % 256 == 0 ? 256 : d->cin % 128 == 0 ? 128 : 64;
  c->cob = d->cout % 256 == 0 ? 256 : d->cout % 128 == 0 ? 128 : 64;
  constexpr int wgs = 256;                       // one resident workgroup per CU
  p->n = d->n; p->h = d->h; p->w = d->w; p->cin = d->cin;
  p->oh = d->oh; p->ow = d->ow; p->cout = d->cout;
  p->stride = d->stride; p->pt = d->pad_top; p->pl = d->pad_left;
  p->kh = d->kh; p->kw = d->kw; p->dil = d->dilation;
  p->tiles_x = ocr_cdiv(d->ow, 32);
  p->tiles_y = ocr_cdiv(d->oh, PW_ROWS);
  p->m_tiles = d->n * p->tiles_x * p->tiles_y;
  p->nci = d->cin / c->cib;
  p->nco = d->cout / c->cob;
  const int blocks = p->nci * p->nco * d->kh * d->kw;
  // one resident workgroup per CU, and never a few workgroups over a full round of 256 (fc6: 72 blocks x
  // 4 splits = 288 took two rounds; x 3 = 216 takes one)
  int want = blocks <= wgs ? wgs / blocks : 1;
  if (want > p->m_tiles) want = p->m_tiles;
  if (want < 1) want = 1;
  p->tiles_per_split = ocr_cdiv(p->m_tiles, want);
  p->splits = ocr_cdiv(p->m_tiles, p->tiles_per_split);
  return true;
}

template <int CIB, int COB, int NWCI>
int pw_launch(const PwP& p, const void* x, const void* dy, void* slab, hipStream_t st) {
  auto kern = wgrad_pw_kernel<CIB, COB, NWCI>;
  const size_t lds = 2 * (size_t)PXS * ((CIB * 2 + 32) + (COB * 2 + 32));
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return OCR_ERR_HIP;
    configured = true;
  }
  PwP q = p;
  const unsigned grid = (unsigned)(p.splits * p.nci * p.nco * p.kh * p.kw);
  q.xcd_swizzle = p.nci * p.nco * p.kh * p.kw > 1 && grid % 8 == 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, q,
                     static_cast<const half_t*>(x), static_cast<const half_t*>(dy), static_cast<float*>(slab));
  return ocr_launch_status();
}

}  // namespace

namespace ocr_detail {

// Number of split-K slabs ([taps][cin][cout] f32 each) the pointwise path writes; 0 = shape not handled.
int wgrad_pw_splits(const ocr_conv_desc* d) {
  PwP p;
  PwCfg c;
  return pw_plan(d, &p, &c) ? p.splits : 0;
}

bool wgrad_pw_is_256x256(const ocr_conv_desc* d) {
  PwP p;
  PwCfg c;
  return pw_plan(d, &p, &c) && c.cib == 256 && c.cob == 256;
}

int wgrad_pw_launch(const ocr_conv_desc* d, const void* x, const void* dy, void* slab, hipStream_t st) {
  PwP p;
  PwCfg c;
  if (!pw_plan(d, &p, &c)) return OCR_ERR_UNSUPPORTED;
  switch (c.cib * 1000 + c.cob) {
    case 256256: return pw_launch<256, 256, 2>(p, x, dy, slab, st);
    case 256128: return pw_launch<256, 128, 4>(p, x, dy, slab, st);
    case 256064: return pw_launch<256, 64, 4>(p, x, dy, slab, st);
    case 128256: return pw_launch<128, 256, 2>(p, x, dy, slab, st);
    case 128128: return pw_launch<128, 128, 2>(p, x, dy, slab, st);
    case 128064: return pw_launch<128, 64, 4>(p, x, dy, slab, st);
    case 64256: return pw_launch<64, 256, 1>(p, x, dy, slab, st);
    case 64128: return pw_launch<64, 128, 1>(p, x, dy, slab, st);
    case 64064: return pw_launch<64, 64, 2>(p, x, dy, slab, st);
  }
  return OCR_ERR_UNSUPPORTED;
}

}  // namespace ocr_detail
