// PixelLink fuse heads: 1x1 convolutions from wide f16 feature maps to a few
// channels (2 pixel + 16 link logits), their gradients, and the "small-channel"
// f32 tensors [P][C] (C <= 32) that the unpool+add pyramid, the head batch
// norms and the final 1x1 predication convs work on.
//
// Reference: nets/model_vgg_16.py:160-175 (BN'd heads), nets/pixellink.py:55-67
// (bias heads), nets/model.py:129-141 (ResNet heads), unpool = legacy
// tf.image.resize_bilinear x2 (nets/model.py:14-15).
//
// The wide->narrow convs are HBM-bound (each feature byte is read once and
// 18 outputs are produced), so they stream the features straight from global
// memory into MFMA B-fragments (16 B per lane) with the tiny weight matrix as A.
#include "common.h"

namespace {

// out[p][co] = sum_ci x[p][ci] * w[co][ci]     (w rows >= cout are zero)
// One wave = 32 pixels.  The MFMA B fragment wants 8 consecutive channels of ONE pixel per lane; read that
// way from global memory every lane walks its own pixel row in 32-byte pieces (3.0 TB/s measured).  Instead a
// wave fetches 32 px x 64 ch tiles with whole 128-byte lines (8 lanes per pixel row), parks them in its own
// 4.6 KB of LDS (row stride 144 B: the sixteen 16-byte fragment reads of a phase hit sixteen bank groups) and
// takes the fragments from there.
// One wave, one 32-pixel group starting at p0; sx = the wave's 32 * RS bytes of LDS.  On return the wave's 32 x cout
// outputs sit in sx as f32 [32][cout] (rows >= P - p0: zero + bias) and have been written to `out`.
constexpr int kSmallRS = 144;
__device__ __forceinline__ void conv1x1_small_group(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                    const float* __restrict__ bias, int P, int cin, int cout,
                                                    float* __restrict__ out, int p0, char* sx) {
  constexpr int RS = kSmallRS;
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  const half_t* wp = w + (size_t)r * cin + 8 * hh;
  const int lrow = lane >> 3, lc = lane & 7;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  u32x4 v[4];
  bool vok[4];
  // branch-free requests (a row past the end / a chunk past cin reads element 0; it is replaced by zero where the row is
  // written to LDS, one K step later — a select here would wait for the load): under `if (inside)` hipcc waited for each
  // of the four loads before requesting the next
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = p0 + lrow + 8 * i;
      vok[i] = pp < P && k0 + lc * 8 < cin;
      v[i] = *reinterpret_cast<const u32x4*>(x + (vok[i] ? (size_t)pp * cin + k0 + lc * 8 : 0));
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < cin; k0 += 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                        // the previous tile's fragments have been read
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4*>(sx + (lrow + 8 * i) * RS + lc * 16) = vok[i] ? v[i] : u32x4{0u, 0u, 0u, 0u};
    // the four weight fragments of this K step, requested together and BEFORE the next tile's rows (loads retire in
    // order: a request the MFMAs below wait for must not stand behind the prefetch)
    half8_t a4[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a4[kk] = *reinterpret_cast<const half8_t*>(wp + (k0 + kk * 16 < cin ? k0 + kk * 16 : 0));
    fetch(k0 + 64);                                         // the next tile is in flight under this one's MFMAs (past the
                                                            // last one: four dummy requests — unconditional, so that the
                                                            // counted waits below know what is outstanding)
    __builtin_amdgcn_sched_barrier(0);                      // (pinned here: hipcc sank the requests to the next iteration's
                                                            // ds_write, i.e. every K step waited out its own fetch)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (k0 + kk * 16 < cin) {
        half8_t b = *reinterpret_cast<const half8_t*>(sx + r * RS + kk * 32 + hh * 16);
        acc = OCR_MFMA_32x32x16(a4[kk], b, acc, 0, 0, 0);
      }
    }
  }
  // the wave's 32 x cout outputs are one contiguous span of `out`: through LDS, then whole dwords in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float* so = reinterpret_cast<float*>(sx);                 // 32 * cout floats <= 4096 B
  float bv[16];                                             // (all sixteen requested before the first is used)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
    const float* bp = bias ? bias : reinterpret_cast<const float*>(w);
    bv[e] = bp[co < cout ? co : 0];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (co < cout) so[r * cout + co] = acc[e] + (bias ? bv[e] : 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int npx = min(32, P - p0);
  float* dst = out + (size_t)p0 * cout;
  for (int i = lane; i < npx * cout; i += 64) dst[i] = so[i];
}

__global__ __launch_bounds__(256) void conv1x1_small_kernel(const half_t* __restrict__ x,
                                                            const half_t* __restrict__ w,
                                                            const float* __restrict__ bias, int P,
                                                            int cin, int cout,
                                                            float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char s_x[4][32 * kSmallRS];
  const int wave = threadIdx.x >> 6;
  conv1x1_small_group(x, w, bias, P, cin, cout, out, (blockIdx.x * 4 + wave) * 32, s_x[wave]);
}

// The same convolution for up to FOUR feature maps in one launch (the four fuse-head sources of
// nets/model_vgg_16.py:160-172 / nets/pixellink.py:58-67: launched one by one they are 4 x (conv + statistics) small
// launches that each leave most of the chip idle), with the batch-norm statistics of the output in the epilogue: a
// workgroup walks `iters` consecutive 128-pixel groups, every wave keeps per-channel sums of z and z^2 of its rows in
// registers (lane = half * 32 + channel: 16 of the 32 staged rows each), and the workgroup leaves ONE partial row
// [2][cout] — at most 1024 rows per map, which one block per map finalises directly (bn_finalize_batch_kernel).
struct HeadConvItem {
  const half_t* x;
  const half_t* w;
  const float* bias;
  float* out;
  float* partial;
  int P, cin, cout, iters, block0;
};
struct HeadConvTab {
  HeadConvItem it[4];
  int count;
};

__global__ __launch_bounds__(256) void conv1x1_small_batch_kernel(HeadConvTab tab) {
  __shared__ __attribute__((aligned(16))) char s_x[4][32 * kSmallRS];
  __shared__ float red[4][2][32];
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const HeadConvItem it = tab.it[k];
  const int bx = (int)blockIdx.x - it.block0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  float s = 0.f, q = 0.f;
  for (int i = 0; i < it.iters; ++i) {
    const int p0 = ((bx * it.iters + i) * 4 + wave) * 32;
    if (p0 >= it.P) break;                                   // (wave-uniform)
    conv1x1_small_group(it.x, it.w, it.bias, it.P, it.cin, it.cout, it.out, p0, s_x[wave]);
    if (it.partial && c < it.cout) {
      const float* so = reinterpret_cast<const float*>(s_x[wave]);
      const int r1 = min(16 * half + 16, it.P - p0);
      for (int r = 16 * half; r < r1; ++r) {
        const float v = so[r * it.cout + c];
        s += v;
        q += v * v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                         // the staged rows have been summed before the next group lands
  }
  if (!it.partial) return;
  s += __shfl(s, (lane + 32) & 63, 64);
  q += __shfl(q, (lane + 32) & 63, 64);
  if (half == 0) { red[wave][0][c] = s; red[wave][1][c] = q; }
  __syncthreads();
  if ((int)threadIdx.x < 2 * it.cout) {
    const int which = threadIdx.x / it.cout, cc = threadIdx.x % it.cout;
    it.partial[((size_t)bx * 2 + which) * it.cout + cc] =
        ((red[0][which][cc] + red[1][which][cc]) + red[2][which][cc]) + red[3][which][cc];
  }
}

// dx[p][ci] (+)= sum_co dz[p][co] * w[ci][co]     (w_ck f16 [cin][32], cols >= cout zero)
// One wave = 32 pixels; the accumulator holds, per lane, 4-channel pieces of ONE pixel, so written directly every
// store touches 32 different lines with 8 bytes each.  Two 32-channel blocks are parked in the wave's LDS tile
// ([32 px][64 ch], row stride 144 B) and leave as whole 128-byte lines (8 lanes per pixel row); the
// accumulate form reads the old gradient the same way.
constexpr int kDgradRS = 64 * 4 + 16;                       // f32 staging: the result is rounded once, after the add
__device__ __forceinline__ void conv1x1_small_dgrad_group(const float* __restrict__ dz,
                                                          const half_t* __restrict__ w_ck,
                                                          int P, int cin, int cout,
                                                          half_t* __restrict__ dx,
                                                          int accumulate, float gscale, int p0, char* so) {
  constexpr int RS = kDgradRS;
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int p = p0 + r;
  const bool ok = p < P;
  const int lrow = lane >> 3, lc = lane & 7;
  half8_t b[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = ks * 16 + 8 * hh + j;
      const bool in = ok && co < cout;                       // (branch-free: see conv1x1_small_group)
      const float t = dz[in ? (size_t)p * cout + co : 0];
      b[ks][j] = (half_t)(in ? t * gscale : 0.f);
    }
  for (int c0 = 0; c0 < cin; c0 += 64) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ct = c0 + t * 32;
      if (ct >= cin) break;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        half8_t a = *reinterpret_cast<const half8_t*>(w_ck + (size_t)(ct + r) * 32 + ks * 16 + 8 * hh);
        acc = OCR_MFMA_32x32x16(a, b[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
        *reinterpret_cast<f32x4*>(so + r * RS + (t * 32 + q * 8 + 4 * hh) * 4) = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 8 * i, pp = p0 + row;
      if (pp < P && c0 + lc * 8 < cin) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(so + row * RS + lc * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(so + row * RS + lc * 32 + 16);
        float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        half_t* dst = dx + (size_t)pp * cin + c0 + lc * 8;
        half8_t o;
        if (accumulate) {
          const half8_t old = *reinterpret_cast<const half8_t*>(dst);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)(f[e] + (float)old[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)f[e];
        }
        *reinterpret_cast<half8_t*>(dst) = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void conv1x1_small_dgrad_kernel(const float* __restrict__ dz,
                                                                  const half_t* __restrict__ w_ck,
                                                                  int P, int cin, int cout,
                                                                  half_t* __restrict__ dx,
                                                                  int accumulate, float gscale) {
  __shared__ __attribute__((aligned(16))) char s_o[4][32 * kDgradRS];
  const int wave = threadIdx.x >> 6;
  conv1x1_small_dgrad_group(dz, w_ck, P, cin, cout, dx, accumulate, gscale, (blockIdx.x * 4 + wave) * 32, s_o[wave]);
}

// ... and for up to four feature maps in one launch (one 128-pixel group per workgroup, as above)
struct HeadDgradItem {
  const float* dz;
  const half_t* w_ck;
  half_t* dx;
  int P, cin, cout, accumulate, block0;
};
struct HeadDgradTab {
  HeadDgradItem it[4];
  int count;
  float gscale;
};
__global__ __launch_bounds__(256) void conv1x1_small_dgrad_batch_kernel(HeadDgradTab tab) {
  __shared__ __attribute__((aligned(16))) char s_o[4][32 * kDgradRS];
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const HeadDgradItem it = tab.it[k];
  const int wave = threadIdx.x >> 6;
  const int p0 = (((int)blockIdx.x - it.block0) * 4 + wave) * 32;
  if (p0 < it.P) conv1x1_small_dgrad_group(it.dz, it.w_ck, it.P, it.cin, it.cout, it.dx, it.accumulate, tab.gscale, p0, s_o[wave]);
}

// partial[s][ci][co] = sum over the strip's pixels of x[p][ci] * dz[p][co]
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_small_wgrad_kernel(const half_t* __restrict__ x,
                                                                  const float* __restrict__ dz,
                                                                  int P, int cin, int strip,
                                                                  float* __restrict__ partial) {
  const int ci = blockIdx.y * 256 + threadIdx.x;
  const int p0 = blockIdx.x * strip;
  int p1 = p0 + strip;
  if (p1 > P) p1 = P;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
  if (ci < cin) {
    for (int p = p0; p < p1; ++p) {
      const float xv = (float)x[(size_t)p * cin + ci];
      const float* d = dz + (size_t)p * COUT;
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[c] += xv * d[c];
    }
    float* dst = partial + ((size_t)blockIdx.x * cin + ci) * COUT;
#pragma unroll
    for (int c = 0; c < COUT; ++c) dst[c] = acc[c];
  }
}

// The same for NARROW feature maps (cin = 8 / 16 / 32 / 64: EAST's 32-channel merge output, 1.6 M pixels at 64 x 640^2):
// thread = one 8-channel chunk of one pixel lane — 16-byte loads of x, dz kept in f32 (the MFMA route pads dz to 64
// 16-bit columns in a pass of its own and multiplies 64 x 64 tiles for a 32 x 9 result) — 8 x COUT accumulators per
// thread, the block's pixel lanes combined through LDS in lane order, one partial [cin][COUT] per block.
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_small_wgrad_narrow_kernel(const half_t* __restrict__ x,
                                                                         const float* __restrict__ dz, int P, int cin,
                                                                         int strip, float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int Q = cin >> 3, lanes = 256 / Q;
  const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
  const int p0 = blockIdx.x * strip;
  int p1 = p0 + strip;
  if (p1 > P) p1 = P;
  float acc[8][COUT];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[e][c] = 0.f;
  for (int p = p0 + pl; p < p1; p += 2 * lanes) {        // two pixels per trip: both rows' loads ahead of the FMAs
    const int pb = p + lanes;
    const bool two = pb < p1;
    const half8_t xa = *reinterpret_cast<const half8_t*>(x + (size_t)p * cin + q * 8);
    half8_t xb = xa;
    if (two) xb = *reinterpret_cast<const half8_t*>(x + (size_t)pb * cin + q * 8);
    float da[COUT], db[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      da[c] = dz[(size_t)p * COUT + c];
      db[c] = two ? dz[(size_t)pb * COUT + c] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float fa = (float)xa[e], fb = (float)xb[e];
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[e][c] += fa * da[c];
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[e][c] += fb * db[c];
    }
  }
#pragma unroll
  for (int c = 0; c < COUT; ++c) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pl * cin + q * 8 + e] = acc[e][c];
    __syncthreads();
    if (threadIdx.x < cin) {
      float tot = 0.f;
      for (int l = 0; l < lanes; ++l) tot += red[l * cin + threadIdx.x];
      partial[((size_t)blockIdx.x * cin + threadIdx.x) * COUT + c] = tot;
    }
  }
}

// ---- weight gradient of the head convolutions for up to four feature maps in one launch -------------------------------
// dw[ci][co] = sum_p x[p][ci] * dz[p][co]: M = ci, N = co (18 padded to 32), K = pixels, both operands K-major in HBM.  The
// construction of wgrad_pw_kernel (conv_wgrad_pw.hip: 64-pixel stages double-buffered in LDS, fragments by
// ds_read_b64_tr_b16 with the K order permuted identically for both operands, buffer loads whose range check does the
// zero padding) on a 128 ci x 32 co block: 8 waves = 4 (ci) x 2 (co), wave tile 32 ci x 16 co.  dz is read as it stands
// — f32 [P][cout] — and rounded to the 16-bit storage type on its way into LDS (the separate pad pass wrote and re-read
// a [P][64] copy).  The launch is bound by the one pass over x: 49 KB of LDS per workgroup, three workgroups per CU.
// Split-K partial blocks go to a [split][cin][32] f32 slab, summed in fixed order by head_wgrad_reduce_kernel.
typedef short hw_short4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) hw_short4v* hw_lds_s4_ptr;
__device__ __forceinline__ half8_t hw_tr_pair16(const char* base, int second_off) {
  hw_short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((hw_lds_s4_ptr)(base));
  hw_short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((hw_lds_s4_ptr)(base + second_off));
  typedef short short8v __attribute__((ext_vector_type(8)));
  short8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(half8_t, v);
}

struct HeadWgradItem {
  const half_t* x;
  const float* dz;
  float* slab;
  int P, cin, cout, m_tiles, tiles_per_split, nci, block0;
};
struct HeadWgradTab {
  HeadWgradItem it[4];
  int count;
};

constexpr int kHwCIB = 128, kHwPXS = 64;
constexpr int kHwXS = kHwCIB * 2 + 32, kHwDS = 32 * 2 + 32;
constexpr int kHwStage = kHwPXS * (kHwXS + kHwDS);

__global__ __launch_bounds__(512) void head_wgrad_kernel(HeadWgradTab tab) {
  constexpr int NT = 512, CIB = kHwCIB, PXS = kHwPXS, XS = kHwXS, DS = kHwDS, STAGE = kHwStage;
  constexpr int XCH = CIB / 8;                      // 16-byte chunks per pixel row of the x tile
  constexpr int NX = PXS * XCH / NT;                // 2
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const HeadWgradItem it = tab.it[k];
  const int b = (int)blockIdx.x - it.block0;
  const int cib = b % it.nci, split = b / it.nci;
  const int ci0 = cib * CIB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wci = wave & 3, wco = wave >> 2;
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  // columns cout..31 of the dz tiles are never written: zero both buffers' dz tiles once
  for (int i = tid; i < 2 * PXS * (DS / 4); i += NT) {
    const int buf = i / (PXS * (DS / 4)), r = i % (PXS * (DS / 4));
    reinterpret_cast<unsigned*>(smem + buf * STAGE + PXS * XS)[r] = 0u;
  }
  f32x4 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  const int a_lane = (4 * g + q) * XS + (wci * 32 + 4 * pp) * 2;
  const int b_lane = (4 * g + q) * DS + (wco * 16 + 4 * pp) * 2;
  const int mt_begin = split * it.tiles_per_split;
  int mt_end = mt_begin + it.tiles_per_split;
  if (mt_end > it.m_tiles) mt_end = it.m_tiles;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(it.x), 0, (int)((size_t)it.P * it.cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(it.dz), 0, (int)((size_t)it.P * it.cout * 4), 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  const int dq = PXS * it.cout / 4;                 // 16-byte pieces of one stage's dz span (64 * cout * 4 B, contiguous)
  u32x4 xr[NX];
  f32x4 dr = {0.f, 0.f, 0.f, 0.f};
  auto load_tile = [&](int mt) {
    const int p0 = mt * PXS;
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int idx = u * NT + tid;
      const int px = idx / XCH, c = idx % XCH;
      const bool ok = p0 + px < it.P;
      const unsigned off = ok ? (unsigned)(((size_t)(p0 + px) * it.cin + ci0 + c * 8) * 2) : OOB;
      xr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
    }
    // whatever lies beyond the last pixel reads as zero (range check, num_records = P * cout * 4); a 16-byte piece that
    // straddles the end (P odd) is fetched dword by dword so that its valid half is not lost with the rest
    const size_t dbytes = (size_t)it.P * it.cout * 4, dby0 = (size_t)p0 * it.cout * 4 + (size_t)tid * 16;
    if (tid >= dq || dby0 >= dbytes || dby0 + 16 <= dbytes) {
      const unsigned doff = (tid < dq && dby0 < dbytes) ? (unsigned)dby0 : OOB;
      dr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, doff, 0, 0));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dr[j] = (dby0 + 4 * j < dbytes) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, (unsigned)(dby0 + 4 * j), 0, 0)) : 0.f;
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
    if (tid < dq) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = tid * 4 + j;
        const int px = e / it.cout, co = e - px * it.cout;
        *reinterpret_cast<half_t*>(ds + px * DS + co * 2) = (half_t)dr[j];
      }
    }
  };
  if (mt_begin < mt_end) {
    load_tile(mt_begin);
    __syncthreads();                                // the zero fill above is done
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
    for (int s2 = 0; s2 < PXS / 32; ++s2) {
      half8_t a[2], bb;
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = hw_tr_pair16(xs + a_lane + s2 * 32 * XS + i * 32, 16 * XS);
      bb = hw_tr_pair16(ds + b_lane + s2 * 32 * DS, 16 * DS);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = OCR_MFMA_16x16x32(a[i], bb, acc[i], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  // D block: lane (li, g) holds rows (ci) 4g..4g+3 of column (co) li
  float* dst = it.slab + ((size_t)split * it.cin + ci0 + wci * 32 + 4 * g) * 32 + wco * 16 + li;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[(size_t)(i * 16 + e) * 32] = acc[i][e];
}

// dw[ci][co] (co < cout) = sum over the splits of slab[split][ci][co], one wave per output element, fixed order
struct HeadWgradRedItem {
  const float* slab;
  float* dw;
  int cin, cout, splits, block0;
};
struct HeadWgradRedTab {
  HeadWgradRedItem it[4];
  int count;
};
__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(HeadWgradRedTab tab) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const HeadWgradRedItem it = tab.it[k];
  const int i = ((int)blockIdx.x - it.block0) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= it.cin * it.cout) return;
  const int ci = i / it.cout, co = i - ci * it.cout;
  float a = 0.f;
  for (int s2 = lane; s2 < it.splits; s2 += 64) a += it.slab[((size_t)s2 * it.cin + ci) * 32 + co];
  a = wave_sum(a);
  if (lane == 0) it.dw[i] = a;
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                    int elems, int S, float scale) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += partial[(size_t)s * elems + i];
  out[i] = a * scale;
}

// ---------------------------------------------------- small-channel f32 ops
// per-channel sum / sum of squares partials of x [P][C]
__device__ __forceinline__ void sc_stats_body(const float* __restrict__ x, int P, int C, float* __restrict__ partial,
                                              int bx, int gdim, float (*red)[256]) {
  const int lanes = 256 / C;
  const int c = threadIdx.x % C, l = threadIdx.x / C;
  float s = 0.f, q = 0.f;
  if (l < lanes) {
    for (size_t p = (size_t)bx * lanes + l; p < (size_t)P; p += (size_t)gdim * lanes) {
      float v = x[p * C + c];
      s += v;
      q += v * v;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    const int which = threadIdx.x / C, cc = threadIdx.x % C;
    float t = 0.f;
    for (int k = 0; k < lanes; ++k) t += red[which][k * C + cc];
    partial[((size_t)bx * 2 + which) * C + cc] = t;
  }
}
__global__ __launch_bounds__(256) void sc_stats_kernel(const float* __restrict__ x, int P, int C,
                                                       float* __restrict__ partial) {
  __shared__ float red[2][256];
  sc_stats_body(x, P, C, partial, (int)blockIdx.x, (int)gridDim.x, red);
}

__device__ __forceinline__ float sc_act(float z, float sc, float sh, int relu) {
  float v = z * sc + sh;
  return (relu && v < 0.f) ? 0.f : v;
}

// legacy tf.image.resize_bilinear x2 (align_corners=False, no half-pixel centres):
// out[2i] = in[i], out[2i+1] = (in[i] + in[min(i+1, H-1)]) / 2, separably.
__device__ __forceinline__ float unpool_at(const float* __restrict__ prev, int img, int oy, int ox,
                                           int lh, int lw, int C, int c) {
  const int y0 = oy >> 1, x0 = ox >> 1;
  const int y1 = (oy & 1) ? (y0 + 1 < lh ? y0 + 1 : lh - 1) : y0;
  const int x1 = (ox & 1) ? (x0 + 1 < lw ? x0 + 1 : lw - 1) : x0;
  const float wy = (oy & 1) ? 0.5f : 0.f, wx = (ox & 1) ? 0.5f : 0.f;
  const float* b = prev + (size_t)img * lh * lw * C + c;
  const float v00 = b[((size_t)y0 * lw + x0) * C], v01 = b[((size_t)y0 * lw + x1) * C];
  const float v10 = b[((size_t)y1 * lw + x0) * C], v11 = b[((size_t)y1 * lw + x1) * C];
  const float top = v00 + (v01 - v00) * wx;
  const float bot = v10 + (v11 - v10) * wx;
  return top + (bot - top) * wy;
}

// out = [act(za*sa+ha)] + [act(zb*sb+hb)] + [unpool2x(prev)]      (any subset)
__global__ void sc_fuse_kernel(const float* __restrict__ za, const float* __restrict__ sa,
                               const float* __restrict__ ha, const float* __restrict__ zb,
                               const float* __restrict__ sb, const float* __restrict__ hb,
                               const float* __restrict__ prev, int n, int h, int w, int C, int relu,
                               float* __restrict__ out) {
  const size_t total = (size_t)n * h * w * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = 0.f;
    if (za) v += sa ? sc_act(za[i], sa[c], ha[c], relu) : za[i];
    if (zb) v += sb ? sc_act(zb[i], sb[c], hb[c], relu) : zb[i];
    if (prev) {
      size_t u = i / C;
      const int ox = (int)(u % w);
      u /= w;
      const int oy = (int)(u % h);
      const int img = (int)(u / h);
      v += unpool_at(prev, img, oy, ox, h >> 1, w >> 1, C, c);
    }
    out[i] = v;
  }
}

// dprev[n, lh, lw, C] = transpose of unpool2x applied to dout[n, 2lh, 2lw, C]
__global__ void sc_unpool_bwd_kernel(const float* __restrict__ dout, int n, int lh, int lw, int C,
                                     float* __restrict__ dprev) {
  const size_t total = (size_t)n * lh * lw * C;
  const int H = lh * 2, W = lw * 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    size_t u = i / C;
    const int x = (int)(u % lw);
    u /= lw;
    const int y = (int)(u % lh);
    const int img = (int)(u / lh);
    const float* b = dout + (size_t)img * H * W * C + c;
    // 1-D weights of input sample y onto output rows: row 2y (1), 2y+1 (.5, or 1 at the
    // clamped last row), 2y-1 (.5)
    int ry[3], rx[3];
    float wy[3], wx[3];
    ry[0] = 2 * y; wy[0] = 1.f;
    ry[1] = 2 * y + 1; wy[1] = (y == lh - 1) ? 1.f : 0.5f;
    ry[2] = 2 * y - 1; wy[2] = (y > 0) ? 0.5f : 0.f;
    rx[0] = 2 * x; wx[0] = 1.f;
    rx[1] = 2 * x + 1; wx[1] = (x == lw - 1) ? 1.f : 0.5f;
    rx[2] = 2 * x - 1; wx[2] = (x > 0) ? 0.5f : 0.f;
    float g = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (wx[d] == 0.f) continue;
        g += wy[a] * wx[d] * b[((size_t)ry[a] * W + rx[d]) * C];
      }
    }
    dprev[i] = g;
  }
}

// BN(+ReLU) backward on [P][C] f32.  MODE 0: partial sums; MODE 1: dz.  (bx, gdim): the block's index and the number of
// blocks of ITS problem — the batched launch below runs several problems in one grid.
template <int MODE>
__device__ __forceinline__ void sc_bn_bwd_body(const float* __restrict__ z, const float* __restrict__ scale,
                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                               const float* __restrict__ invstd, const float* __restrict__ dgamma,
                                               const float* __restrict__ dbeta, const float* __restrict__ dout, int P,
                                               int C, int relu, float inv_count, float* __restrict__ partial,
                                               float* __restrict__ dz, int bx, int gdim, float (*red)[256]) {
  const int lanes = 256 / C;
  const int c = threadIdx.x % C, l = threadIdx.x / C;
  float s = 0.f, q = 0.f;
  if (l < lanes) {
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const float kd = MODE ? dbeta[c] * inv_count : 0.f, kx = MODE ? dgamma[c] * inv_count : 0.f;
    for (size_t p = (size_t)bx * lanes + l; p < (size_t)P; p += (size_t)gdim * lanes) {
      const float zv = z[p * C + c];
      const float a = zv * sc + sh;
      const float g = (!relu || a > 0.f) ? dout[p * C + c] : 0.f;
      const float xh = (zv - mu) * is;
      if (MODE == 0) {
        s += g;
        q += g * xh;
      } else {
        dz[p * C + c] = sc * (g - kd - xh * kx);
      }
    }
  }
  if (MODE == 0) {
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < 2 * C) {
      const int which = threadIdx.x / C, cc = threadIdx.x % C;
      float t = 0.f;
      for (int k = 0; k < lanes; ++k) t += red[which][k * C + cc];
      partial[((size_t)bx * 2 + which) * C + cc] = t;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void sc_bn_bwd_kernel(const float* __restrict__ z,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ dgamma,
                                                        const float* __restrict__ dbeta,
                                                        const float* __restrict__ dout, int P, int C,
                                                        int relu, float inv_count,
                                                        float* __restrict__ partial,
                                                        float* __restrict__ dz) {
  __shared__ float red[2][256];
  sc_bn_bwd_body<MODE>(z, scale, shift, mean, invstd, dgamma, dbeta, dout, P, C, relu, inv_count, partial, dz,
                       (int)blockIdx.x, (int)gridDim.x, red);
}

struct ScBnBwdItem {
  const float *z, *scale, *shift, *mean, *invstd, *dout;
  float *dgamma, *dbeta, *dz, *partial;
  int P, C, relu, T, block0;
};
struct ScBnBwdTab {
  ScBnBwdItem it[4];
  int count;
};
template <int MODE>
__global__ __launch_bounds__(256) void sc_bn_bwd_batch_kernel(ScBnBwdTab tab) {
  __shared__ float red[2][256];
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const ScBnBwdItem it = tab.it[k];
  sc_bn_bwd_body<MODE>(it.z, it.scale, it.shift, it.mean, it.invstd, it.dgamma, it.dbeta, it.dout, it.P, it.C, it.relu,
                       (float)(1.0 / (double)it.P), it.partial, it.dz, (int)blockIdx.x - it.block0, it.T, red);
}

// per-channel column sums (bias gradients of the un-normalised PixelLink heads) for up to four maps: sc_stats_kernel's
// partial rows, batched
struct ScStatsItem {
  const float* x;
  float* partial;
  int P, C, T, block0;
};
struct ScStatsTab {
  ScStatsItem it[4];
  int count;
};
__global__ __launch_bounds__(256) void sc_stats_batch_kernel(ScStatsTab tab) {
  __shared__ float red[2][256];
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const ScStatsItem it = tab.it[k];
  sc_stats_body(it.x, it.P, it.C, it.partial, (int)blockIdx.x - it.block0, it.T, red);
}

// out = act(z * scale + shift) for up to four [P][C] tensors in one launch (the two predication heads' final batch norm)
struct ScActItem {
  const float *z, *scale, *shift;
  float* out;
  size_t total;
  int C, block0, nblocks;
};
struct ScActTab {
  ScActItem it[4];
  int count, relu;
};
__global__ __launch_bounds__(256) void sc_act_batch_kernel(ScActTab tab) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const ScActItem it = tab.it[k];
  const int bx = (int)blockIdx.x - it.block0;
  for (size_t i = (size_t)bx * 256 + threadIdx.x; i < it.total; i += (size_t)it.nblocks * 256) {
    const int c = (int)(i % it.C);
    const float v = it.z[i] * it.scale[c] + it.shift[c];
    it.out[i] = (tab.relu && v < 0.f) ? 0.f : v;
  }
}

// pointwise f32 conv on a channel slice: out[p][oo+co] = b[co] + sum_ci x[p][xo+ci] w[ci][co]
__global__ void sc_pointwise_fwd_kernel(const float* __restrict__ x, int ldx, int xo, int cin,
                                        const float* __restrict__ w, const float* __restrict__ bias,
                                        int P, float* __restrict__ out, int ldo, int oo, int cout) {
  const size_t total = (size_t)P * cout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int co = (int)(i % cout);
    const size_t p = i / cout;
    float a = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < cin; ++ci) a += x[p * ldx + xo + ci] * w[ci * cout + co];
    out[p * ldo + oo + co] = a;
  }
}

// dx[p][xo+ci] = sum_co dout[p][oo+co] w[ci][co]
__global__ void sc_pointwise_dgrad_kernel(const float* __restrict__ dout, int ldo, int oo, int cout,
                                          const float* __restrict__ w, int P,
                                          float* __restrict__ dx, int ldx, int xo, int cin) {
  const size_t total = (size_t)P * cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ci = (int)(i % cin);
    const size_t p = i / cin;
    float a = 0.f;
    for (int co = 0; co < cout; ++co) a += dout[p * ldo + oo + co] * w[ci * cout + co];
    dx[p * ldx + xo + ci] = a;
  }
}

// Fast paths of the two kernels above for the shapes the PixelLink predication convs have
// (16 -> 16 link, 2 -> 2 pixel; square so one template serves forward and input gradient): one
// thread per PIXEL — the input row is read once into registers, the C x C weights sit in LDS, and a
// thread writes its C outputs as contiguous floats.  Same summation order per output as the generic
// kernels (ascending input channel).
template <int C, bool TRANSPOSE>
__global__ __launch_bounds__(256) void sc_pointwise_square_kernel(const float* __restrict__ in, int ldi, int io,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ bias, int P,
                                                                  float* __restrict__ out, int ldo, int oo) {
  __shared__ float ws[C * C];
  // forward: out[co] = b[co] + sum_ci in[ci] * w[ci][co];  dgrad (TRANSPOSE): out[ci] = sum_co in[co] * w[ci][co]
  for (int i = threadIdx.x; i < C * C; i += 256) ws[i] = TRANSPOSE ? w[(i % C) * C + i / C] : w[i];   // ws[k][j]: in k -> out j
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)P; p += (size_t)gridDim.x * 256) {
    float v[C];
#pragma unroll
    for (int k = 0; k < C; ++k) v[k] = in[p * ldi + io + k];
#pragma unroll
    for (int j = 0; j < C; ++j) {
      float a = (!TRANSPOSE && bias) ? bias[j] : 0.f;
#pragma unroll
      for (int k = 0; k < C; ++k) a += v[k] * ws[k * C + j];
      out[p * ldo + oo + j] = a;
    }
  }
}

// partial[blk][ci*cout+co] (weights) and partial[blk][cin*cout+co] (bias) over the block's strip
// of pixels; the strip is staged through LDS 64 pixels at a time, one thread per output pair.
__global__ __launch_bounds__(256) void sc_pointwise_wgrad_kernel(
    const float* __restrict__ x, int ldx, int xo, int cin, const float* __restrict__ dout, int ldo,
    int oo, int cout, int P, int strip, float* __restrict__ partial) {
  __shared__ float xs[64 * 32], ds[64 * 32];
  const int pairs = cin * cout + cout;  // weights then bias
  const int p0 = blockIdx.x * strip;
  int p1 = p0 + strip;
  if (p1 > P) p1 = P;
  float acc[2] = {0.f, 0.f};  // pairs <= 512
  for (int pb = p0; pb < p1; pb += 64) {
    const int np = (p1 - pb) < 64 ? (p1 - pb) : 64;
    __syncthreads();
    for (int i = threadIdx.x; i < np * cin; i += 256) xs[i] = x[(size_t)(pb + i / cin) * ldx + xo + i % cin];
    for (int i = threadIdx.x; i < np * cout; i += 256) ds[i] = dout[(size_t)(pb + i / cout) * ldo + oo + i % cout];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = threadIdx.x + k * 256;
      if (j < pairs) {
        const bool is_b = j >= cin * cout;
        const int ci = is_b ? 0 : j / cout, co = is_b ? j - cin * cout : j % cout;
        float a = acc[k];
        if (is_b) for (int q = 0; q < np; ++q) a += ds[q * cout + co];
        else for (int q = 0; q < np; ++q) a += xs[q * cin + ci] * ds[q * cout + co];
        acc[k] = a;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = threadIdx.x + k * 256;
    if (j < pairs) partial[(size_t)blockIdx.x * pairs + j] = acc[k];
  }
}

// 16 -> 16 fast path of sc_pointwise_wgrad_kernel: the block's 256 threads are 16 pair blocks (4 ci x 4
// co, 16 accumulators in registers) x 16 pixel lanes; a pixel lane walks every 16th pixel of the
// strip reading its 4 + 4 values straight from global memory (no LDS in the loop: the generic kernel
// pays two LDS reads per FMA).  The 16 lane sums are combined in lane order: deterministic.
__device__ __forceinline__ void sc_pointwise_wgrad16_body(
    const float* __restrict__ x, int ldx, int xo, const float* __restrict__ dout, int ldo, int oo, int P,
    int strip, float* __restrict__ partial, int bx, float (*red)[16 * 16 + 16]) {
  const int pb = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int cib = pb >> 2, cob = pb & 3;
  const int p0 = bx * strip;
  int p1 = p0 + strip;
  if (p1 > P) p1 = P;
  float acc[4][4], bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bacc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  }
  for (int q = p0 + pl; q < p1; q += 16) {
    float xv[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[i] = x[(size_t)q * ldx + xo + cib * 4 + i];
      dv[i] = dout[(size_t)q * ldo + oo + cob * 4 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bacc[i] += dv[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += xv[i] * dv[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[pl][(cib * 4 + i) * 16 + cob * 4 + j] = acc[i][j];
    if (cib == 0) red[pl][256 + cob * 4 + i] = bacc[i];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 16 * 16 + 16; j += 256) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][j];
    partial[(size_t)bx * (16 * 16 + 16) + j] = t;
  }
}
__global__ __launch_bounds__(256) void sc_pointwise_wgrad16_kernel(
    const float* __restrict__ x, int ldx, int xo, const float* __restrict__ dout, int ldo, int oo, int P,
    int strip, float* __restrict__ partial) {
  __shared__ float red[16][16 * 16 + 16];
  sc_pointwise_wgrad16_body(x, ldx, xo, dout, ldo, oo, P, strip, partial, (int)blockIdx.x, red);
}

// 2 -> 2 fast path (the pixel predication conv): every thread walks pixels and keeps all 2x2 + 2 sums;
// wave butterfly, then the four waves in order.
__device__ __forceinline__ void sc_pointwise_wgrad2_body(
    const float* __restrict__ x, int ldx, int xo, const float* __restrict__ dout, int ldo, int oo, int P,
    int strip, float* __restrict__ partial, int bx, float (*red)[6]) {
  const int p0 = bx * strip;
  int p1 = p0 + strip;
  if (p1 > P) p1 = P;
  float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // w00 w01 w10 w11 b0 b1
  for (int q = p0 + (int)threadIdx.x; q < p1; q += 256) {
    const float x0 = x[(size_t)q * ldx + xo], x1 = x[(size_t)q * ldx + xo + 1];
    const float d0 = dout[(size_t)q * ldo + oo], d1 = dout[(size_t)q * ldo + oo + 1];
    a[0] += x0 * d0; a[1] += x0 * d1; a[2] += x1 * d0; a[3] += x1 * d1; a[4] += d0; a[5] += d1;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    a[j] = wave_sum(a[j]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = a[j];
  }
  __syncthreads();
  if (threadIdx.x < 6)
    partial[(size_t)bx * 6 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void sc_pointwise_wgrad2_kernel(
    const float* __restrict__ x, int ldx, int xo, const float* __restrict__ dout, int ldo, int oo, int P,
    int strip, float* __restrict__ partial) {
  __shared__ float red[4][6];
  sc_pointwise_wgrad2_body(x, ldx, xo, dout, ldo, oo, P, strip, partial, (int)blockIdx.x, red);
}

// both predication convolutions' weight (+ bias) gradients in one launch: blocks [0, B) the 16 -> 16 link head on
// x[:, 2:18], blocks [B, 2B) the 2 -> 2 pixel head on x[:, 0:2]; then ONE fixed-order reduction of both partial slabs
__global__ __launch_bounds__(256) void sc_pointwise_pair_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dza, const float* __restrict__ dzb, int P, int B, int strip,
    float* __restrict__ part_a, float* __restrict__ part_b) {
  __shared__ float red16[16][16 * 16 + 16];
  __shared__ float red2[4][6];
  if ((int)blockIdx.x < B) sc_pointwise_wgrad16_body(x, 18, 2, dzb, 16, 0, P, strip, part_b, (int)blockIdx.x, red16);
  else sc_pointwise_wgrad2_body(x, 18, 0, dza, 2, 0, P, strip, part_a, (int)blockIdx.x - B, red2);
}
struct SumRowsItem {
  const float* partial;
  float *out_a, *out_b;
  int n_a, elems, S, block0;
};
struct SumRowsTab {
  SumRowsItem it[4];
  int count;
};
__global__ __launch_bounds__(256) void sum_rows_batch_kernel(SumRowsTab tab) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < tab.count && (int)blockIdx.x >= tab.it[j].block0) k = j;
  const SumRowsItem it = tab.it[k];
  const int i = ((int)blockIdx.x - it.block0) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= it.elems) return;
  float a = 0.f;
  for (int s2 = lane; s2 < it.S; s2 += 64) a += it.partial[(size_t)s2 * it.elems + i];
  a = wave_sum(a);
  if (lane == 0) {
    if (i < it.n_a) { if (it.out_a) it.out_a[i] = a; }
    else if (it.out_b) it.out_b[i - it.n_a] = a;
  }
}

// ---- the two predication convolutions as ONE pass over the fused head tensor -------------------------------------------
// x [P][CA + CB] (channels [0, CA): pixel head, [CA, CA + CB): link head; nets/model_vgg_16.py:166,173,
// nets/pixellink.py:61,67): za = x[:, :CA] wa (+ ba), zb = x[:, CA:] wb (+ bb), one thread per pixel, the weights in LDS,
// the same summation order per output as sc_pointwise_square_kernel (ascending input channel) — and, for the batch-normed
// variant, the statistics of both outputs in the same pass (per-thread sums over its pixels, one partial row per block).
template <int CA, int CB>
__global__ __launch_bounds__(256) void sc_pointwise_pair_fwd_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ wa, const float* __restrict__ ba,
                                                                    const float* __restrict__ wb, const float* __restrict__ bb,
                                                                    int P, float* __restrict__ za, float* __restrict__ zb,
                                                                    float* __restrict__ pa, float* __restrict__ pb) {
  constexpr int C = CA + CB;
  __shared__ float wsa[CA * CA], wsb[CB * CB];
  __shared__ float red[4][2 * C];
  for (int i = threadIdx.x; i < CA * CA; i += 256) wsa[i] = wa[i];
  for (int i = threadIdx.x; i < CB * CB; i += 256) wsb[i] = wb[i];
  __syncthreads();
  float s[C], q[C];
#pragma unroll
  for (int j = 0; j < C; ++j) { s[j] = 0.f; q[j] = 0.f; }
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)P; p += (size_t)gridDim.x * 256) {
    float v[C], oa[CA], ob[CB];
    if constexpr ((C & 1) == 0) {                     // rows of C floats are 8-byte aligned: half the load instructions
#pragma unroll
      for (int k = 0; k < C / 2; ++k) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(x + p * C + 2 * k);
        v[2 * k] = t[0];
        v[2 * k + 1] = t[1];
      }
    } else {
#pragma unroll
      for (int k = 0; k < C; ++k) v[k] = x[p * C + k];
    }
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      float a = ba ? ba[j] : 0.f;
#pragma unroll
      for (int k = 0; k < CA; ++k) a += v[k] * wsa[k * CA + j];
      oa[j] = a;
      s[j] += a;
      q[j] += a * a;
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      float a = bb ? bb[j] : 0.f;
#pragma unroll
      for (int k = 0; k < CB; ++k) a += v[CA + k] * wsb[k * CB + j];
      ob[j] = a;
      s[CA + j] += a;
      q[CA + j] += a * a;
    }
    if constexpr (CA == 2) *reinterpret_cast<f32x2*>(za + p * CA) = f32x2{oa[0], oa[1]};
    else
#pragma unroll
      for (int j = 0; j < CA; ++j) za[p * CA + j] = oa[j];
    if constexpr ((CB & 3) == 0) {
#pragma unroll
      for (int j = 0; j < CB / 4; ++j)
        *reinterpret_cast<f32x4*>(zb + p * CB + 4 * j) = f32x4{ob[4 * j], ob[4 * j + 1], ob[4 * j + 2], ob[4 * j + 3]};
    } else {
#pragma unroll
      for (int j = 0; j < CB; ++j) zb[p * CB + j] = ob[j];
    }
  }
  if (!pa) return;                                   // (uniform: no statistics wanted)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < C; ++j) {
    const float ts = wave_sum(s[j]), tq = wave_sum(q[j]);
    if (lane == 0) { red[wave][j] = ts; red[wave][C + j] = tq; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    const int which = threadIdx.x / C, j = threadIdx.x % C;
    const float t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    if (j < CA) pa[((size_t)blockIdx.x * 2 + which) * CA + j] = t;
    else pb[((size_t)blockIdx.x * 2 + which) * CB + (j - CA)] = t;
  }
}

// dx[p] = [dza[p] wa^T | dzb[p] wb^T]: the input gradient of both predication convolutions, every channel of dx written
template <int CA, int CB>
__global__ __launch_bounds__(256) void sc_pointwise_pair_dgrad_kernel(const float* __restrict__ dza, const float* __restrict__ wa,
                                                                      const float* __restrict__ dzb, const float* __restrict__ wb,
                                                                      int P, float* __restrict__ dx) {
  constexpr int C = CA + CB;
  __shared__ float wsa[CA * CA], wsb[CB * CB];       // ws[k][j]: in k -> out j  (transposed weights)
  for (int i = threadIdx.x; i < CA * CA; i += 256) wsa[i] = wa[(i % CA) * CA + i / CA];
  for (int i = threadIdx.x; i < CB * CB; i += 256) wsb[i] = wb[(i % CB) * CB + i / CB];
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)P; p += (size_t)gridDim.x * 256) {
    float va[CA], vb[CB], o[C];
    if constexpr (CA == 2) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(dza + p * CA);
      va[0] = t[0];
      va[1] = t[1];
    } else {
#pragma unroll
      for (int k = 0; k < CA; ++k) va[k] = dza[p * CA + k];
    }
    if constexpr ((CB & 3) == 0) {
#pragma unroll
      for (int k = 0; k < CB / 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(dzb + p * CB + 4 * k);
        vb[4 * k] = t[0]; vb[4 * k + 1] = t[1]; vb[4 * k + 2] = t[2]; vb[4 * k + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int k = 0; k < CB; ++k) vb[k] = dzb[p * CB + k];
    }
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < CA; ++k) a += va[k] * wsa[k * CA + j];
      o[j] = a;
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < CB; ++k) a += vb[k] * wsb[k * CB + j];
      o[CA + j] = a;
    }
    if constexpr ((C & 1) == 0) {
#pragma unroll
      for (int j = 0; j < C / 2; ++j) *reinterpret_cast<f32x2*>(dx + p * C + 2 * j) = f32x2{o[2 * j], o[2 * j + 1]};
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) dx[p * C + j] = o[j];
    }
  }
}

unsigned sgrid(size_t items) {
  size_t b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

int sc_blocks(int P, int C) {
  const int lanes = 256 / C;
  int b = ocr_cdiv(P, lanes * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return b;
}

}  // namespace

extern "C" int ocr_conv1x1_small_f16(const void* x, const void* w_kc32, const void* bias, int P,
                                     int cin, int cout, void* out_f32, void* stream) {
  OCR_CHECK_ARG(x && w_kc32 && out_f32 && P > 0);
  OCR_CHECK_SHAPE(cin % 16 == 0 && cout >= 1 && cout <= 32);
  hipLaunchKernelGGL(conv1x1_small_kernel, dim3(ocr_cdiv(P, 128)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const half_t*>(x),
                     static_cast<const half_t*>(w_kc32), static_cast<const float*>(bias), P, cin,
                     cout, static_cast<float*>(out_f32));
  return ocr_launch_status();
}

extern "C" int ocr_conv1x1_small_dgrad_f16(const void* dz_f32, const void* w_ck32, int P, int cin,
                                           int cout, float grad_scale, void* dx_f16, int accumulate,
                                           void* stream) {
  OCR_CHECK_ARG(dz_f32 && w_ck32 && dx_f16 && P > 0);
  OCR_CHECK_SHAPE(cin % 32 == 0 && cout >= 1 && cout <= 32);
  hipLaunchKernelGGL(conv1x1_small_dgrad_kernel, dim3(ocr_cdiv(P, 128)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(dz_f32),
                     static_cast<const half_t*>(w_ck32), P, cin, cout,
                     static_cast<half_t*>(dx_f16), accumulate, grad_scale);
  return ocr_launch_status();
}

static int small_wgrad_strips(int P) {
  int s = ocr_cdiv(P, 2048);
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return s;
}

namespace {
// dz f32 [P][C] -> f16 [P][64] (zero padded): the N operand of the MFMA weight-gradient kernel
__global__ void pad_dz_kernel(const float* __restrict__ dz, int P, int C, half_t* __restrict__ out) {
  const size_t total = (size_t)P * 8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t p = i >> 3;
    const int c0 = (int)(i & 7) * 8;
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((c0 + e < C) ? dz[p * C + c0 + e] : 0.f);
    *reinterpret_cast<half8_t*>(out + i * 8) = o;
  }
}
__global__ void take_cols_kernel(const float* __restrict__ in, int rows, int C, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * C) out[i] = in[(size_t)(i / C) * 64 + i % C];
}
}  // namespace

// MFMA route: P % 32 == 0 and cin % 64 == 0 (the 1x1 weight gradient with cout padded to 64);
// otherwise the VALU strip kernel.
static bool small_wgrad_narrow(int cin) { return cin == 8 || cin == 16 || cin == 32 || cin == 64; }
static int small_wgrad_narrow_strips(int P) {
  int s = ocr_cdiv(P, 1024);
  if (s > 4096) s = 4096;
  return s;
}
static bool small_wgrad_mfma(int P, int cin) { return !small_wgrad_narrow(cin) && P % 32 == 0 && cin % 32 == 0; }

static ocr_conv_desc small_wgrad_desc(int P, int cin) {
  ocr_conv_desc d = {1, P / 32, 32, cin, P / 32, 32, 64, 1, 1, 1, 1, 0, 0, 0, 0};
  return d;
}

static size_t al256(size_t v) { return (v + 255) / 256 * 256; }

extern "C" size_t ocr_conv1x1_small_wgrad_workspace(int P, int cin, int cout) {
  if (small_wgrad_mfma(P, cin)) {
    ocr_conv_desc d = small_wgrad_desc(P, cin);
    return al256((size_t)P * 64 * 2) + al256((size_t)cin * 64 * 4) + ocr_conv2d_wgrad_workspace(&d);
  }
  if (small_wgrad_narrow(cin)) return (size_t)small_wgrad_narrow_strips(P) * cin * cout * sizeof(float);
  return (size_t)small_wgrad_strips(P) * cin * cout * sizeof(float);
}

extern "C" int ocr_conv1x1_small_wgrad_f16(const void* x, const void* dz_f32, int P, int cin,
                                           int cout, void* dw_f32, void* workspace, size_t ws_bytes,
                                           void* stream) {
  OCR_CHECK_ARG(x && dz_f32 && dw_f32 && workspace && P > 0);
  OCR_CHECK_SHAPE(cout >= 1 && cout <= 32);
  if (ws_bytes < ocr_conv1x1_small_wgrad_workspace(P, cin, cout)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (small_wgrad_mfma(P, cin)) {
    char* ws = static_cast<char*>(workspace);
    half_t* dz16 = reinterpret_cast<half_t*>(ws);
    float* dw64 = reinterpret_cast<float*>(ws + al256((size_t)P * 64 * 2));
    char* slabs = ws + al256((size_t)P * 64 * 2) + al256((size_t)cin * 64 * 4);
    hipLaunchKernelGGL(pad_dz_kernel, dim3(sgrid((size_t)P * 8)), dim3(256), 0, st,
                       static_cast<const float*>(dz_f32), P, cout, dz16);
    ocr_conv_desc d = small_wgrad_desc(P, cin);
    int rc = ocr_conv2d_wgrad_f16(&d, x, dz16, dw64, slabs, ocr_conv2d_wgrad_workspace(&d), stream);
    if (rc != OCR_OK) return rc;
    hipLaunchKernelGGL(take_cols_kernel, dim3(ocr_cdiv(cin * cout, 256)), dim3(256), 0, st, dw64, cin,
                       cout, static_cast<float*>(dw_f32));
    return ocr_launch_status();
  }
  const half_t* xp = static_cast<const half_t*>(x);
  const float* dp = static_cast<const float*>(dz_f32);
  float* ws = static_cast<float*>(workspace);
  if (small_wgrad_narrow(cin)) {
    const int S = small_wgrad_narrow_strips(P);
    const int strip = ocr_cdiv(P, S);
    switch (cout) {
      case 1: hipLaunchKernelGGL(conv1x1_small_wgrad_narrow_kernel<1>, dim3(S), dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
      case 2: hipLaunchKernelGGL(conv1x1_small_wgrad_narrow_kernel<2>, dim3(S), dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
      case 8: hipLaunchKernelGGL(conv1x1_small_wgrad_narrow_kernel<8>, dim3(S), dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
      case 9: hipLaunchKernelGGL(conv1x1_small_wgrad_narrow_kernel<9>, dim3(S), dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
      case 16: hipLaunchKernelGGL(conv1x1_small_wgrad_narrow_kernel<16>, dim3(S), dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
      case 18: hipLaunchKernelGGL(conv1x1_small_wgrad_narrow_kernel<18>, dim3(S), dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
      default: return OCR_ERR_UNSUPPORTED;
    }
    const int elems = cin * cout;
    hipLaunchKernelGGL(ocr_sum_rows_kernel, dim3(sum_rows_grid(elems)), dim3(256), 0, st, ws,
                       static_cast<float*>(dw_f32), elems, S, 1.f);
    return ocr_launch_status();
  }
  const int S = small_wgrad_strips(P);
  const int strip = ocr_cdiv(P, S);
  dim3 grid(S, ocr_cdiv(cin, 256));
  switch (cout) {
    case 1: hipLaunchKernelGGL(conv1x1_small_wgrad_kernel<1>, grid, dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
    case 2: hipLaunchKernelGGL(conv1x1_small_wgrad_kernel<2>, grid, dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
    case 8: hipLaunchKernelGGL(conv1x1_small_wgrad_kernel<8>, grid, dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
    case 9: hipLaunchKernelGGL(conv1x1_small_wgrad_kernel<9>, grid, dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
    case 16: hipLaunchKernelGGL(conv1x1_small_wgrad_kernel<16>, grid, dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
    case 18: hipLaunchKernelGGL(conv1x1_small_wgrad_kernel<18>, grid, dim3(256), 0, st, xp, dp, P, cin, strip, ws); break;
    default: return OCR_ERR_UNSUPPORTED;
  }
  const int elems = cin * cout;
  hipLaunchKernelGGL(ocr_sum_rows_kernel, dim3(sum_rows_grid(elems)), dim3(256), 0, st, ws,
                     static_cast<float*>(dw_f32), elems, S, 1.f);
  return ocr_launch_status();
}

extern "C" int ocr_sc_num_partials(int P, int C) {
  if (P <= 0 || C <= 0 || C > 128) return OCR_ERR_UNSUPPORTED;
  return sc_blocks(P, C);
}

extern "C" int ocr_sc_stats(const void* x, int P, int C, void* partial, void* stream) {
  OCR_CHECK_ARG(x && partial && P > 0);
  OCR_CHECK_SHAPE(C > 0 && C <= 128);
  hipLaunchKernelGGL(sc_stats_kernel, dim3(sc_blocks(P, C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(x), P, C,
                     static_cast<float*>(partial));
  return ocr_launch_status();
}

extern "C" int ocr_sc_fuse(const void* za, const void* sa, const void* ha, const void* zb,
                           const void* sb, const void* hb, const void* prev, int n, int h, int w,
                           int C, int relu, void* out, void* stream) {
  OCR_CHECK_ARG(out && n > 0 && h > 0 && w > 0 && C > 0);
  OCR_CHECK_ARG((sa == nullptr) == (ha == nullptr) && (sb == nullptr) == (hb == nullptr));
  OCR_CHECK_ARG(!prev || (h % 2 == 0 && w % 2 == 0));
  hipLaunchKernelGGL(sc_fuse_kernel, dim3(sgrid((size_t)n * h * w * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(za),
                     static_cast<const float*>(sa), static_cast<const float*>(ha),
                     static_cast<const float*>(zb), static_cast<const float*>(sb),
                     static_cast<const float*>(hb), static_cast<const float*>(prev), n, h, w, C,
                     relu, static_cast<float*>(out));
  return ocr_launch_status();
}

extern "C" int ocr_sc_unpool_bwd(const void* dout, int n, int lh, int lw, int C, void* dprev,
                                 void* stream) {
  OCR_CHECK_ARG(dout && dprev && n > 0 && lh > 0 && lw > 0 && C > 0);
  hipLaunchKernelGGL(sc_unpool_bwd_kernel, dim3(sgrid((size_t)n * lh * lw * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(dout), n, lh, lw, C,
                     static_cast<float*>(dprev));
  return ocr_launch_status();
}

extern "C" size_t ocr_bn_reduce_workspace(int T, int C);
extern "C" int ocr_sc_bn_bwd(const void* z, const void* scale, const void* shift,
                             const void* save_mean, const void* save_invstd, const void* dout, int P,
                             int C, int relu, void* dgamma, void* dbeta, void* dz, void* partial,
                             void* stream) {
  OCR_CHECK_ARG(z && scale && shift && save_mean && save_invstd && dout && dgamma && dbeta && dz &&
                partial);
  OCR_CHECK_SHAPE(C > 0 && C <= 128 && P > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = sc_blocks(P, C);
  const float* zp = static_cast<const float*>(z);
  hipLaunchKernelGGL(sc_bn_bwd_kernel<0>, dim3(T), dim3(256), 0, st, zp,
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     (const float*)nullptr, (const float*)nullptr, static_cast<const float*>(dout),
                     P, C, relu, 0.f, static_cast<float*>(partial), (float*)nullptr);
  // partial [T][2][C]: sum rows into dbeta / dgamma (T <= 1024, tiny)
  float* part = static_cast<float*>(partial);
  hipLaunchKernelGGL(ocr_sum_rows_split_kernel, dim3(sum_rows_grid(2 * C)), dim3(256), 0, st, part,
                     static_cast<float*>(dbeta), C, static_cast<float*>(dgamma), 2 * C, T, 1.f);
  hipLaunchKernelGGL(sc_bn_bwd_kernel<1>, dim3(T), dim3(256), 0, st, zp,
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     static_cast<const float*>(dgamma), static_cast<const float*>(dbeta),
                     static_cast<const float*>(dout), P, C, relu, (float)(1.0 / (double)P),
                     (float*)nullptr, static_cast<float*>(dz));
  return ocr_launch_status();
}

extern "C" int ocr_sc_pointwise_fwd(const void* x, int ldx, int xo, int cin, const void* w,
                                    const void* bias, int P, void* out, int ldo, int oo, int cout,
                                    void* stream) {
  OCR_CHECK_ARG(x && w && out && P > 0 && cin > 0 && cout > 0);
  if (cin == cout && (cin == 16 || cin == 2)) {
    const dim3 grid(sgrid((size_t)P));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cin == 16)
      hipLaunchKernelGGL((sc_pointwise_square_kernel<16, false>), grid, dim3(256), 0, st, static_cast<const float*>(x),
                         ldx, xo, static_cast<const float*>(w), static_cast<const float*>(bias), P,
                         static_cast<float*>(out), ldo, oo);
    else
      hipLaunchKernelGGL((sc_pointwise_square_kernel<2, false>), grid, dim3(256), 0, st, static_cast<const float*>(x),
                         ldx, xo, static_cast<const float*>(w), static_cast<const float*>(bias), P,
                         static_cast<float*>(out), ldo, oo);
    return ocr_launch_status();
  }
  hipLaunchKernelGGL(sc_pointwise_fwd_kernel, dim3(sgrid((size_t)P * cout)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(x), ldx, xo, cin,
                     static_cast<const float*>(w), static_cast<const float*>(bias), P,
                     static_cast<float*>(out), ldo, oo, cout);
  return ocr_launch_status();
}

extern "C" int ocr_sc_pointwise_dgrad(const void* dout, int ldo, int oo, int cout, const void* w,
                                      int P, void* dx, int ldx, int xo, int cin, void* stream) {
  OCR_CHECK_ARG(dout && w && dx && P > 0 && cin > 0 && cout > 0);
  if (cin == cout && (cin == 16 || cin == 2)) {
    const dim3 grid(sgrid((size_t)P));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cin == 16)
      hipLaunchKernelGGL((sc_pointwise_square_kernel<16, true>), grid, dim3(256), 0, st,
                         static_cast<const float*>(dout), ldo, oo, static_cast<const float*>(w),
                         (const float*)nullptr, P, static_cast<float*>(dx), ldx, xo);
    else
      hipLaunchKernelGGL((sc_pointwise_square_kernel<2, true>), grid, dim3(256), 0, st,
                         static_cast<const float*>(dout), ldo, oo, static_cast<const float*>(w),
                         (const float*)nullptr, P, static_cast<float*>(dx), ldx, xo);
    return ocr_launch_status();
  }
  hipLaunchKernelGGL(sc_pointwise_dgrad_kernel, dim3(sgrid((size_t)P * cin)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(dout), ldo, oo,
                     cout, static_cast<const float*>(w), P, static_cast<float*>(dx), ldx, xo, cin);
  return ocr_launch_status();
}

extern "C" size_t ocr_sc_pointwise_wgrad_workspace(int cin, int cout) {
  return (size_t)(256 + 1) * (cin * cout + cout) * sizeof(float);
}

// dw [cin][cout] and db [cout] (db may be NULL)
extern "C" int ocr_sc_pointwise_wgrad(const void* x, int ldx, int xo, int cin, const void* dout,
                                      int ldo, int oo, int cout, int P, void* dw, void* db,
                                      void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(x && dout && dw && workspace && P > 0 && cin > 0 && cout > 0);
  if (ws_bytes < ocr_sc_pointwise_wgrad_workspace(cin, cout)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = 256;
  const int pairs = cin * cout + cout;
  float* ws = static_cast<float*>(workspace);
  OCR_CHECK_SHAPE(cin <= 32 && cout <= 32 && pairs <= 512);
  if (cin == 16 && cout == 16)
    hipLaunchKernelGGL(sc_pointwise_wgrad16_kernel, dim3(B), dim3(256), 0, st, static_cast<const float*>(x), ldx,
                       xo, static_cast<const float*>(dout), ldo, oo, P, ocr_cdiv(P, B), ws);
  else if (cin == 2 && cout == 2)
    hipLaunchKernelGGL(sc_pointwise_wgrad2_kernel, dim3(B), dim3(256), 0, st, static_cast<const float*>(x), ldx,
                       xo, static_cast<const float*>(dout), ldo, oo, P, ocr_cdiv(P, B), ws);
  else
    hipLaunchKernelGGL(sc_pointwise_wgrad_kernel, dim3(B), dim3(256), 0, st,
                       static_cast<const float*>(x), ldx, xo, cin, static_cast<const float*>(dout),
                       ldo, oo, cout, P, ocr_cdiv(P, B), ws);
  hipLaunchKernelGGL(ocr_sum_rows_split_kernel, dim3(sum_rows_grid(pairs)), dim3(256), 0, st, ws,
                     static_cast<float*>(dw), cin * cout, static_cast<float*>(db), pairs, B, 1.f);
  return ocr_launch_status();
}

// out[c] = sum_p x[p][c]   (bias gradient of the un-normalised PixelLink heads);
// partial must hold (ocr_sc_num_partials + 1) * 2 * C floats
extern "C" int ocr_sc_colsum(const void* x, int P, int C, void* out, void* partial, void* stream) {
  OCR_CHECK_ARG(x && out && partial && P > 0);
  OCR_CHECK_SHAPE(C > 0 && C <= 128);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = sc_blocks(P, C);
  float* part = static_cast<float*>(partial);
  hipLaunchKernelGGL(sc_stats_kernel, dim3(T), dim3(256), 0, st, static_cast<const float*>(x), P, C, part);
  hipLaunchKernelGGL(ocr_sum_rows_split_kernel, dim3(sum_rows_grid(2 * C)), dim3(256), 0, st, part,
                     static_cast<float*>(out), C, (float*)nullptr, 2 * C, T, 1.f);
  return ocr_launch_status();
}

// sigmoid heads of the EAST branch (F_score / geo_map, nets/model_vgg_16.py:129-131) on f32 [n] maps
namespace {
__global__ void sc_sigmoid_kernel(const float* __restrict__ z, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = 1.f / (1.f + expf(-z[i]));
}
__global__ void sc_sigmoid_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                      size_t n, float* __restrict__ dz) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float s = out[i];
    dz[i] = dout[i] * s * (1.f - s);
  }
}
// merged sigmoid heads: one [P][C] pre-activation map -> two activation maps [P][c0], [P][C - c0] (and back)
__global__ void sc_sigmoid_split_kernel(const float* __restrict__ z, unsigned P, unsigned C, unsigned c0,
                                        float* __restrict__ out0, float* __restrict__ out1) {
  const size_t n = (size_t)P * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned p = (unsigned)(i / C), c = (unsigned)(i - (size_t)p * C);
    const float v = 1.f / (1.f + expf(-z[i]));
    if (c < c0) out0[(size_t)p * c0 + c] = v;
    else out1[(size_t)p * (C - c0) + (c - c0)] = v;
  }
}
__global__ void sc_sigmoid_split_bwd_kernel(const float* __restrict__ out0, const float* __restrict__ dout0,
                                            const float* __restrict__ out1, const float* __restrict__ dout1,
                                            unsigned P, unsigned C, unsigned c0, float* __restrict__ dz) {
  const size_t n = (size_t)P * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned p = (unsigned)(i / C), c = (unsigned)(i - (size_t)p * C);
    float s, d;
    if (c < c0) {
      const size_t j = (size_t)p * c0 + c;
      s = out0[j];
      d = dout0 ? dout0[j] : 0.f;
    } else {
      const size_t j = (size_t)p * (C - c0) + (c - c0);
      s = out1[j];
      d = dout1 ? dout1[j] : 0.f;
    }
    dz[i] = d * s * (1.f - s);
  }
}
}  // namespace

extern "C" int ocr_sc_sigmoid_split(const void* z, int P, int C, int c0, void* out0, void* out1, void* stream) {
  OCR_CHECK_ARG(z && out0 && out1 && P > 0 && C > 1 && c0 > 0 && c0 < C);
  hipLaunchKernelGGL(sc_sigmoid_split_kernel, dim3(sgrid((size_t)P * C)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(z), (unsigned)P, (unsigned)C, (unsigned)c0, static_cast<float*>(out0),
                     static_cast<float*>(out1));
  return ocr_launch_status();
}

extern "C" int ocr_sc_sigmoid_split_bwd(const void* out0, const void* dout0, const void* out1, const void* dout1, int P,
                                        int C, int c0, void* dz, void* stream) {
  OCR_CHECK_ARG(out0 && out1 && dz && P > 0 && C > 1 && c0 > 0 && c0 < C);
  hipLaunchKernelGGL(sc_sigmoid_split_bwd_kernel, dim3(sgrid((size_t)P * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(out0), static_cast<const float*>(dout0),
                     static_cast<const float*>(out1), static_cast<const float*>(dout1), (unsigned)P, (unsigned)C,
                     (unsigned)c0, static_cast<float*>(dz));
  return ocr_launch_status();
}

extern "C" int ocr_sc_sigmoid(const void* z, int64_t n, void* out, void* stream) {
  OCR_CHECK_ARG(z && out && n > 0);
  hipLaunchKernelGGL(sc_sigmoid_kernel, dim3(sgrid((size_t)n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(z), (size_t)n,
                     static_cast<float*>(out));
  return ocr_launch_status();
}

extern "C" int ocr_sc_sigmoid_bwd(const void* out, const void* dout, int64_t n, void* dz, void* stream) {
  OCR_CHECK_ARG(out && dout && dz && n > 0);
  hipLaunchKernelGGL(sc_sigmoid_bwd_kernel, dim3(sgrid((size_t)n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(out),
                     static_cast<const float*>(dout), (size_t)n, static_cast<float*>(dz));
  return ocr_launch_status();
}

// =================================================================================================================
// Batched entry points of the fuse heads: one launch per kernel KIND over the (up to four)
// feature maps instead of one per map.  `items` are HOST arrays of small descriptors (like ocr_conv_desc), copied
// into the kernel arguments; the pointers inside them are device pointers.
// =================================================================================================================
extern "C" int ocr_conv1x1_small_batch_rows(int P) {
  if (P <= 0) return OCR_ERR_INVALID_ARG;
  const int iters = ocr_cdiv(P, 128 * 1024);
  return ocr_cdiv(P, 128 * iters);
}

extern "C" int ocr_conv1x1_small_batch_f16(const ocr_head_conv_item* items, int count, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 4);
  HeadConvTab tab;
  tab.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_head_conv_item& a = items[i];
    OCR_CHECK_ARG(a.x && a.w_kc32 && a.out && a.P > 0);
    OCR_CHECK_SHAPE(a.cin % 16 == 0 && a.cout >= 1 && a.cout <= 32);
    const int iters = ocr_cdiv(a.P, 128 * 1024);
    tab.it[i] = HeadConvItem{static_cast<const half_t*>(a.x), static_cast<const half_t*>(a.w_kc32),
                             static_cast<const float*>(a.bias), static_cast<float*>(a.out),
                             static_cast<float*>(a.stats_partial), a.P, a.cin, a.cout, iters, blocks};
    blocks += ocr_cdiv(a.P, 128 * iters);
  }
  for (int i = count; i < 4; ++i) tab.it[i] = tab.it[0];
  hipLaunchKernelGGL(conv1x1_small_batch_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tab);
  return ocr_launch_status();
}

extern "C" int ocr_conv1x1_small_dgrad_batch_f16(const ocr_head_dgrad_item* items, int count, float grad_scale, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 4);
  HeadDgradTab tab;
  tab.count = count;
  tab.gscale = grad_scale;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_head_dgrad_item& a = items[i];
    OCR_CHECK_ARG(a.dz && a.w_ck32 && a.dx && a.P > 0);
    OCR_CHECK_SHAPE(a.cin % 32 == 0 && a.cout >= 1 && a.cout <= 32);
    tab.it[i] = HeadDgradItem{static_cast<const float*>(a.dz), static_cast<const half_t*>(a.w_ck32),
                              static_cast<half_t*>(a.dx), a.P, a.cin, a.cout, a.accumulate, blocks};
    blocks += ocr_cdiv(a.P, 128);
  }
  for (int i = count; i < 4; ++i) tab.it[i] = tab.it[0];
  hipLaunchKernelGGL(conv1x1_small_dgrad_batch_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tab);
  return ocr_launch_status();
}

static void head_wgrad_plan(int P, int cin, int* m_tiles, int* tiles_per_split, int* splits) {
  *m_tiles = ocr_cdiv(P, kHwPXS);
  int sp = ocr_cdiv(*m_tiles, 32);                      // ~32 stages (2048 pixels x 128 channels = 512 KB of x) per workgroup
  if (sp > 256) sp = 256;
  if (sp < 1) sp = 1;
  *tiles_per_split = ocr_cdiv(*m_tiles, sp);
  *splits = ocr_cdiv(*m_tiles, *tiles_per_split);
  (void)cin;
}

extern "C" size_t ocr_conv1x1_small_wgrad_batch_slab_bytes(int P, int cin) {
  if (P <= 0 || cin <= 0) return 0;
  int m, t, sp;
  head_wgrad_plan(P, cin, &m, &t, &sp);
  return (size_t)sp * cin * 32 * sizeof(float);
}

extern "C" int ocr_conv1x1_small_wgrad_batch_f16(const ocr_head_wgrad_item* items, int count, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 4);
  HeadWgradTab tab;
  HeadWgradRedTab red;
  tab.count = red.count = count;
  int blocks = 0, rblocks = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_head_wgrad_item& a = items[i];
    OCR_CHECK_ARG(a.x && a.dz && a.dw && a.slab && a.P > 0);
    OCR_CHECK_SHAPE(a.cin % kHwCIB == 0 && a.cout >= 1 && a.cout <= 32);
    OCR_CHECK_SHAPE((size_t)a.P * a.cin * 2 < (1ull << 31) && (size_t)a.P * a.cout * 4 < (1ull << 31));   // 32-bit buffer offsets
    int m, t, sp;
    head_wgrad_plan(a.P, a.cin, &m, &t, &sp);
    tab.it[i] = HeadWgradItem{static_cast<const half_t*>(a.x), static_cast<const float*>(a.dz),
                              static_cast<float*>(a.slab), a.P, a.cin, a.cout, m, t, a.cin / kHwCIB, blocks};
    blocks += sp * (a.cin / kHwCIB);
    red.it[i] = HeadWgradRedItem{static_cast<const float*>(a.slab), static_cast<float*>(a.dw), a.cin, a.cout, sp, rblocks};
    rblocks += ocr_cdiv(a.cin * a.cout, 4);
  }
  for (int i = count; i < 4; ++i) { tab.it[i] = tab.it[0]; red.it[i] = red.it[0]; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(head_wgrad_kernel, dim3(blocks), dim3(512), 0, st, tab);
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, st, red);
  return ocr_launch_status();
}

extern "C" int ocr_bn_bwd_sums_batch(const ocr_bn_sums_item* items, int count, void* stream);

// BN(+ReLU) backward of up to four [P][C] f32 head tensors: partial sums, their finalisation (dgamma, dbeta) and the apply
// step — three launches for all of them (ocr_sc_bn_bwd: three per tensor).  partial: ocr_sc_num_partials(P, C) * 2 * C floats each.
extern "C" int ocr_sc_bn_bwd_batch(const ocr_sc_bn_bwd_item* items, int count, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 4);
  ScBnBwdTab tab;
  ocr_bn_sums_item sums[4];
  tab.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_sc_bn_bwd_item& a = items[i];
    OCR_CHECK_ARG(a.z && a.scale && a.shift && a.save_mean && a.save_invstd && a.dout && a.dgamma && a.dbeta && a.dz && a.partial);
    OCR_CHECK_SHAPE(a.C > 0 && a.C <= 32 && a.P > 0);
    const int T = sc_blocks(a.P, a.C);
    tab.it[i] = ScBnBwdItem{static_cast<const float*>(a.z), static_cast<const float*>(a.scale),
                            static_cast<const float*>(a.shift), static_cast<const float*>(a.save_mean),
                            static_cast<const float*>(a.save_invstd), static_cast<const float*>(a.dout),
                            static_cast<float*>(a.dgamma), static_cast<float*>(a.dbeta), static_cast<float*>(a.dz),
                            static_cast<float*>(a.partial), a.P, a.C, a.relu, T, blocks};
    sums[i] = ocr_bn_sums_item{a.partial, T, a.C, a.dbeta, a.dgamma};
    blocks += T;
  }
  for (int i = count; i < 4; ++i) tab.it[i] = tab.it[0];
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sc_bn_bwd_batch_kernel<0>, dim3(blocks), dim3(256), 0, st, tab);
  int rc = ocr_bn_bwd_sums_batch(sums, count, stream);
  if (rc != OCR_OK) return rc;
  hipLaunchKernelGGL(sc_bn_bwd_batch_kernel<1>, dim3(blocks), dim3(256), 0, st, tab);
  return ocr_launch_status();
}

// out[c] = sum_p x[p][c] for up to four [P][C] tensors (the bias gradients of the un-normalised heads): two launches
extern "C" int ocr_sc_colsum_batch(const ocr_sc_colsum_item* items, int count, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 4);
  ScStatsTab tab;
  ocr_bn_sums_item sums[4];
  tab.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_sc_colsum_item& a = items[i];
    OCR_CHECK_ARG(a.x && a.out && a.partial && a.P > 0);
    OCR_CHECK_SHAPE(a.C > 0 && a.C <= 32);
    const int T = sc_blocks(a.P, a.C);
    tab.it[i] = ScStatsItem{static_cast<const float*>(a.x), static_cast<float*>(a.partial), a.P, a.C, T, blocks};
    // kind 0 (sums) -> out; kind 1 (sums of squares) -> the row behind the partial rows (scratch, unused)
    sums[i] = ocr_bn_sums_item{a.partial, T, a.C, a.out, static_cast<float*>(a.partial) + (size_t)T * 2 * a.C};
    blocks += T;
  }
  for (int i = count; i < 4; ++i) tab.it[i] = tab.it[0];
  hipLaunchKernelGGL(sc_stats_batch_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tab);
  return ocr_bn_bwd_sums_batch(sums, count, stream);
}

extern "C" int ocr_sc_act_batch(const ocr_sc_act_item* items, int count, int relu, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 4);
  ScActTab tab;
  tab.count = count;
  tab.relu = relu;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_sc_act_item& a = items[i];
    OCR_CHECK_ARG(a.z && a.scale && a.shift && a.out && a.total > 0 && a.C > 0);
    const int nb = (int)sgrid((size_t)a.total);
    tab.it[i] = ScActItem{static_cast<const float*>(a.z), static_cast<const float*>(a.scale),
                          static_cast<const float*>(a.shift), static_cast<float*>(a.out), (size_t)a.total, a.C, blocks, nb};
    blocks += nb;
  }
  for (int i = count; i < 4; ++i) tab.it[i] = tab.it[0];
  hipLaunchKernelGGL(sc_act_batch_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tab);
  return ocr_launch_status();
}

// ---- the two predication convolutions (2 -> 2 pixel, 16 -> 16 link) on the 18-channel fused head tensor, as one pass ---
extern "C" int ocr_sc_pointwise_pair_num_partials(int P) {
  if (P <= 0) return OCR_ERR_INVALID_ARG;
  int b = ocr_cdiv(P, 256 * 4);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return b;
}

extern "C" int ocr_sc_pointwise_pair_fwd(const void* x18, const void* w_px, const void* b_px, const void* w_lk,
                                         const void* b_lk, int P, void* z_px, void* z_lk, void* partial_px,
                                         void* partial_lk, void* stream) {
  OCR_CHECK_ARG(x18 && w_px && w_lk && z_px && z_lk && P > 0 && ((partial_px == nullptr) == (partial_lk == nullptr)));
  hipLaunchKernelGGL((sc_pointwise_pair_fwd_kernel<2, 16>), dim3(ocr_sc_pointwise_pair_num_partials(P)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(x18), static_cast<const float*>(w_px),
                     static_cast<const float*>(b_px), static_cast<const float*>(w_lk), static_cast<const float*>(b_lk), P,
                     static_cast<float*>(z_px), static_cast<float*>(z_lk), static_cast<float*>(partial_px),
                     static_cast<float*>(partial_lk));
  return ocr_launch_status();
}

extern "C" size_t ocr_sc_pointwise_pair_bwd_workspace(void) { return (size_t)256 * (16 * 16 + 16 + 6) * sizeof(float); }

// dx18 = [dz_px w_px^T | dz_lk w_lk^T]; dw / db of both convolutions (db_* may be NULL): three launches
extern "C" int ocr_sc_pointwise_pair_bwd(const void* x18, const void* dz_px, const void* dz_lk, const void* w_px,
                                         const void* w_lk, int P, void* dx18, void* dw_px, void* db_px, void* dw_lk,
                                         void* db_lk, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(x18 && dz_px && dz_lk && w_px && w_lk && dx18 && dw_px && dw_lk && workspace && P > 0);
  if (ws_bytes < ocr_sc_pointwise_pair_bwd_workspace()) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = 256;
  float* part_b = static_cast<float*>(workspace);                  // [B][16*16 + 16]
  float* part_a = part_b + (size_t)B * (16 * 16 + 16);             // [B][2*2 + 2]
  hipLaunchKernelGGL((sc_pointwise_pair_dgrad_kernel<2, 16>), dim3(sgrid((size_t)P)), dim3(256), 0, st,
                     static_cast<const float*>(dz_px), static_cast<const float*>(w_px), static_cast<const float*>(dz_lk),
                     static_cast<const float*>(w_lk), P, static_cast<float*>(dx18));
  hipLaunchKernelGGL(sc_pointwise_pair_wgrad_kernel, dim3(2 * B), dim3(256), 0, st, static_cast<const float*>(x18),
                     static_cast<const float*>(dz_px), static_cast<const float*>(dz_lk), P, B, ocr_cdiv(P, B), part_a, part_b);
  SumRowsTab tab;
  tab.count = 2;
  tab.it[0] = SumRowsItem{part_b, static_cast<float*>(dw_lk), static_cast<float*>(db_lk), 256, 256 + 16, B, 0};
  tab.it[1] = SumRowsItem{part_a, static_cast<float*>(dw_px), static_cast<float*>(db_px), 4, 4 + 2, B, (256 + 16 + 3) / 4};
  tab.it[2] = tab.it[3] = tab.it[0];
  hipLaunchKernelGGL(sum_rows_batch_kernel, dim3((256 + 16 + 3) / 4 + 2), dim3(256), 0, st, tab);
  return ocr_launch_status();
}
