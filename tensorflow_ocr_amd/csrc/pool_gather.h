// The gradient a max-pool routes back to ONE 8-channel chunk of one input position, taken from the pooled gradient
// and the forward's first-maximum index (ocr_maxpool_f16 / ocr_bn_relu_maxpool_f16) instead of from a materialised
// full-resolution gradient tensor: sum, in ascending (ky, kx) order, of dy over the windows that cover the position
// and selected it.  maxpool_bwd_idx_kernel (bn_pool.hip) IS this routine; the consumers that gather on load
// (bn_relu_bwd_gather_kernel, conv_stem_wgrad_kernel) round the sums to 16 bits exactly as that kernel stores them.
#pragma once
#include "common.h"

struct PoolGather {
  const unsigned char* argmax;   // [n][oh][ow][c]: ky*k + kx of the first maximum of each window
  const half_t* dy;              // [n][oh][ow][c]: gradient of the pooled tensor
  int oh, ow, k, stride, pt, pl;
};

// K, S: compile-time window / stride (0: the run-time values of `pg`)
template <int K, int S>
__device__ __forceinline__ void pool_gather8(const PoolGather& pg, int img, int iy, int ix, int c, int ch,
                                             float (&g)[8]) {
  const int k = K ? K : pg.k, stride = S ? S : pg.stride;
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = 0.f;
  // windows covering (iy, ix): ky = (iy + pt) mod stride, + stride, ... (one division per axis, not a divisibility
  // test per tap)
  const int ky0 = (iy + pg.pt) % stride, kx0 = (ix + pg.pl) % stride;
  auto tap = [&](int ky, int kx, int oy, int ox) __attribute__((always_inline)) {
    const size_t o = (((size_t)img * pg.oh + oy) * pg.ow + ox) * c + ch * 8;
    const unsigned long long am = *reinterpret_cast<const unsigned long long*>(pg.argmax + o);
    const half8_t d = *reinterpret_cast<const half8_t*>(pg.dy + o);
    const unsigned pos = (unsigned)(ky * k + kx);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (((am >> (8 * e)) & 0xffu) == pos) g[e] += (float)d[e];
  };
  if constexpr (K != 0) {
    // compile-time window: branch-free — every window slot's operands are requested up front (an out-of-range slot reads
    // element 0 and compares against a position no window holds), so the NJ x NJ loads are in flight together instead
    // of one dependent load per divergent branch; an unselected slot adds +0, which leaves the running sum as it is
    constexpr int NJ = (K + S - 1) / S;          // windows per axis that can cover one position
    int oyv[NJ], oxv[NJ], kyv[NJ], kxv[NJ];
    bool vy[NJ], vx[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      kyv[j] = ky0 + j * S;
      kxv[j] = kx0 + j * S;
      const int ny = iy + pg.pt - kyv[j], nx = ix + pg.pl - kxv[j];
      vy[j] = kyv[j] < K && ny >= 0 && ny / S < pg.oh;
      vx[j] = kxv[j] < K && nx >= 0 && nx / S < pg.ow;
      oyv[j] = vy[j] ? ny / S : 0;
      oxv[j] = vx[j] ? nx / S : 0;
    }
    u32x2 am[NJ * NJ];
    half8_t d[NJ * NJ];
    const unsigned obase = (unsigned)img * (unsigned)pg.oh;      // (32-bit element offsets: the launchers check < 2^31)
#pragma unroll
    for (int jy = 0; jy < NJ; ++jy)
#pragma unroll
      for (int jx = 0; jx < NJ; ++jx) {
        const unsigned o = ((obase + (unsigned)oyv[jy]) * (unsigned)pg.ow + (unsigned)oxv[jx]) * (unsigned)c + (unsigned)ch * 8u;
        am[jy * NJ + jx] = *reinterpret_cast<const u32x2*>(pg.argmax + o);
        d[jy * NJ + jx] = *reinterpret_cast<const half8_t*>(pg.dy + o);
      }
#pragma unroll
    for (int jy = 0; jy < NJ; ++jy)
#pragma unroll
      for (int jx = 0; jx < NJ; ++jx) {
        const unsigned pos = (vy[jy] && vx[jx]) ? (unsigned)(kyv[jy] * K + kxv[jx]) : 255u;
        const u32x2 a = am[jy * NJ + jx];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          g[e] += (((a[e >> 2] >> (8 * (e & 3))) & 0xffu) == pos) ? (float)d[jy * NJ + jx][e] : 0.f;
      }
  } else {
    for (int ky = ky0; ky < k; ky += stride) {
      const int ny = iy + pg.pt - ky;
      if (ny < 0) break;
      const int oy = ny / stride;
      if (oy >= pg.oh) continue;
      for (int kx = kx0; kx < k; kx += stride) {
        const int nx = ix + pg.pl - kx;
        if (nx < 0) break;
        const int ox = nx / stride;
        if (ox >= pg.ow) continue;
        tap(ky, kx, oy, ox);
      }
    }
  }
}

template <int K, int S>
__device__ __forceinline__ half8_t pool_gather8_f16(const PoolGather& pg, int img, int iy, int ix, int c, int ch) {
  float g[8];
  pool_gather8<K, S>(pg, img, iy, ix, c, ch, g);
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)g[e];
  return o;
}
