// Implicit-GEMM NHWC convolution on gfx950 MFMA (f16 in, f32 accumulate, f16 out).
//
// Replaces slim.conv2d on the hot path (reference nets/vgg.py:14-39,
// nets/resnet_v1.py:97-105, nets/model_vgg_16.py:111-131) and, with the
// transposed/flipped weight pack, its input gradient.
//
// Work decomposition (one workgroup = 4 waves = 256 threads):
//   * output tile  = 8 rows x 32 columns of one image (256 pixels) x BN couts
//   * the input halo tile for one CK-wide channel chunk is staged ONCE in LDS
//     ([HT][WT][CK] f16, pixel stride padded by 16 B so ds_read_b128 is
//     conflict-free) and re-read at shifted addresses by all kh*kw taps: no
//     im2col is ever materialised and each input byte crosses L2->LDS once per
//     cout tile;
//   * per tap a [BN][CK] weight slice is streamed through a 2-deep LDS ring
//     (global->register loads of tap t+1 fly under the MFMAs of tap t);
//   * MFMA v_mfma_f32_32x32x16_f16 with A = weights (rows = cout) and
//     B = activations (cols = pixels), so each lane ends up holding 4
//     consecutive couts of one pixel per register quad -> 8-byte packed LDS
//     writes in the epilogue, then 16-byte coalesced row stores to HBM;
//   * the epilogue optionally adds bias / ReLU, accumulates into an existing
//     f16 tensor, and emits per-tile per-cout sum and sum of squares of the
//     stored (f16-rounded) values for training-mode batch norm.
#include "common.h"
#include "conv_epilogue.h"
#include <stdlib.h>

namespace {

struct ConvP {
  int n, h, w, cin, oh, ow, cout, kh, kw, stride, dil, pt, pl, flip, flags;
  int tiles_x, tiles_y, n_tiles, HT, WT, halo_bytes;
  BnRed br;       // br.y != nullptr: fused BN-backward reduction (see conv_epilogue.h)
};


template <int BN, int CK, int WCO>
__global__ __launch_bounds__(512) void conv_igemm_kernel(
    ConvP p, const half_t* __restrict__ x, const half_t* __restrict__ w,
    const float* __restrict__ bias, half_t* __restrict__ y,
    float* __restrict__ stats) {
  constexpr int NT = 512;
  constexpr int WPX = 8 / WCO;
  constexpr int TCO = BN / WCO / 32;
  constexpr int TPX = 256 / WPX / 32;
  constexpr int PSTR = CK * 2 + 16;
  constexpr int KSTEPS = CK / 16;
  constexpr int CPP = CK / 8;  // 16-byte chunks per pixel / per weight row
  constexpr int NWLD = (BN * CPP + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + p.halo_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31;
  const int hh = lane >> 5;
  const int wco = wave % WCO;
  const int wpx = wave / WCO;

  int bid = blockIdx.x;
  const int nt = bid % p.n_tiles;
  int mt = bid / p.n_tiles;
  const int txi = mt % p.tiles_x;
  int tmp = mt / p.tiles_x;
  const int tyi = tmp % p.tiles_y;
  const int img = tmp / p.tiles_y;
  const int co0 = nt * BN;

  const int iy0 = tyi * TILE_H * p.stride - p.pt;
  const int ix0 = txi * TILE_W * p.stride - p.pl;
  const int ntaps = p.kh * p.kw;
  const int nchunks = p.cin / CK;
  const int WT = p.WT;
  const int halo_px = p.HT * WT;

  f32x16 acc[TCO][TPX];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int t = 0; t < TPX; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;

  // per-lane LDS byte offsets
  const int a_lane = (wco * TCO * 32 + r) * PSTR + hh * 16;               // weights
  const int b_lane = (r * p.stride) * PSTR + hh * 16;                     // activations
  const int b_row = p.stride * WT * PSTR;                                 // per tile row

  u32x4 wreg[NWLD];
  auto load_w = [&](int tap, int cc) {
    const int tapw = p.flip ? (ntaps - 1 - tap) : tap;
    const half_t* src = w + ((size_t)tapw * p.cout + co0) * p.cin + cc * CK;
#pragma unroll
    for (int u = 0; u < NWLD; ++u) {
      int idx = u * NT + tid;
      int rr = idx / CPP, c = idx % CPP;
      if (idx < BN * CPP) wreg[u] = *reinterpret_cast<const u32x4*>(src + (size_t)rr * p.cin + c * 8);
    }
  };
  auto store_w = [&](int buf) {
    char* dst = wbuf + buf * (BN * PSTR);
#pragma unroll
    for (int u = 0; u < NWLD; ++u) {
      int idx = u * NT + tid;
      int rr = idx / CPP, c = idx % CPP;
      if (idx < BN * CPP) *reinterpret_cast<u32x4*>(dst + rr * PSTR + c * 16) = wreg[u];
    }
  };

  // halo staging split into issue-early (global -> registers, right after the previous chunk's
  // halo has been handed to LDS) and write-late (registers -> LDS at the chunk boundary), so the
  // global latency of chunk cc+1 hides under the kh*kw taps of chunk cc.
  constexpr int NH = BN >= 128 ? 7 : 1;   // narrow tiles keep their registers for occupancy
  const int halo_total = halo_px * CPP;
  const bool prefetch = BN >= 128 && halo_total <= NH * NT && nchunks > 1;
  u32x4 hreg[NH];
  auto halo_load = [&](int cc) {
    const half_t* xb = x + (size_t)img * p.h * p.w * p.cin + cc * CK;
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int idx = u * NT + tid;
      hreg[u] = u32x4{0u, 0u, 0u, 0u};
      if (idx < halo_total) {
        const int hp = idx / CPP, c = idx % CPP;
        const int hy = hp / WT, hx = hp - hy * WT;
        const int iy = iy0 + hy, ix = ix0 + hx;
        if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w)
          hreg[u] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * p.w + ix) * p.cin + c * 8);
      }
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int idx = u * NT + tid;
      if (idx < halo_total)
        *reinterpret_cast<u32x4*>(halo + (idx / CPP) * PSTR + (idx % CPP) * 16) = hreg[u];
    }
  };

  int wb = 0;
  if (prefetch) halo_load(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();  // previous chunk's halo fully consumed
    // ---- stage the halo tile of this channel chunk ----
    if (prefetch) {
      halo_store();
    } else {
      const int total = halo_total;
      const half_t* xb = x + (size_t)img * p.h * p.w * p.cin + cc * CK;
      for (int base = 0; base < total; base += NT * 4) {
        u32x4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * NT + tid;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          off[u] = -1;
          if (idx < total) {
            int hp = idx / CPP, c = idx % CPP;
            int hy = hp / WT, hx = hp - hy * WT;
            int iy = iy0 + hy, ix = ix0 + hx;
            off[u] = hp * PSTR + c * 16;
            if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w)
              v[u] = *reinterpret_cast<const u32x4*>(
                  xb + ((size_t)iy * p.w + ix) * p.cin + c * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (off[u] >= 0) *reinterpret_cast<u32x4*>(halo + off[u]) = v[u];
      }
    }
    load_w(0, cc);
    store_w(wb);
    if (ntaps > 1) load_w(1, cc);
    if (prefetch && cc + 1 < nchunks) halo_load(cc + 1);
    for (int tap = 0; tap < ntaps; ++tap) {
      // after this barrier: buf[wb] (this tap, and at tap 0 the halo) is visible and
      // buf[wb^1] is free (its readers finished tap-1)
      __syncthreads();
      auto restage = [&]() {
        if (tap + 1 < ntaps) {
          store_w(wb ^ 1);
          if (tap + 2 < ntaps) load_w(tap + 2, cc);
        }
      };
      if constexpr (BN == 128) restage();
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const char* ab = wbuf + wb * (BN * PSTR) + a_lane;
      const char* bb = halo + b_lane + ((ky * p.dil) * WT + kx * p.dil) * PSTR +
                       (wpx * TPX) * b_row;
      if constexpr (BN == 128) {
      // software-pipelined k-steps: the LDS reads of step ks+1 are issued before the MFMAs of
      // step ks (two named fragment sets; sched_barrier keeps hipcc from sinking the reads)
      half8_t a0[TCO], b0[TPX], a1[TCO], b1[TPX];
#pragma unroll
      for (int i = 0; i < TCO; ++i) a0[i] = *reinterpret_cast<const half8_t*>(ab + i * 32 * PSTR);
#pragma unroll
      for (int t = 0; t < TPX; ++t) b0[t] = *reinterpret_cast<const half8_t*>(bb + t * b_row);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks += 2) {
        if (ks + 1 < KSTEPS) {
#pragma unroll
          for (int i = 0; i < TCO; ++i)
            a1[i] = *reinterpret_cast<const half8_t*>(ab + i * 32 * PSTR + (ks + 1) * 32);
#pragma unroll
          for (int t = 0; t < TPX; ++t)
            b1[t] = *reinterpret_cast<const half8_t*>(bb + t * b_row + (ks + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
          for (int t = 0; t < TPX; ++t)
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[i], b0[t], acc[i][t], 0, 0, 0);
        if (ks + 1 < KSTEPS) {
          if (ks + 2 < KSTEPS) {
#pragma unroll
            for (int i = 0; i < TCO; ++i)
              a0[i] = *reinterpret_cast<const half8_t*>(ab + i * 32 * PSTR + (ks + 2) * 32);
#pragma unroll
            for (int t = 0; t < TPX; ++t)
              b0[t] = *reinterpret_cast<const half8_t*>(bb + t * b_row + (ks + 2) * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int t = 0; t < TPX; ++t)
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b1[t], acc[i][t], 0, 0, 0);
        }
      }
      } else {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          half8_t a[TCO], b[TPX];
#pragma unroll
          for (int i = 0; i < TCO; ++i)
            a[i] = *reinterpret_cast<const half8_t*>(ab + i * 32 * PSTR + ks * 32);
#pragma unroll
          for (int t = 0; t < TPX; ++t)
            b[t] = *reinterpret_cast<const half8_t*>(bb + t * b_row + ks * 32);
          if (ks == 0) {
            // the next tap's weight slice goes to the other LDS buffer while the first
            // fragments are in flight (not between the barrier and the first reads)
            __builtin_amdgcn_sched_barrier(0);
            restage();
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int t = 0; t < TPX; ++t)
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[t], acc[i][t], 0, 0, 0);
        }
      }
      wb ^= 1;
    }
  }

  // ---- epilogue: accumulators -> LDS [256 px][<=128 couts] f16 -> coalesced rows ----
  if constexpr (BN <= 128) {
    __syncthreads();
    conv_epilogue<BN, TCO, TPX, WCO, NT>(acc, smem, p.flags, bias, y, stats, img, tyi, txi, mt, co0,
                                         p.oh, p.ow, p.cout, wco, wpx, true, p.br.y ? &p.br : nullptr);
  } else {
    // 256-wide tiles leave through LDS in two 128-cout halves (waves wco 0,1 then 2,3)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      conv_epilogue<128, TCO, TPX, 2, NT>(acc, smem, p.flags, bias, y, stats, img, tyi, txi, mt,
                                          co0 + h * 128, p.oh, p.ow, p.cout, wco & 1, wpx,
                                          (wco >> 1) == h, p.br.y ? &p.br : nullptr);
    }
  }
}

template <int BN, int CK, int WCO>
int launch(const ConvP& p, const void* x, const void* w, const void* bias, void* y,
           void* stats, hipStream_t st) {
  constexpr int PSTR = CK * 2 + 16;
  size_t main_bytes = (size_t)p.halo_bytes + 2 * BN * PSTR;
  size_t epi_bytes = conv_epilogue_lds(BN > 128 ? 128 : BN, 512);
  size_t lds = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  if (lds > 160 * 1024) return OCR_ERR_UNSUPPORTED;
  auto kern = conv_igemm_kernel<BN, CK, WCO>;
  static size_t configured = 0;
  if (lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(160 * 1024)) != hipSuccess)
      return OCR_ERR_HIP;
    configured = 160 * 1024;
  }
  const int m_tiles = p.n * p.tiles_x * p.tiles_y;
  dim3 grid((unsigned)(m_tiles * p.n_tiles));
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p,
                     static_cast<const half_t*>(x), static_cast<const half_t*>(w),
                     static_cast<const float*>(bias), static_cast<half_t*>(y),
                     static_cast<float*>(stats));
  return ocr_launch_status();
}

int fill_params(const ocr_conv_desc* d, ConvP* p, int* bn, int* ck) {
  OCR_CHECK_ARG(d != nullptr);
  OCR_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->oh > 0 && d->ow > 0);
  OCR_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0);
  OCR_CHECK_SHAPE(d->cin % 32 == 0 && d->cout % 32 == 0);
  p->n = d->n; p->h = d->h; p->w = d->w; p->cin = d->cin;
  p->oh = d->oh; p->ow = d->ow; p->cout = d->cout;
  p->kh = d->kh; p->kw = d->kw; p->stride = d->stride; p->dil = d->dilation;
  p->pt = d->pad_top; p->pl = d->pad_left; p->flip = d->flip_taps; p->flags = d->flags;
  p->tiles_x = ocr_cdiv(d->ow, TILE_W);
  p->tiles_y = ocr_cdiv(d->oh, TILE_H);
  p->HT = (TILE_H - 1) * d->stride + (d->kh - 1) * d->dilation + 1;
  p->WT = (TILE_W - 1) * d->stride + (d->kw - 1) * d->dilation + 1;
  *bn = (d->cout % 256 == 0) ? 256 : (d->cout % 128 == 0) ? 128 : (d->cout % 64 == 0) ? 64 : 32;
  if (const char* e = getenv("OCR_CONV_BN")) { int v = atoi(e); if (v == 128 && *bn == 256) *bn = 128; }
  int c = (d->cin % 64 == 0) ? 64 : 32;
  if (const char* e = getenv("OCR_CONV_CK")) { if (atoi(e) == 32) c = 32; }
  // keep halo + weight ring within the 160 KiB LDS
  auto need = [&](int ckk) {
    return (size_t)p->HT * p->WT * (ckk * 2 + 16) + 2 * (size_t)(*bn) * (ckk * 2 + 16);
  };
  if (c == 64 && need(64) > 160 * 1024) c = 32;
  if (need(c) > 160 * 1024) return OCR_ERR_UNSUPPORTED;
  *ck = c;
  p->halo_bytes = p->HT * p->WT * (c * 2 + 16);
  p->n_tiles = d->cout / *bn;
  p->br = BnRed{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return OCR_OK;
}

}  // namespace

extern "C" int ocr_conv2d_num_mtiles(const ocr_conv_desc* d) {
  if (!d) return OCR_ERR_INVALID_ARG;
  return d->n * ocr_cdiv(d->ow, TILE_W) * ocr_cdiv(d->oh, TILE_H);
}

static int dispatch(ConvP& p, int bn, int ck, const void* x, const void* w_kc, const void* bias, void* y,
                    void* stats, hipStream_t st) {
  if (bn == 256 && ck == 64) return launch<256, 64, 4>(p, x, w_kc, bias, y, stats, st);
  if (bn == 256 && ck == 32) return launch<256, 32, 4>(p, x, w_kc, bias, y, stats, st);
  if (bn == 128 && ck == 64) return launch<128, 64, 2>(p, x, w_kc, bias, y, stats, st);
  if (bn == 128 && ck == 32) return launch<128, 32, 2>(p, x, w_kc, bias, y, stats, st);
  if (bn == 64 && ck == 64) return launch<64, 64, 2>(p, x, w_kc, bias, y, stats, st);
  if (bn == 32 && ck == 64) return launch<32, 64, 1>(p, x, w_kc, bias, y, stats, st);
  if (bn == 32) return launch<32, 32, 1>(p, x, w_kc, bias, y, stats, st);
  return launch<64, 32, 2>(p, x, w_kc, bias, y, stats, st);
}

extern "C" int ocr_conv2d_f16(const ocr_conv_desc* d, const void* x, const void* w_kc,
                              const void* bias, void* y, void* stats, void* stream) {
  ConvP p;
  int bn = 0, ck = 0;
  int rc = fill_params(d, &p, &bn, &ck);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x && w_kc && y);
  OCR_CHECK_ARG(!(d->flags & OCR_CONV_BIAS) || bias);
  OCR_CHECK_ARG(!(d->flags & OCR_CONV_STATS) || stats);
  return dispatch(p, bn, ck, x, w_kc, bias, y, stats, static_cast<hipStream_t>(stream));
}

extern "C" int ocr_conv2d_bnred_f16(const ocr_conv_desc* d, const void* x, const void* w_kc, void* y,
                                    void* partial, const void* bn_y, const void* bn_scale,
                                    const void* bn_shift, const void* bn_mean, const void* bn_invstd,
                                    int bn_relu, void* stream) {
  ConvP p;
  int bn = 0, ck = 0;
  int rc = fill_params(d, &p, &bn, &ck);
  if (rc != OCR_OK) return rc;
  OCR_CHECK_ARG(x && w_kc && y && partial && bn_y && bn_scale && bn_shift && bn_mean && bn_invstd);
  OCR_CHECK_ARG(!(d->flags & (OCR_CONV_BIAS | OCR_CONV_RELU)));
  p.flags |= OCR_CONV_STATS;
  p.br = BnRed{static_cast<const half_t*>(bn_y), static_cast<const float*>(bn_scale),
               static_cast<const float*>(bn_shift), static_cast<const float*>(bn_mean),
               static_cast<const float*>(bn_invstd), bn_relu};
  return dispatch(p, bn, ck, x, w_kc, nullptr, y, partial, static_cast<hipStream_t>(stream));
}
