// Shared epilogue of the MFMA convolution kernels: 32x32 f32 accumulator tiles
// (rows = cout, cols = pixels of an 8x32 spatial tile) -> LDS [256 px][BN] f16
// -> 16-byte coalesced row stores, with optional bias / ReLU / accumulate and
// the per-tile per-cout sum / sum-of-squares needed by training-mode batch norm.
#pragma once
// Cache policy of the shared epilogue's output stores (-DOCR_EPILOGUE_NT=1: non-temporal; measurement switch)
#ifndef OCR_EPILOGUE_NT
#define OCR_EPILOGUE_NT 1          // A/B in one gpurun call: headline 18.26-18.30 -> 18.17-18.20 ms, PixelLink 15.81 -> 15.74
#endif
#if OCR_EPILOGUE_NT
#define OCR_EPILOGUE_STORE(ptr, val) __builtin_nontemporal_store(val, ptr)
#else
#define OCR_EPILOGUE_STORE(ptr, val) (*(ptr) = (val))
#endif
#include "common.h"

constexpr int TILE_H = 8;
constexpr int TILE_W = 32;

// Optional fusion of the batch-norm BACKWARD reduction of the layer that produced this conv's
// output position (used by the input-gradient form): instead of (sum v, sum v^2) the per-tile
// partials become (sum dz, sum dz*xhat) with dz = v * [relu(bn(y)) > 0], xhat = (y - mean)*invstd,
// y = that layer's stored conv output (same [n,oh,ow,cout] geometry as the tensor being written).
struct BnRed {
  const half_t* y;
  const float *scale, *shift, *mean, *invstd;
  int relu;
  // TAIL mode (ResNet bottleneck output, relu(shortcut + bn(y))): `mask` is that OUTPUT; the value this
  // convolution stores becomes dz = [mask > 0] * value — the gradient past the ReLU, replacing the separate
  // mask pass — and the sums are (sum dz, sum dz * xhat(y)).  Only kernels that end in conv_epilogue_store
  // implement it (ocr_conv2d_bnred_tail_f16 routes accordingly).
  const half_t* mask;
  // tail mode, optional: gradient of a SUBSAMPLED view of this tensor (ResNet's stride-2 identity shortcut,
  // subsample(x, 2) = x[:, ::2, ::2]), [n][ceil(sub_h/2)][ceil(sub_w/2)][cout]: added at the even positions
  // instead of being zero-inserted into a full-size tensor first.  sub_h x sub_w = this tensor's spatial size;
  // the flat pixel index is split by multiply-shift (m, l: q = (P * m) >> (31 + l), exact below 2^31).
  const half_t* sub;
  int sub_h, sub_w;
  unsigned long long sub_m_hw, sub_m_w;
  int sub_l_hw, sub_l_w;
  // tail mode, instead of `mask`: the ReLU mask as BITS, one byte per 8 channels (bit e = out[.. + e] > 0), written
  // by the kernel that produced the output (bn_add_relu_kernel / conv_pwx_kernel): 1/16 of the bytes of `out`
  const unsigned char* mask_bits;
  // plain (non-tail) mode with relu: STORE dz = value * [relu(bn(y)) > 0] instead of the value — for layers whose
  // activation is relu(conv + bias) (scale 1, shift 0: y IS the activation) there is no BN-backward pass that could
  // apply the mask later, and partial row 0 (sum dz) is their bias gradient (PixelLink's VGG, nets/pixellink.py:41-48)
  int store_dz;
  // conv1_1 as the producing layer (conv_c64_persist_kernel, epilogue mode 6): y is NOT read — nobody stored it — but
  // evaluated again per tile from the prepared image [n][oh][ow][4] and the first layer's packed weights [3][cout][16]
  // by the forward's own MFMA sequence (conv_first.hip: first_mfma), hence the same 16-bit values
  const half_t* first_x4;
  const half_t* first_wf;
  // ... epilogue mode 7: the gradient is NOT stored; S1 = V^T dz (V: the image patches, dz: the gradient past conv1_1's
  // ReLU) leaves as one [32 slots][64 channels] f32 block per workgroup — all conv1_1's weight gradient needs
  // (ocr_conv2d_bnred_first_wgrad_f16)
  float* first_s1;
};

// LDS needed by the epilogue for a BN-wide tile.
constexpr size_t conv_epilogue_lds(int bn, int nt = 256) {
  return 256 * (bn * 2 + 16) > nt * 16 * sizeof(float) ? 256 * (bn * 2 + 16) : nt * 16 * sizeof(float);   // staging tile | reduction scratch (aliased)
}

// Second half of the epilogue, shared by both accumulator layouts: LDS [256 px][BN] f16 -> HBM rows.
// BATCH: the variant for kernels launched WITH global operands in the epilogue (ACCUM / BN-backward / tail):
// it requests a batch's operands ahead of the batch's stores at the price of ~40 registers, which the
// memory-bound forward launches of the same kernels cannot afford (occupancy) — they keep the serial loop.
template <int BN, int NT, bool BATCH = false>
__device__ __forceinline__ void conv_epilogue_store(char* smem, int flags, half_t* __restrict__ y,
                                                    float* __restrict__ stats, int img, int tyi, int txi,
                                                    int mt, int co0, int oh, int ow, int cout,
                                                    const BnRed* br) {
  constexpr int OSTR = BN * 2 + 16;
  const int tid = threadIdx.x;
  char* otile = smem;
  // `red` ALIASES the output tile (a barrier separates the tile's last read from its first write): a 256-cout
  // tile fits the LDS in one piece (256 x (512 + 16) = 132 KB), a 128-cout one leaves room for a second workgroup
  float* red = reinterpret_cast<float*>(smem);
  __syncthreads();
  constexpr int NC = BN / 8;    // 16-byte chunks per output row
  constexpr int RG = NT / NC;   // row groups
  constexpr int PPT = 256 / RG; // pixels per thread
  const int c = tid % NC, rg = tid / NC;
  const bool accum = (flags & OCR_CONV_ACCUM_F16) != 0;
  const bool do_stats = (flags & OCR_CONV_STATS) != 0;
  float s[8], q2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q2[e] = 0.f; }
  // Pixels in batches of four: every global operand of the batch (the old gradient under ACCUM, the tail
  // mask, the BN-backward operand) is requested BEFORE the batch's first store — the stores may alias those
  // tensors as far as the compiler knows, and a load placed behind one waits for it (measured on the
  // 1x1 input-gradient kernels of ResNet-50: three dependent loads per pixel made the fused tail 2x slower
  // than the separate element-wise passes it replaces).
  const bool tail = br != nullptr && (br->mask != nullptr || br->mask_bits != nullptr);
  const bool tbits = tail && br->mask_bits != nullptr;
  // this thread's 8 couts are fixed: the ReLU-mask coefficients once, up front; mean / invstd enter the sums
  // linearly and are applied at the end (sum dz*xhat = invstd * (sum dz*y - mean * sum dz))
  float bsc[8], bsh[8];
  const bool remask = do_stats && br != nullptr && !tail && br->relu;
  if (remask) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { bsc[e] = br->scale[co0 + c * 8 + e]; bsh[e] = br->shift[co0 + c * 8 + e]; }
  }
  if constexpr (BATCH) {
    // ALL of the thread's pixels in one batch (up to 8 x 3 operands of 16 bytes in flight per thread): one workgroup
    // is resident per CU and each batch exposes one HBM latency — with batches of four the expanding 1x1
    // input-gradient convolutions of ResNet (64 -> 256 at 160^2) ran at 3 TB/s in tail mode, 1.8 with the
    // BN-backward operand alone
    constexpr int UB = PPT < 8 ? PPT : 8;
    static_assert(PPT % UB == 0, "pixels per thread");
    for (int k0 = 0; k0 < PPT; k0 += UB) {
      half8_t old[UB], mk[UB], yv[UB];
      unsigned mb[UB];
      auto pixel = [&](int u, bool& ok) __attribute__((always_inline)) {     // (recomputed, not carried: registers)
        const int px = rg + (k0 + u) * RG;
        const int oy = tyi * TILE_H + (px >> 5), ox = txi * TILE_W + (px & 31);
        ok = oy < oh && ox < ow;
        return (((size_t)img * oh + oy) * ow + ox) * cout + co0 + c * 8;
      };
      // BRANCH-FREE requests: an operand this launch does not have is read from a dummy row (element 0 of the output /
      // the partials) and ignored.  With `if (accum) ... if (tail) ... if (stats) ...` around the loads hipcc emitted a
      // scalar branch per load and — unable to count the outstanding loads across the paths — `s_waitcnt vmcnt(0)` at
      // every merge: the "batch" went out one load at a time, each behind the previous one's round trip.
      const bool has_y = do_stats && br != nullptr && br->y != nullptr;
      const half_t* dummy = y != nullptr ? y : reinterpret_cast<const half_t*>(stats);
      const half_t* p_old = accum ? y : dummy;
      const half_t* p_mk = (tail && !tbits) ? br->mask : dummy;
      const unsigned char* p_mb = tbits ? br->mask_bits : reinterpret_cast<const unsigned char*>(dummy);
      const half_t* p_yv = has_y ? br->y : dummy;
      const size_t m_old = accum ? ~(size_t)0 : 0, m_mk = (tail && !tbits) ? ~(size_t)0 : 0, m_mb = tbits ? ~(size_t)0 : 0,
                   m_yv = has_y ? ~(size_t)0 : 0;
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        bool ok;
        size_t off = pixel(u, ok);
        off = ok ? off : 0;
        old[u] = *reinterpret_cast<const half8_t*>(p_old + (off & m_old));
        mk[u] = *reinterpret_cast<const half8_t*>(p_mk + (off & m_mk));
        mb[u] = p_mb[(off >> 3) & m_mb];
        yv[u] = *reinterpret_cast<const half8_t*>(p_yv + (off & m_yv));
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        bool ok;
        const size_t off = pixel(u, ok);
        if (!ok) continue;
        half8_t w = *reinterpret_cast<const half8_t*>(otile + (rg + (k0 + u) * RG) * OSTR + c * 16);
        if (accum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (half_t)((float)w[e] + (float)old[u][e]);
        }
        if (tail && br->sub != nullptr) {            // (three launches per ResNet-50 step: loaded in place)
          const unsigned P = (unsigned)(off / (size_t)cout);                    // flat pixel index
          const unsigned im = (unsigned)(((unsigned long long)P * br->sub_m_hw) >> (31 + br->sub_l_hw));
          const unsigned rem = P - im * (unsigned)(br->sub_h * br->sub_w);
          const unsigned yy = (unsigned)(((unsigned long long)rem * br->sub_m_w) >> (31 + br->sub_l_w));
          const unsigned xx = rem - yy * (unsigned)br->sub_w;
          if (((yy | xx) & 1u) == 0u) {
            const int sh = (br->sub_h + 1) >> 1, sw = (br->sub_w + 1) >> 1;
            const half8_t sv = *reinterpret_cast<const half8_t*>(
                br->sub + (((size_t)im * sh + (yy >> 1)) * sw + (xx >> 1)) * cout + co0 + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (half_t)((float)w[e] + (float)sv[e]);
          }
        }
        if (tbits) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (mb[u] >> e & 1u) ? w[e] : (half_t)0.f;
        } else if (tail) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (float)mk[u][e] > 0.f ? w[e] : (half_t)0.f;
        } else if (remask && br->store_dz) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!((float)yv[u][e] * bsc[e] + bsh[e] > OCR_RELU_TIE)) w[e] = (half_t)0.f;
        }
        if (y != nullptr) OCR_EPILOGUE_STORE(reinterpret_cast<half8_t*>(y + off), w);   // (null: a statistics-only launch)
        if (do_stats) {
          if (br != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float yf = (float)yv[u][e];
              bool pass = true;                      // tail mode: w is dz already
              if (remask) pass = yf * bsc[e] + bsh[e] > OCR_RELU_TIE;   // mask of the stored activation
              const float dz = pass ? (float)w[e] : 0.f;
              s[e] += dz;
              q2[e] += dz * yf;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float f = (float)w[e];
              s[e] += f;
              q2[e] += f * f;
            }
          }
        }
      }
    }
  } else {
#pragma unroll 4
    for (int k = 0; k < PPT; ++k) {
      const int px = rg + k * RG;
      const int oy = tyi * TILE_H + (px >> 5), ox = txi * TILE_W + (px & 31);
      if (oy < oh && ox < ow) {
        half8_t w = *reinterpret_cast<const half8_t*>(otile + px * OSTR + c * 16);
        const size_t off = (((size_t)img * oh + oy) * ow + ox) * cout + co0 + c * 8;
        if (accum) {
          const half8_t old = *reinterpret_cast<const half8_t*>(y + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (half_t)((float)w[e] + (float)old[e]);
        }
        if (tail && br->sub != nullptr) {
          const unsigned P = (unsigned)(((size_t)img * oh + oy) * ow + ox);      // flat pixel index
          const unsigned im = (unsigned)(((unsigned long long)P * br->sub_m_hw) >> (31 + br->sub_l_hw));
          const unsigned rem = P - im * (unsigned)(br->sub_h * br->sub_w);
          const unsigned yy = (unsigned)(((unsigned long long)rem * br->sub_m_w) >> (31 + br->sub_l_w));
          const unsigned xx = rem - yy * (unsigned)br->sub_w;
          if (((yy | xx) & 1u) == 0u) {
            const int sh = (br->sub_h + 1) >> 1, sw = (br->sub_w + 1) >> 1;
            const half8_t sv = *reinterpret_cast<const half8_t*>(
                br->sub + (((size_t)im * sh + (yy >> 1)) * sw + (xx >> 1)) * cout + co0 + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (half_t)((float)w[e] + (float)sv[e]);
          }
        }
        if (tbits) {
          const unsigned mb = br->mask_bits[off >> 3];
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (mb >> e & 1u) ? w[e] : (half_t)0.f;
        } else if (tail) {
          const half8_t mk = *reinterpret_cast<const half8_t*>(br->mask + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = (float)mk[e] > 0.f ? w[e] : (half_t)0.f;
        } else if (remask && br->store_dz) {
          const half8_t yv0 = *reinterpret_cast<const half8_t*>(br->y + off);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (!((float)yv0[e] * bsc[e] + bsh[e] > OCR_RELU_TIE)) w[e] = (half_t)0.f;
        }
        if (y != nullptr) OCR_EPILOGUE_STORE(reinterpret_cast<half8_t*>(y + off), w);   // (null: a statistics-only launch)
        if (do_stats) {
          if (br != nullptr) {
            const half8_t yv = *reinterpret_cast<const half8_t*>(br->y + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float yf = (float)yv[e];
              bool pass = true;                    // tail mode: w is dz already
              if (remask) pass = yf * bsc[e] + bsh[e] > OCR_RELU_TIE;   // mask of the stored activation
              const float dz = pass ? (float)w[e] : 0.f;
              s[e] += dz;
              q2[e] += dz * yf;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float f = (float)w[e];
              s[e] += f;
              q2[e] += f * f;
            }
          }
        }
      }
    }
  }
  if (do_stats) {
    if (br != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        q2[e] = (q2[e] - br->mean[co0 + c * 8 + e] * s[e]) * br->invstd[co0 + c * 8 + e];
    }
    __syncthreads();                                 // every thread is done reading the output tile `red` aliases
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(rg * NC + c) * 16 + e] = s[e];
      red[(rg * NC + c) * 16 + 8 + e] = q2[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int cc2 = tid >> 4, e = tid & 15;
      float tot = 0.f;
      for (int g = 0; g < RG; ++g) tot += red[(g * NC + cc2) * 16 + e];
      stats[((size_t)mt * 2 + (e >> 3)) * cout + co0 + cc2 * 8 + (e & 7)] = tot;
    }
  }
}

template <int BN, int TCO, int TPX, int WCO, int NT = 256>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[TCO][TPX], char* smem, int flags,
                                              const float* __restrict__ bias,
                                              half_t* __restrict__ y, float* __restrict__ stats,
                                              int img, int tyi, int txi, int mt, int co0, int oh,
                                              int ow, int cout, int wco, int wpx, bool active,
                                              const BnRed* br = nullptr) {
  constexpr int OSTR = BN * 2 + 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int r = lane & 31, hh = lane >> 5;
  char* otile = smem;
  const bool has_bias = (flags & OCR_CONV_BIAS) != 0;
  const bool relu = (flags & OCR_CONV_RELU) != 0;
  if (active) {
  // the wave's bias values, all requested before the first is used (16-byte loads from a dummy row when there is no
  // bias): loaded group by group under `if (has_bias)` each group waited out its own round trip
  f32x4 bvv[TCO][4];
  {
    const float* bp = has_bias ? bias + co0 : reinterpret_cast<const float*>(smem);
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = wco * TCO * 32 + i * 32 + q * 8 + hh * 4;
        if (has_bias) bvv[i][q] = *reinterpret_cast<const f32x4*>(bp + col);
        else bvv[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
#pragma unroll
  for (int i = 0; i < TCO; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = wco * TCO * 32 + i * 32 + q * 8 + hh * 4;
      float bv[4] = {bvv[i][q][0], bvv[i][q][1], bvv[i][q][2], bvv[i][q][3]};
#pragma unroll
      for (int t = 0; t < TPX; ++t) {
        const int px = (wpx * TPX + t) * 32 + r;
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][t][q * 4 + e] + bv[e];
          if (relu) v = v > 0.f ? v : 0.f;
          o[e] = (half_t)v;
        }
        *reinterpret_cast<half4_t*>(otile + px * OSTR + col * 2) = o;
      }
    }
  }
  }
  conv_epilogue_store<BN, NT>(smem, flags, y, stats, img, tyi, txi, mt, co0, oh, ow, cout, br);
}

// Same epilogue for 16x16 accumulator tiles (v_mfma_f32_16x16x32_f16): lane l holds couts
// 4*(l>>4)..+3 of pixel l&15 of each tile; acc[i][t] covers couts i*16.., pixels t*16.. of the wave.
template <int BN, int TCO, int TPX, int WCO, int NT = 256, bool BATCH = false>
__device__ __forceinline__ void conv_epilogue16(f32x4 (&acc)[TCO * 2][TPX * 2], char* smem, int flags,
                                                const float* __restrict__ bias,
                                                half_t* __restrict__ y, float* __restrict__ stats,
                                                int img, int tyi, int txi, int mt, int co0, int oh,
                                                int ow, int cout, int wco, int wpx, bool active,
                                                const BnRed* br = nullptr) {
  constexpr int OSTR = BN * 2 + 16;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g4 = lane >> 4;
  char* otile = smem;
  const bool has_bias = (flags & OCR_CONV_BIAS) != 0;
  const bool relu = (flags & OCR_CONV_RELU) != 0;
  if (active) {
    f32x4 bvv[TCO * 2];                           // (all of the wave's bias values requested up front: see conv_epilogue)
#pragma unroll
    for (int i = 0; i < TCO * 2; ++i) {
      const int col = wco * TCO * 32 + i * 16 + g4 * 4;
      if (has_bias) bvv[i] = *reinterpret_cast<const f32x4*>(bias + co0 + col);
      else bvv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TCO * 2; ++i) {
      const int col = wco * TCO * 32 + i * 16 + g4 * 4;
      float bv[4] = {bvv[i][0], bvv[i][1], bvv[i][2], bvv[i][3]};
#pragma unroll
      for (int t = 0; t < TPX * 2; ++t) {
        const int px = wpx * TPX * 32 + t * 16 + r;
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][t][e] + bv[e];
          if (relu) v = v > 0.f ? v : 0.f;
          o[e] = (half_t)v;
        }
        *reinterpret_cast<half4_t*>(otile + px * OSTR + col * 2) = o;
      }
    }
  }
  conv_epilogue_store<BN, NT, BATCH>(smem, flags, y, stats, img, tyi, txi, mt, co0, oh, ow, cout, br);
}
