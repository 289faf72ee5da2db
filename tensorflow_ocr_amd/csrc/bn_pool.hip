// HBM-bound streaming kernels around the convolutions: image preparation,
// training-mode batch norm (statistics finalisation, fused normalise+ReLU+2x2
// max-pool, fused backward), and the general k x k max-pool (pool5, ResNet pool1,
// subsample).  All f16 tensors are NHWC and are accessed 16 bytes (8 channels)
// per lane.
//
// Reference semantics: slim.batch_norm decay 0.997 / eps 1e-5 / scale=True
// (nets/resnet_utils.py:232-246, nets/model_vgg_16.py:144), slim.max_pool2d SAME
// (nets/vgg.py:16-32), mean_image_subtraction (nets/model.py:18-31).
#include "common.h"
#include "pool_gather.h"

namespace {

// ---------------------------------------------------------------- image prep
// (x - mean) / div, IEEE division (hipcc's default is the correctly rounded one), so the f16 values
// equal those of a host-side `(images - 120) / 60` followed by the cast; div = 1 is exact.
__global__ void prep_images_kernel(const float* __restrict__ img, half_t* __restrict__ out,
                                   size_t npix, float m0, float m1, float m2, float div) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float r = (img[i * 3 + 0] - m0) / div, g = (img[i * 3 + 1] - m1) / div, b = (img[i * 3 + 2] - m2) / div;
  half4_t o = {(half_t)r, (half_t)g, (half_t)b, (half_t)0.f};
  *reinterpret_cast<half4_t*>(out + i * 4) = o;
}

// ------------------------------------------------- per-channel partial reduce
// partial [T][2][C] f32 -> stage [R][2][C] f64 (R = ceil(T/256)) -> final per-channel sums, in ONE
// launch: every block reduces its 256 rows, publishes them and takes a ticket; the block that draws
// the last ticket of its 64-channel group sums the R stage rows IN ROW ORDER (so the result does not
// depend on which block happens to be last: bitwise reproducible) and runs the finalisation.  No
// block ever waits for another one (no spinning), the ticket counter resets itself.
struct BnFin {            // MODE 0: batch-norm forward statistics -> scale/shift (+ moving stats)
  double count;
  const float *gamma, *beta;
  float eps, decay;
  float *moving_mean, *moving_var, *scale, *shift, *save_mean, *save_invstd;
};
struct BnBwdFin {         // MODE 1: batch-norm backward sums -> dbeta, dgamma
  float *dgamma, *dbeta;
};
struct BnBwdFinC {        // MODE 2: ... and the coefficients of dy = A*dz + B*y + C (the apply step as an affine map of
  float *dgamma, *dbeta;  //         (dz, y), for a consumer that applies it while loading: conv_pwx_kernel)
  const float *scale, *mean, *invstd;
  float inv_count;
  float *A, *B, *C;
};

// Arrival counters, zero at load, self-resetting.  Two levels: blocks take a ticket of their group of 32
// row blocks, the last of a group takes a ticket of the channel group — R same-address atomics in a
// row cost ~0.1 us each (63 us for the 512 row blocks of conv1_2), 32 + R/32 do not.
constexpr int kTicketGroup = 32, kTicketGroups = 128;            // R <= 4096 row blocks
__device__ unsigned ocr_bn_tickets[16 * 32 * (1 + kTicketGroups)];   // [slot][channel group][0: level 2 | 1 + g: level 1]

template <typename FIN>
__device__ __forceinline__ void bn_fin_apply(const FIN& f, int c, double s, double q);

template <>
__device__ __forceinline__ void bn_fin_apply<BnFin>(const BnFin& f, int c, double s, double q) {
  double mean = s / f.count;
  double var = q / f.count - mean * mean;
  if (var < 0.0) var = 0.0;
  float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
  float g = f.gamma ? f.gamma[c] : 1.f;
  float b = f.beta ? f.beta[c] : 0.f;
  float sc = g * invstd;
  f.scale[c] = sc;
  f.shift[c] = b - (float)mean * sc;
  if (f.save_mean) f.save_mean[c] = (float)mean;
  if (f.save_invstd) f.save_invstd[c] = invstd;
  if (f.moving_mean) {
    // fused batch norm feeds the unbiased variance to the moving average
    double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
    f.moving_mean[c] = f.moving_mean[c] * f.decay + (float)mean * (1.f - f.decay);
    f.moving_var[c] = f.moving_var[c] * f.decay + (float)unbiased * (1.f - f.decay);
  }
}

template <>
__device__ __forceinline__ void bn_fin_apply<BnBwdFin>(const BnBwdFin& f, int c, double s, double q) {
  f.dbeta[c] = (float)s;
  f.dgamma[c] = (float)q;
}

template <>
__device__ __forceinline__ void bn_fin_apply<BnBwdFinC>(const BnBwdFinC& f, int c, double s, double q) {
  f.dbeta[c] = (float)s;
  f.dgamma[c] = (float)q;
  // bn_relu_bwd_kernel<1>: dy = sc * (dz - k_dz - (y - mu) * is * k_dzx),  k_dz = dbeta / N, k_dzx = dgamma / N
  const float sc = f.scale[c], mu = f.mean[c], is = f.invstd[c];
  const float k_dz = (float)s * f.inv_count, k_dzx = (float)q * f.inv_count;
  f.A[c] = sc;
  f.B[c] = -sc * is * k_dzx;
  f.C[c] = sc * (mu * is * k_dzx - k_dz);
}

// Partial rows per block.  The launch is a latency chain: first phase (rows / 64 batches of 8 loads per thread),
// tickets, then the last arriver of a channel group sums the R = T / rows stage rows (R / 16 batches of
// agent-scope loads) — and every block costs ~0.1 us of dispatch.  Measured (MI355X, us per launch, rows =
// 64 | 256 | 512): T x C = 6400 x 256: 38.6 | 14.7 | 13.1; 8192 x 128: 28.6 | 11.9 | 11.4; 2048 x 256: 14.9 | 10.1 |
// 10.7; 400 x 1024: 13.9 | 9.9 | 9.0 (one block per channel group: 8.7).  So: up to 1024 rows one block per channel
// group finalises directly; beyond that T / 16 rows per block, at least 256.
static inline int red_rows(int T) {
  if (T <= 1024) return (T + 63) / 64 * 64;
  int rows = ((T + 15) / 16 + 31) / 32 * 32;
  if (rows < 256) rows = 256;
  if (rows > 2048) rows = 2048;
  return rows;
}

template <typename FIN>
__global__ __launch_bounds__(256) void reduce_finalize_kernel(const float* __restrict__ partial,
                                                              double* __restrict__ stage, int T, int C, int slot,
                                                              int rows, FIN fin) {
  // a block = one 64-channel group x `rows` partial rows; thread = 4 channels (one 16-byte load per
  // row and sum) x one of 16 row lanes
  __shared__ double red[16][2][64];
  __shared__ unsigned s_ticket;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c4 = blockIdx.y * 64 + cl * 4;        // first of this thread's 4 channels
  const int t0 = blockIdx.x * rows;
  const int R = gridDim.x;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  const bool live = c4 < C;
  const bool vec = (C & 3) == 0;
  if (live) {
    const int t1 = min(t0 + rows, T);
    for (int t = t0 + rl; t < t1; t += 64) {      // rows rl, rl+16, ...: four rows (8 loads) in flight, added in row order
      f32x4 a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int tt = t + 16 * k;
        const bool ok = tt < t1;
        a[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        b[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float* pa = partial + ((size_t)tt * 2 + 0) * C + c4;
          const float* pb = partial + ((size_t)tt * 2 + 1) * C + c4;
          if (vec) {
            a[k] = *reinterpret_cast<const f32x4*>(pa);
            b[k] = *reinterpret_cast<const f32x4*>(pb);
          } else {                                 // C not a multiple of 4 (the 18-channel heads): scalar, bounded
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c4 + e < C) { a[k][e] = pa[e]; b[k][e] = pb[e]; }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[e] += (double)a[k][e];
          q[e] += (double)b[k][e];
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[rl][0][cl * 4 + e] = s[e];
    red[rl][1][cl * 4 + e] = q[e];
  }
  __syncthreads();
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int which = threadIdx.x >> 6;             // threads 0..63: sums, 64..127: second sums
  double tot = 0.0;
  if (which < 2) {
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k][which][threadIdx.x & 63];
    if (R > 1 && c < C) stage[((size_t)blockIdx.x * 2 + which) * C + c] = tot;
  }
  if (R == 1) {                                   // single block per channel group: finalise directly
    if (which == 1) red[0][1][threadIdx.x & 63] = tot;
    __syncthreads();
    if (which == 0 && c < C) bn_fin_apply(fin, c, tot, red[0][1][threadIdx.x & 63]);
    return;
  }
  __threadfence();                                // publish this block's stage rows
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* tk = &ocr_bn_tickets[(size_t)(slot * 32 + blockIdx.y) * (1 + kTicketGroups)];
    const int g = blockIdx.x / kTicketGroup, ng = (R + kTicketGroup - 1) / kTicketGroup;
    const int gsize = min(kTicketGroup, R - g * kTicketGroup);
    unsigned last = 0u;
    if (atomicAdd(tk + 1 + g, 1u) == (unsigned)(gsize - 1)) {     // last of its group: everyone else of the group is done
      tk[1 + g] = 0u;                                             // ready for the next launch that uses this slot
      __threadfence();
      if (atomicAdd(tk, 1u) == (unsigned)(ng - 1)) {
        tk[0] = 0u;
        last = 1u;
      }
    }
    s_ticket = last;
  }
  __syncthreads();
  if (!s_ticket) return;                          // not the last block of this channel group
  __threadfence();
  // the last arriver sums the R stage rows: thread = (sum kind, channel) x one of two row lanes... keep
  // it simple and wide: 128 (kind, channel) columns x 2 row lanes, eight rows in flight, fixed order
  {
    const int col = threadIdx.x & 127, lane2 = threadIdx.x >> 7;      // col: kind = col >> 6, channel = col & 63
    const int cc = blockIdx.y * 64 + (col & 63);
    double acc = 0.0;
    if (cc < C) {
      const unsigned long long* st = reinterpret_cast<const unsigned long long*>(stage);
      for (int r0 = lane2; r0 < R; r0 += 16) {   // agent-scope loads (other CUs wrote these)
        unsigned long long u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = r0 + 2 * k;
          u[k] = r < R ? __hip_atomic_load(st + ((size_t)r * 2 + (col >> 6)) * C + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += __builtin_bit_cast(double, u[k]);
      }
    }
    red[lane2][col >> 6][col & 63] = acc;         // (every reader of the first use passed the barriers above)
  }
  __syncthreads();
  if (threadIdx.x < 64 && c < C)
    bn_fin_apply(fin, c, red[0][0][threadIdx.x] + red[1][0][threadIdx.x], red[0][1][threadIdx.x] + red[1][1][threadIdx.x]);
}

// Up to eight SMALL reductions (T <= 2048 partial rows, C <= 32 channels: the fuse heads' batch norms) finalised in one
// launch, one 1024-thread block each: thread = (kind, channel) column x one of 16 row lanes, eight rows in flight, rows
// added in row order per lane, the 16 lanes in lane order — bitwise reproducible, no tickets.
struct BnBatchItem {
  const float* partial;
  int T, C;
  int mode;                 // 0: BnFin (forward statistics), 1: BnBwdFin (dbeta, dgamma)
  BnFin fin;
  BnBwdFin bwd;
};
struct BnBatchTab {
  BnBatchItem it[8];
};
__global__ __launch_bounds__(1024) void bn_finalize_batch_kernel(BnBatchTab tab) {
  __shared__ double red[1024];
  const BnBatchItem& it = tab.it[blockIdx.x];
  // thread = one of the 2 C (kind, channel) columns x one of RL = 1024 / (2 C) row lanes (28 lanes for the 18-channel
  // heads): a lane adds rows lane, lane + RL, ... in row order, eight loads in flight; the lanes are combined in lane order
  const int cols = 2 * it.C;
  const int RL = 1024 / cols;
  const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
  double acc = 0.0;
  if (rl < RL) {
    const float* base = it.partial + col;            // row t: partial[t * 2C + kind * C + c] = partial[t * cols + col]
    for (int t0 = rl; t0 < it.T; t0 += RL * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {            // (branch-free: a row past the end re-reads row 0 and counts as zero)
        const int t = t0 + RL * k;
        const float x = base[(size_t)(t < it.T ? t : 0) * cols];
        v[k] = t < it.T ? x : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += (double)v[k];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < it.C) {
    const int c = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < RL; ++k) {
      s += red[k * cols + c];
      q += red[k * cols + it.C + c];
    }
    if (it.mode == 0) bn_fin_apply<BnFin>(it.fin, c, s, q);
    else bn_fin_apply<BnBwdFin>(it.bwd, c, s, q);
  }
}

// launches on one stream are ordered; the rotating slot keeps launches that might overlap on
// DIFFERENT streams (side-stream experiments) off each other's counters
static int bn_ticket_slot() {
  static int next = 0;
  next = (next + 1) & 15;
  return next;
}

__global__ void bn_inference_params_kernel(const float* gamma, const float* beta, const float* mm,
                                           const float* mv, float eps, int C, float* scale,
                                           float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.f / sqrtf(mv[c] + eps);
  float sc = (gamma ? gamma[c] : 1.f) * invstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - mm[c] * sc;
}

// ------------------------------------------------ fused normalise+ReLU(+pool)
// pool == 0: one unit = one pixel.  pool == 2: one unit = one 2x2 window
// (SAME, stride 2: windows hanging over an odd edge ignore the missing pixels).
template <bool RELU>
__device__ __forceinline__ void bn_act8(const half8_t& v, const float* sc, const float* sh, float* out) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float f = (float)v[e] * sc[e] + sh[e];
    out[e] = RELU ? (f > 0.f ? f : 0.f) : f;
  }
}

// The forward pass reads y ONCE (the next reader is the backward pass, a whole step later): a non-temporal load keeps it out
// of the way of the activation the pass writes, which the next convolution reads at once.  -DOCR_BN_NT=0: plain loads (A/B).
#ifndef OCR_BN_NT
#define OCR_BN_NT 1
#endif
#if OCR_BN_NT
#define OCR_BN_LOAD_Y(p) __builtin_nontemporal_load(p)
#else
#define OCR_BN_LOAD_Y(p) (*(p))
#endif
// (Non-temporal STORES of the activation were measured too, always and above 200 MiB: 18.71-18.74 / 18.66-18.73 against
// 18.68-18.69 ms per step — nothing; the next convolution reads it at once.)
template <bool RELU, int POOL>
__global__ __launch_bounds__(256) void bn_relu_kernel(const half_t* __restrict__ y,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int n, int h,
                                                      int w, int c, half_t* __restrict__ a_full,
                                                      half_t* __restrict__ a_pool,
                                                      unsigned char* __restrict__ argmax = nullptr,
                                                      half_t* __restrict__ y_pool = nullptr) {
  const int chunks = c >> 3;
  const int oh = POOL ? (h + 1) / 2 : h, ow = POOL ? (w + 1) / 2 : w;
  const size_t units = (size_t)n * oh * ow;
  const size_t total = units * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    const size_t u = i / chunks;
    float sc[8], sh[8];
    const f32x4* scp = reinterpret_cast<const f32x4*>(scale + ch * 8);
    const f32x4* shp = reinterpret_cast<const f32x4*>(shift + ch * 8);
    f32x4 s0 = scp[0], s1 = scp[1], h0 = shp[0], h1 = shp[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = h0[e]; sh[4 + e] = h1[e]; }
    if (POOL == 0) {
      half8_t v = OCR_BN_LOAD_Y(reinterpret_cast<const half8_t*>(y + u * c + ch * 8));
      float f[8];
      bn_act8<RELU>(v, sc, sh, f);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)f[e];
      *reinterpret_cast<half8_t*>(a_full + u * c + ch * 8) = o;
    } else {
      const int ox = (int)(u % ow);
      const size_t t = u / ow;
      const int oy = (int)(t % oh);
      const int img = (int)(t / oh);
      float m[8];
      half8_t ym;                                   // the conv output AT the first maximum (y_pool)
      unsigned long long am = 0ull;                 // first-max candidate (dy*2+dx) of each channel, one byte each
#pragma unroll
      for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; ym[e] = (half_t)0.f; }
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int iy = oy * 2 + dy, ix = ox * 2 + dx;
          if (iy < h && ix < w) {
            const size_t off = (((size_t)img * h + iy) * w + ix) * c + ch * 8;
            half8_t v = OCR_BN_LOAD_Y(reinterpret_cast<const half8_t*>(y + off));
            float f[8];
            bn_act8<RELU>(v, sc, sh, f);
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              o[e] = (half_t)f[e];
              float fr = (float)o[e];
              if (fr > m[e]) {
                m[e] = fr;
                ym[e] = v[e];
                am = (am & ~(0xffull << (8 * e))) | ((unsigned long long)(dy * 2 + dx) << (8 * e));
              }
            }
            if (a_full) *reinterpret_cast<half8_t*>(a_full + off) = o;
          }
        }
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)m[e];
      *reinterpret_cast<half8_t*>(a_pool + u * c + ch * 8) = o;
      if (y_pool) *reinterpret_cast<half8_t*>(y_pool + u * c + ch * 8) = ym;
      if (argmax) {
        // bit 2 of each byte: the pooled activation is positive (the backward's ReLU mask, so that it
        // need not read a_pool)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (m[e] > 0.f) am |= 4ull << (8 * e);
        *reinterpret_cast<unsigned long long*>(argmax + u * c + ch * 8) = am;
      }
    }
  }
}

// ------------------------------------------------------------ fused backward
// dz = (da_full [+ routed da_pool]) * [z > 0];   partial sums of dz and dz*xhat.
// MODE 0: reduce (writes partial[blk][2][C]);  MODE 1: apply (writes dy f16).
struct BnBwdP {
  int n, h, w, c, relu, pool;
  float inv_count;
};

template <int MODE>
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(
    BnBwdP p, const half_t* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ dgamma,
    const float* __restrict__ dbeta, const half_t* __restrict__ da_full,
    const half_t* __restrict__ da_pool, float* __restrict__ partial, half_t* __restrict__ dy) {
  __shared__ float red[256 * 16];
  const int c = p.c, chunks = c >> 3;
  const int lanes = 256 / chunks;  // unit lanes per block
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const int oh = p.pool ? (p.h + 1) / 2 : p.h, ow = p.pool ? (p.w + 1) / 2 : p.w;
  const size_t units = (size_t)p.n * oh * ow;

  float sc[8], sh[8], mu[8], is[8], k_dz[8], k_dzx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cc = ch * 8 + e;
    sc[e] = scale[cc];
    sh[e] = shift[cc];
    mu[e] = mean[cc];
    is[e] = invstd[cc];
    if (MODE == 1) {
      k_dz[e] = dbeta[cc] * p.inv_count;
      k_dzx[e] = dgamma[cc] * p.inv_count;
    }
  }
  float s_dz[8], s_dzx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s_dz[e] = 0.f; s_dzx[e] = 0.f; }

  for (size_t u = (size_t)blockIdx.x * lanes + ul; u < units; u += (size_t)gridDim.x * lanes) {
    if (!p.pool) {
      const size_t off = u * c + ch * 8;
      half8_t v = *reinterpret_cast<const half8_t*>(y + off);
      half8_t g = *reinterpret_cast<const half8_t*>(da_full + off);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float yv = (float)v[e];
        float z = (float)(half_t)(yv * sc[e] + sh[e]);  // the stored f16 activation
        float dz = (!p.relu || z > 0.f) ? (float)g[e] : 0.f;
        float xh = (yv - mu[e]) * is[e];
        if (MODE == 0) {
          s_dz[e] += dz;
          s_dzx[e] += dz * xh;
        } else {
          o[e] = (half_t)(sc[e] * (dz - k_dz[e] - xh * k_dzx[e]));
        }
      }
      if (MODE == 1) *reinterpret_cast<half8_t*>(dy + off) = o;
    } else {
      const int ox = (int)(u % ow);
      const size_t t = u / ow;
      const int oy = (int)(t % oh);
      const int img = (int)(t / oh);
      half8_t gp = *reinterpret_cast<const half8_t*>(da_pool + u * c + ch * 8);
      half8_t v[4];
      float a[4][8];
      bool valid[4];
      size_t offs[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int iy = oy * 2 + (k >> 1), ix = ox * 2 + (k & 1);
        valid[k] = iy < p.h && ix < p.w;
        offs[k] = (((size_t)img * p.h + (valid[k] ? iy : 0)) * p.w + (valid[k] ? ix : 0)) * c + ch * 8;
        if (valid[k]) v[k] = *reinterpret_cast<const half8_t*>(y + offs[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = -INFINITY;
          if (valid[k]) {
            z = (float)v[k][e] * sc[e] + sh[e];
            if (p.relu && z < 0.f) z = 0.f;
            z = (float)(half_t)z;  // the stored f16 activation
          }
          a[k][e] = z;
        }
      }
      int arg[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int best = 0;
        float bv = a[0][e];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (a[k][e] > bv) { bv = a[k][e]; best = k; }
        arg[e] = best;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!valid[k]) continue;
        half8_t gf;
        if (da_full) gf = *reinterpret_cast<const half8_t*>(da_full + offs[k]);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float yv = (float)v[k][e];
          float g = (arg[e] == k) ? (float)gp[e] : 0.f;
          if (da_full) g += (float)gf[e];
          float dz = (!p.relu || a[k][e] > 0.f) ? g : 0.f;
          float xh = (yv - mu[e]) * is[e];
          if (MODE == 0) {
            s_dz[e] += dz;
            s_dzx[e] += dz * xh;
          } else {
            o[e] = (half_t)(sc[e] * (dz - k_dz[e] - xh * k_dzx[e]));
          }
        }
        if (MODE == 1) *reinterpret_cast<half8_t*>(dy + offs[k]) = o;
      }
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(ul * chunks + ch) * 16 + e] = s_dz[e];
      red[(ul * chunks + ch) * 16 + 8 + e] = s_dzx[e];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * c; j += 256) {
      const int which = j / c, cc = j % c;
      const int ch2 = cc >> 3, e = (cc & 7) + which * 8;
      float tot = 0.f;
      for (int l = 0; l < lanes; ++l) tot += red[(l * chunks + ch2) * 16 + e];
      partial[((size_t)blockIdx.x * 2 + which) * c + cc] = tot;
    }
  }
}

// The max-pool's backward and the reduce pass of the batch norm below it in ONE pass (ResNet root block: conv1 -> BN ->
// ReLU -> 3x3/2 pool): the activation's gradient is gathered from the pooled gradient and the forward's first-maximum
// index (pool_gather.h: what maxpool_bwd_idx_kernel computes), written for the weight gradient that reads it next, and
// enters the (sum dz, sum dz*xhat) partials while it is in registers — instead of a gather pass that writes it and a
// reduce pass (MODE 0 of bn_relu_bwd_kernel) that reads it and y back: 839 MB less traffic at 64 x 640^2.  Same unit ->
// thread map, summation order and 16-bit rounding as those two kernels, so the same numbers bit for bit.
template <int K, int S>
__global__ __launch_bounds__(256) void bn_relu_bwd_gather_kernel(
    BnBwdP p, PoolGather pg, const half_t* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ partial, half_t* __restrict__ da_out) {
  __shared__ float red[256 * 16];
  const int c = p.c, chunks = c >> 3;
  const int lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const unsigned units = (unsigned)p.n * p.h * p.w;          // (< 2^31: checked by the launcher — 32-bit index arithmetic)
  float sc[8], sh[8], mu[8], is[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cc = ch * 8 + e;
    sc[e] = scale[cc];
    sh[e] = shift[cc];
    mu[e] = mean[cc];
    is[e] = invstd[cc];
  }
  float s_dz[8], s_dzx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s_dz[e] = 0.f; s_dzx[e] = 0.f; }
  for (unsigned u = blockIdx.x * lanes + ul; u < units; u += gridDim.x * lanes) {
    const unsigned t = u / (unsigned)p.w;
    const int ix = (int)(u - t * (unsigned)p.w);
    const int img = (int)(t / (unsigned)p.h);
    const int iy = (int)(t - (unsigned)img * (unsigned)p.h);
    const unsigned off = u * (unsigned)c + (unsigned)ch * 8u;
    const half8_t v = *reinterpret_cast<const half8_t*>(y + off);
    const half8_t g = pool_gather8_f16<K, S>(pg, img, iy, ix, c, ch);
    if (da_out != nullptr) *reinterpret_cast<half8_t*>(da_out + off) = g;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float yv = (float)v[e];
      const float z = (float)(half_t)(yv * sc[e] + sh[e]);  // the stored f16 activation
      const float dz = (!p.relu || z > 0.f) ? (float)g[e] : 0.f;
      const float xh = (yv - mu[e]) * is[e];
      s_dz[e] += dz;
      s_dzx[e] += dz * xh;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(ul * chunks + ch) * 16 + e] = s_dz[e];
    red[(ul * chunks + ch) * 16 + 8 + e] = s_dzx[e];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * c; j += 256) {
    const int which = j / c, cc = j % c;
    const int ch2 = cc >> 3, e = (cc & 7) + which * 8;
    float tot = 0.f;
    for (int l = 0; l < lanes; ++l) tot += red[(l * chunks + ch2) * 16 + e];
    partial[((size_t)blockIdx.x * 2 + which) * c + cc] = tot;
  }
}

// Pooled layers whose only consumer is the pool (conv1_2, conv2_2): with the forward's first-max
// index and the pooled activation at hand the backward needs no recomputation of the four candidate
// activations — the routed gradient is da_pool where the pooled activation was positive (ReLU; bit 2 of
// the stored byte) at the stored position (bits 0-1) and 0
// elsewhere.  Same unit -> thread map, same summation order and the same numbers as the pooled branch
// of bn_relu_bwd_kernel (which spends ~25 VALU operations per full-resolution element re-deriving the
// argmax and was VALU-bound: 2.8 / 3.6 TB/s); this one is HBM-bound.
template <int MODE>
__global__ __launch_bounds__(256) void bn_pool_bwd_idx_kernel(
    BnBwdP p, const half_t* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ dgamma,
    const float* __restrict__ dbeta, const half_t* __restrict__ a_pool, const unsigned char* __restrict__ argmax,
    const half_t* __restrict__ da_pool, float* __restrict__ partial, half_t* __restrict__ dy) {
  __shared__ float red[256 * 16];
  const int c = p.c, chunks = c >> 3;
  const int lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const int oh = (p.h + 1) / 2, ow = (p.w + 1) / 2;
  const size_t units = (size_t)p.n * oh * ow;
  float sc[8], mu[8], is[8], k_dz[8], k_dzx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cc = ch * 8 + e;
    sc[e] = scale[cc];
    mu[e] = mean[cc];
    is[e] = invstd[cc];
    if (MODE == 1) {
      k_dz[e] = dbeta[cc] * p.inv_count;
      k_dzx[e] = dgamma[cc] * p.inv_count;
    }
  }
  float s_dz[8], s_dzx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s_dz[e] = 0.f; s_dzx[e] = 0.f; }
  for (size_t u = (size_t)blockIdx.x * lanes + ul; u < units; u += (size_t)gridDim.x * lanes) {
    const int ox = (int)(u % ow);
    const size_t t = u / ow;
    const int oy = (int)(t % oh);
    const int img = (int)(t / oh);
    const half8_t gp = *reinterpret_cast<const half8_t*>(da_pool + u * c + ch * 8);
    const unsigned long long am = *reinterpret_cast<const unsigned long long*>(argmax + u * c + ch * 8);
    half8_t v[4];
    bool valid[4];
    size_t offs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int iy = oy * 2 + (k >> 1), ix = ox * 2 + (k & 1);
      valid[k] = iy < p.h && ix < p.w;
      offs[k] = (((size_t)img * p.h + (valid[k] ? iy : 0)) * p.w + (valid[k] ? ix : 0)) * c + ch * 8;
      if (valid[k]) v[k] = *reinterpret_cast<const half8_t*>(y + offs[k]);
    }
    float g[8];
    int arg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      arg[e] = (int)((am >> (8 * e)) & 3ull);
      g[e] = (!p.relu || ((am >> (8 * e)) & 4ull)) ? (float)gp[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!valid[k]) continue;
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dz = arg[e] == k ? g[e] : 0.f;
        const float xh = ((float)v[k][e] - mu[e]) * is[e];
        if (MODE == 0) {
          s_dz[e] += dz;
          s_dzx[e] += dz * xh;
        } else {
          o[e] = (half_t)(sc[e] * (dz - k_dz[e] - xh * k_dzx[e]));
        }
      }
      if (MODE == 1) *reinterpret_cast<half8_t*>(dy + offs[k]) = o;
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(ul * chunks + ch) * 16 + e] = s_dz[e];
      red[(ul * chunks + ch) * 16 + 8 + e] = s_dzx[e];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * c; j += 256) {
      const int which = j / c, cc = j % c;
      const int ch2 = cc >> 3, e = (cc & 7) + which * 8;
      float tot = 0.f;
      for (int l = 0; l < lanes; ++l) tot += red[(l * chunks + ch2) * 16 + e];
      partial[((size_t)blockIdx.x * 2 + which) * c + cc] = tot;
    }
  }
}

// -------------------------------------------------- bias + ReLU backward
// dz = da * [a > 0] (a = the stored conv output after bias+ReLU); per-block column sums of dz
// for the bias gradient.  nets/pixellink.py:41-48 (slim.conv2d with biases, ReLU).
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const half_t* __restrict__ a,
                                                            const half_t* __restrict__ da,
                                                            size_t npix, int c, int relu,
                                                            half_t* __restrict__ dz,
                                                            float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int chunks = c >> 3;
  const int lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (size_t u = (size_t)blockIdx.x * lanes + ul; u < npix; u += (size_t)gridDim.x * lanes) {
    const size_t off = u * c + ch * 8;
    half8_t av = *reinterpret_cast<const half8_t*>(a + off);
    half8_t g = *reinterpret_cast<const half8_t*>(da + off);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = (!relu || (float)av[e] > 0.f) ? (float)g[e] : 0.f;
      o[e] = (half_t)d;
      s[e] += d;
    }
    *reinterpret_cast<half8_t*>(dz + off) = o;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[(ul * chunks + ch) * 8 + e] = s[e];
  __syncthreads();
  for (int cc = threadIdx.x; cc < c; cc += 256) {
    float tot = 0.f;
    for (int l = 0; l < lanes; ++l) tot += red[(l * chunks + (cc >> 3)) * 8 + (cc & 7)];
    partial[(size_t)blockIdx.x * c + cc] = tot;
  }
}

// ------------------------------------------- per-channel statistics of an f16 tensor
// (batch norm after a sub-sampled / otherwise post-processed conv output, where the conv
// epilogue's fused statistics do not apply): partial [blocks][2][C] like OCR_CONV_STATS.
__global__ __launch_bounds__(256) void channel_stats_kernel(const half_t* __restrict__ x, size_t npix,
                                                            int c, float* __restrict__ partial) {
  __shared__ float red[256 * 16];
  const int chunks = c >> 3;
  const int lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  for (size_t u = (size_t)blockIdx.x * lanes + ul; u < npix; u += (size_t)gridDim.x * lanes) {
    half8_t v = *reinterpret_cast<const half8_t*>(x + u * c + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e];
      s[e] += f;
      q[e] += f * f;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(ul * chunks + ch) * 16 + e] = s[e];
    red[(ul * chunks + ch) * 16 + 8 + e] = q[e];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * c; j += 256) {
    const int which = j / c, cc = j % c;
    float tot = 0.f;
    for (int l = 0; l < lanes; ++l) tot += red[(l * chunks + (cc >> 3)) * 16 + (cc & 7) + which * 8];
    partial[((size_t)blockIdx.x * 2 + which) * c + cc] = tot;
  }
}

// ResNet bottleneck tail: out = relu(y*scale + shift + shortcut)   (nets/resnet_v1.py:107)
// sc_scale != nullptr: the shortcut is a projection whose batch norm is applied here too
// (shortcut = sc*sc_scale + sc_shift, rounded to 16 bits as the separate pass stored it): its normalised copy is
// never written.  bits != nullptr: also one byte per 8 channels, bit e = out[.. + e] > 0 — the ReLU mask the
// backward tail (ocr_conv2d_bnred_tail_f16) reads instead of `out` itself.
template <bool PROJ, bool BITS>
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const half_t* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const half_t* __restrict__ sc,
                                                          const float* __restrict__ sc_scale,
                                                          const float* __restrict__ sc_shift, size_t npix, int c,
                                                          half_t* __restrict__ out, unsigned char* __restrict__ bits) {
  const int chunks = c >> 3;
  const size_t total = npix * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks) * 8;
    const half8_t v = *reinterpret_cast<const half8_t*>(y + i * 8);
    const half8_t r = *reinterpret_cast<const half8_t*>(sc + i * 8);
    float a[8], b[8], pa[8], pb[8];
    *reinterpret_cast<f32x4*>(a) = *reinterpret_cast<const f32x4*>(scale + ch);
    *reinterpret_cast<f32x4*>(a + 4) = *reinterpret_cast<const f32x4*>(scale + ch + 4);
    *reinterpret_cast<f32x4*>(b) = *reinterpret_cast<const f32x4*>(shift + ch);
    *reinterpret_cast<f32x4*>(b + 4) = *reinterpret_cast<const f32x4*>(shift + ch + 4);
    if (PROJ) {
      *reinterpret_cast<f32x4*>(pa) = *reinterpret_cast<const f32x4*>(sc_scale + ch);
      *reinterpret_cast<f32x4*>(pa + 4) = *reinterpret_cast<const f32x4*>(sc_scale + ch + 4);
      *reinterpret_cast<f32x4*>(pb) = *reinterpret_cast<const f32x4*>(sc_shift + ch);
      *reinterpret_cast<f32x4*>(pb + 4) = *reinterpret_cast<const f32x4*>(sc_shift + ch + 4);
    }
    half8_t o;
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // the normalised residual is rounded to 16 bits first (it is what BN backward differentiates); explicit
      // fused multiply-adds: conv_pwx_kernel evaluates the same expression and must round identically
      const float z = (float)(half_t)__builtin_fmaf((float)v[e], a[e], b[e]);
      float rv = (float)r[e];
      if (PROJ) rv = (float)(half_t)__builtin_fmaf(rv, pa[e], pb[e]);
      const float f = z + rv;
      o[e] = (half_t)(f > 0.f ? f : 0.f);
      if (BITS) m |= (f > 0.f && (float)o[e] > 0.f ? 1u : 0u) << e;
    }
    *reinterpret_cast<half8_t*>(out + i * 8) = o;
    if (BITS) bits[i] = (unsigned char)m;
  }
}

// dz = dout * [out > 0]  (+ optionally accumulated into an existing gradient)
__global__ void relu_bwd_kernel(const half_t* __restrict__ out, const half_t* __restrict__ dout,
                                size_t n8, half_t* __restrict__ dz) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    half8_t o = *reinterpret_cast<const half8_t*>(out + i * 8);
    half8_t g = *reinterpret_cast<const half8_t*>(dout + i * 8);
    half8_t d;
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = (float)o[e] > 0.f ? g[e] : (half_t)0.f;
    *reinterpret_cast<half8_t*>(dz + i * 8) = d;
  }
}

__global__ void add_inplace_kernel(half_t* __restrict__ a, const half_t* __restrict__ b, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    half8_t x = *reinterpret_cast<const half8_t*>(a + i * 8);
    half8_t y = *reinterpret_cast<const half8_t*>(b + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (half_t)((float)x[e] + (float)y[e]);
    *reinterpret_cast<half8_t*>(a + i * 8) = x;
  }
}

// -------------------------------- legacy bilinear x2 on wide f16 maps (EAST merge branch)
// tf.image.resize_bilinear(size = 2x, align_corners=False) of TF 1.4: out[2i] = in[i],
// out[2i+1] = (in[i] + in[min(i+1, H-1)]) / 2, separably (nets/model_vgg_16.py:15-16,121).
__global__ void unpool_f16_kernel(const half_t* __restrict__ x, int n, int lh, int lw, int c,
                                  half_t* __restrict__ y) {
  const int chunks = c >> 3, H = 2 * lh, W = 2 * lw;
  const size_t total = (size_t)n * H * W * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    size_t u = i / chunks;
    const int ox = (int)(u % W);
    u /= W;
    const int oy = (int)(u % H);
    const int img = (int)(u / H);
    const int y0 = oy >> 1, x0 = ox >> 1;
    const int y1 = (oy & 1) ? (y0 + 1 < lh ? y0 + 1 : lh - 1) : y0;
    const int x1 = (ox & 1) ? (x0 + 1 < lw ? x0 + 1 : lw - 1) : x0;
    const float wy = (oy & 1) ? 0.5f : 0.f, wx = (ox & 1) ? 0.5f : 0.f;
    const half_t* b = x + (size_t)img * lh * lw * c + ch * 8;
    half8_t v00 = *reinterpret_cast<const half8_t*>(b + ((size_t)y0 * lw + x0) * c);
    half8_t v01 = *reinterpret_cast<const half8_t*>(b + ((size_t)y0 * lw + x1) * c);
    half8_t v10 = *reinterpret_cast<const half8_t*>(b + ((size_t)y1 * lw + x0) * c);
    half8_t v11 = *reinterpret_cast<const half8_t*>(b + ((size_t)y1 * lw + x1) * c);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float top = (float)v00[e] + ((float)v01[e] - (float)v00[e]) * wx;
      const float bot = (float)v10[e] + ((float)v11[e] - (float)v10[e]) * wx;
      o[e] = (half_t)(top + (bot - top) * wy);
    }
    *reinterpret_cast<half8_t*>(y + i * 8) = o;
  }
}

__global__ void unpool_bwd_f16_kernel(const half_t* __restrict__ dy, int n, int lh, int lw, int c,
                                      half_t* __restrict__ dx, int accumulate) {
  const int chunks = c >> 3, H = 2 * lh, W = 2 * lw;
  const size_t total = (size_t)n * lh * lw * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    size_t u = i / chunks;
    const int x = (int)(u % lw);
    u /= lw;
    const int y = (int)(u % lh);
    const int img = (int)(u / lh);
    const half_t* b = dy + (size_t)img * H * W * c + ch * 8;
    int ry[3] = {2 * y, 2 * y + 1, 2 * y - 1}, rx[3] = {2 * x, 2 * x + 1, 2 * x - 1};
    float wy[3] = {1.f, (y == lh - 1) ? 1.f : 0.5f, (y > 0) ? 0.5f : 0.f};
    float wx[3] = {1.f, (x == lw - 1) ? 1.f : 0.5f, (x > 0) ? 0.5f : 0.f};
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (wx[d] == 0.f) continue;
        half8_t v = *reinterpret_cast<const half8_t*>(b + ((size_t)ry[a] * W + rx[d]) * c);
        const float wgt = wy[a] * wx[d];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += wgt * (float)v[e];
      }
    }
    half8_t o;
    if (accumulate) {
      half8_t old = *reinterpret_cast<const half8_t*>(dx + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += (float)old[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)g[e];
    *reinterpret_cast<half8_t*>(dx + i * 8) = o;
  }
}

// y += unpool(t) in place, with the per-channel (sum, sum of squares) partials of the result — the tail of EAST's merge
// step h = conv1x1(concat(unpool(g), f)) evaluated as conv_a(g) upsampled + conv_b(f): a 1x1 convolution acts on the
// channel axis, the bilinear resize on the spatial axes, so they commute, and the convolution of the upsampled branch
// runs on a quarter of the pixels and the WIDE upsampled tensor (2048 channels at 1/16 scale) is never written.
// Sampling as unpool_f16_kernel; thread map and partial layout as channel_stats_kernel.
__global__ __launch_bounds__(256) void unpool_add_stats_kernel(const half_t* __restrict__ t, int n, int lh, int lw, int c,
                                                               half_t* __restrict__ y, float* __restrict__ partial) {
  __shared__ float red[256 * 16];
  const int chunks = c >> 3, H = 2 * lh, W = 2 * lw;
  const int lanes = 256 / chunks;
  const int ch = threadIdx.x % chunks, ul = threadIdx.x / chunks;
  const size_t npix = (size_t)n * H * W;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  for (size_t u = (size_t)blockIdx.x * lanes + ul; u < npix; u += (size_t)gridDim.x * lanes) {
    const int ox = (int)(u % W);
    const size_t r = u / W;
    const int oy = (int)(r % H);
    const int img = (int)(r / H);
    const int y0 = oy >> 1, x0 = ox >> 1;
    const int y1 = (oy & 1) ? (y0 + 1 < lh ? y0 + 1 : lh - 1) : y0;
    const int x1 = (ox & 1) ? (x0 + 1 < lw ? x0 + 1 : lw - 1) : x0;
    const float wy = (oy & 1) ? 0.5f : 0.f, wx = (ox & 1) ? 0.5f : 0.f;
    const half_t* b = t + (size_t)img * lh * lw * c + ch * 8;
    const half8_t v00 = *reinterpret_cast<const half8_t*>(b + ((size_t)y0 * lw + x0) * c);
    const half8_t v01 = *reinterpret_cast<const half8_t*>(b + ((size_t)y0 * lw + x1) * c);
    const half8_t v10 = *reinterpret_cast<const half8_t*>(b + ((size_t)y1 * lw + x0) * c);
    const half8_t v11 = *reinterpret_cast<const half8_t*>(b + ((size_t)y1 * lw + x1) * c);
    const half8_t old = *reinterpret_cast<const half8_t*>(y + u * c + ch * 8);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float top = (float)v00[e] + ((float)v01[e] - (float)v00[e]) * wx;
      const float bot = (float)v10[e] + ((float)v11[e] - (float)v10[e]) * wx;
      o[e] = (half_t)((float)old[e] + (top + (bot - top) * wy));
      const float f = (float)o[e];
      s[e] += f;
      q[e] += f * f;
    }
    *reinterpret_cast<half8_t*>(y + u * c + ch * 8) = o;
  }
  if (partial == nullptr) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(ul * chunks + ch) * 16 + e] = s[e];
    red[(ul * chunks + ch) * 16 + 8 + e] = q[e];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * c; j += 256) {
    const int which = j / c, cc = j % c;
    float tot = 0.f;
    for (int l = 0; l < lanes; ++l) tot += red[(l * chunks + (cc >> 3)) * 16 + (cc & 7) + which * 8];
    partial[((size_t)blockIdx.x * 2 + which) * c + cc] = tot;
  }
}

// --------------------------------------------------------- general max-pool
struct PoolP {
  int n, h, w, c, oh, ow, k, stride, pt, pl;
};

// BN: x is a RAW conv output and every window element is relu(x*scale + shift) rounded to 16 bits first — what
// bn_relu_kernel would have stored (same expression, bn_act8) — so that activation is never written when the pool is its
// only reader (ResNet root: conv1 -> BN -> ReLU -> 3x3/2 max-pool, nets/resnet_v1.py:193-194).
// KK = 2 | 3: the window compiled in — its rows are requested up front (a tap outside the map reads the image's first
// row and is skipped) instead of one dependent load per tap behind a bounds branch; 0: the run-time window.
template <bool BN, int KK = 0>
__global__ void maxpool_fwd_kernel(PoolP p, const half_t* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int relu, half_t* __restrict__ y,
                                   unsigned char* __restrict__ argmax) {
  const int chunks = p.c >> 3;
  const size_t total = (size_t)p.n * p.oh * p.ow * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    size_t u = i / chunks;
    const int ox = (int)(u % p.ow);
    u /= p.ow;
    const int oy = (int)(u % p.oh);
    const int img = (int)(u / p.oh);
    float sc[8], sh[8];
    if (BN) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = scale[ch * 8 + e]; sh[e] = shift[ch * 8 + e]; }
    }
    float m[8];
    unsigned long long am = 0xffffffffffffffffull;   // byte e = window position of the FIRST maximum
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    constexpr bool K3 = KK != 0;
    constexpr int NT = KK ? KK * KK : 1, KD = KK ? KK : 1;
    half8_t v9[NT];
    bool ok9[NT];
    if constexpr (K3) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int iy = oy * p.stride + t / KD - p.pt, ix = ox * p.stride + t % KD - p.pl;
        ok9[t] = iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
        const size_t off = ok9[t] ? (((size_t)img * p.h + iy) * p.w + ix) * p.c : (size_t)img * p.h * p.w * p.c;
        v9[t] = *reinterpret_cast<const half8_t*>(x + off + ch * 8);
      }
    }
    const int kk = K3 ? KD : p.k;
    auto tap = [&](int ky, int kx, half8_t v) __attribute__((always_inline)) {
      if (BN) {
        float f[8];
        if (relu) bn_act8<true>(v, sc, sh, f);
        else bn_act8<false>(v, sc, sh, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t)f[e];
      }
      const unsigned long long pos = (unsigned long long)(ky * kk + kx);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((float)v[e] > m[e]) {
          m[e] = (float)v[e];
          am = (am & ~(0xffull << (8 * e))) | (pos << (8 * e));
        }
    };
    if constexpr (K3) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (ok9[t]) tap(t / KD, t % KD, v9[t]);
    } else {
      for (int ky = 0; ky < p.k; ++ky)
        for (int kx = 0; kx < p.k; ++kx) {
          const int iy = oy * p.stride + ky - p.pt, ix = ox * p.stride + kx - p.pl;
          if (iy < 0 || iy >= p.h || ix < 0 || ix >= p.w) continue;
          tap(ky, kx, *reinterpret_cast<const half8_t*>(x + (((size_t)img * p.h + iy) * p.w + ix) * p.c + ch * 8));
        }
    }
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)m[e];
    *reinterpret_cast<half8_t*>(y + i * 8) = o;
    if (argmax) *reinterpret_cast<unsigned long long*>(argmax + i * 8) = am;
  }
}

// gather form: every input pixel re-derives the arg-max (first maximum in
// row-major window order) of each window that covers it.
// same gather, but the arg-max comes from the index tensor the forward pass wrote (one byte per
// output element): no re-scan of x
// K, S: the window / stride compiled in (pool_gather.h: branch-free, every window slot's operands requested up front;
// 32-bit element offsets — the launcher checks the sizes) or 0, 0 for the run-time form.
template <int K, int S>
__global__ void maxpool_bwd_idx_kernel(PoolP p, const unsigned char* __restrict__ argmax,
                                       const half_t* __restrict__ dy, half_t* __restrict__ dx,
                                       int accumulate) {
  const int chunks = p.c >> 3;
  const size_t total = (size_t)p.n * p.h * p.w * chunks;
  const PoolGather pg{argmax, dy, p.oh, p.ow, p.k, p.stride, p.pt, p.pl};
  // the old gradient under `accumulate` is requested like the gather's operands — unconditionally (a dummy row otherwise):
  // behind a flag branch hipcc would wait for the gather's loads first
  const half_t* p_old = accumulate ? dx : dy;
  const size_t m_old = accumulate ? ~(size_t)0 : 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int ch, ix, iy, img;
    if (K != 0) {                                    // (total < 2^31 here)
      const unsigned iu = (unsigned)i;
      const unsigned u0 = iu / (unsigned)chunks;
      ch = (int)(iu - u0 * (unsigned)chunks);
      const unsigned t = u0 / (unsigned)p.w;
      ix = (int)(u0 - t * (unsigned)p.w);
      img = (int)(t / (unsigned)p.h);
      iy = (int)(t - (unsigned)img * (unsigned)p.h);
    } else {
      ch = (int)(i % chunks);
      size_t u = i / chunks;
      ix = (int)(u % p.w);
      u /= p.w;
      iy = (int)(u % p.h);
      img = (int)(u / p.h);
    }
    const half8_t old = *reinterpret_cast<const half8_t*>(p_old + ((i * 8) & m_old));
    float g[8];
    pool_gather8<K, S>(pg, img, iy, ix, p.c, ch, g);
    if (accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += (float)old[e];
    }
    half8_t o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (half_t)g[e];
    *reinterpret_cast<half8_t*>(dx + i * 8) = o8;
  }
}

__global__ void maxpool_bwd_kernel(PoolP p, const half_t* __restrict__ x,
                                   const half_t* __restrict__ dy, half_t* __restrict__ dx,
                                   int accumulate) {
  const int chunks = p.c >> 3;
  const size_t total = (size_t)p.n * p.h * p.w * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    size_t u = i / chunks;
    const int ix = (int)(u % p.w);
    u /= p.w;
    const int iy = (int)(u % p.h);
    const int img = (int)(u / p.h);
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    const half_t* xb = x + (size_t)img * p.h * p.w * p.c + ch * 8;
    const int ky0 = (iy + p.pt) % p.stride, kx0 = (ix + p.pl) % p.stride;      // (as in the index variant above)
    for (int ky = ky0; ky < p.k; ky += p.stride) {
      const int ny = iy + p.pt - ky;
      if (ny < 0) break;
      const int oy = ny / p.stride;
      if (oy >= p.oh) continue;
      for (int kx = kx0; kx < p.k; kx += p.stride) {
        const int nx = ix + p.pl - kx;
        if (nx < 0) break;
        const int ox = nx / p.stride;
        if (ox >= p.ow) continue;
        // window (oy,ox): find first max
        float bv[8];
        int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bv[e] = -INFINITY; bi[e] = p.k == 1 ? 0 : -1; }
        if (p.k > 1)           // (a 1x1 window -- subsample -- has nothing to scan)
        for (int wy = 0; wy < p.k; ++wy)
          for (int wx = 0; wx < p.k; ++wx) {
            const int yy = oy * p.stride + wy - p.pt, xx = ox * p.stride + wx - p.pl;
            if (yy < 0 || yy >= p.h || xx < 0 || xx >= p.w) continue;
            half8_t v = *reinterpret_cast<const half8_t*>(xb + ((size_t)yy * p.w + xx) * p.c);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if ((float)v[e] > bv[e]) { bv[e] = (float)v[e]; bi[e] = wy * p.k + wx; }
          }
        half8_t d = *reinterpret_cast<const half8_t*>(
            dy + (((size_t)img * p.oh + oy) * p.ow + ox) * p.c + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (bi[e] == ky * p.k + kx) g[e] += (float)d[e];
      }
    }
    half8_t o;
    if (accumulate) {
      half8_t old = *reinterpret_cast<const half8_t*>(dx + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] += (float)old[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)g[e];
    *reinterpret_cast<half8_t*>(dx + i * 8) = o;
  }
}

inline unsigned stream_grid(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

int bwd_blocks(int n, int h, int w, int c, int pool) {
  const int chunks = c >> 3;
  const int lanes = 256 / chunks;
  const int oh = pool ? (h + 1) / 2 : h, ow = pool ? (w + 1) / 2 : w;
  size_t units = (size_t)n * oh * ow;
  size_t b = (units + lanes - 1) / lanes;
  if (b > 2048) b = 2048;
  return (int)b;
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int ocr_prep_images_norm_f16(const void* images_f32, int64_t npix, float m0, float m1,
                                        float m2, float div, void* out, void* stream) {
  OCR_CHECK_ARG(images_f32 && out && npix > 0 && div != 0.f);
  hipLaunchKernelGGL(prep_images_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(images_f32),
                     static_cast<half_t*>(out), (size_t)npix, m0, m1, m2, div);
  return ocr_launch_status();
}

extern "C" int ocr_prep_images_f16(const void* images_f32, int64_t npix, float m0, float m1,
                                   float m2, void* out, void* stream) {
  return ocr_prep_images_norm_f16(images_f32, npix, m0, m1, m2, 1.0f, out, stream);
}

extern "C" size_t ocr_bn_reduce_workspace(int T, int C) {
  return (size_t)ocr_cdiv(T, red_rows(T)) * 2 * C * sizeof(double);
}

extern "C" int ocr_bn_finalize(const void* partial, int T, int C, double count, const void* gamma,
                               const void* beta, float eps, float decay, void* moving_mean,
                               void* moving_var, void* scale, void* shift, void* save_mean,
                               void* save_invstd, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(partial && scale && shift && workspace && T > 0 && C > 0 && count > 0);
  OCR_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr));
  if (ws_bytes < ocr_bn_reduce_workspace(T, C)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(C, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  BnFin fin{count, static_cast<const float*>(gamma), static_cast<const float*>(beta), eps, decay,
            static_cast<float*>(moving_mean), static_cast<float*>(moving_var), static_cast<float*>(scale),
            static_cast<float*>(shift), static_cast<float*>(save_mean), static_cast<float*>(save_invstd)};
  hipLaunchKernelGGL(reduce_finalize_kernel<BnFin>, dim3(R, ocr_cdiv(C, 64)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<double*>(workspace), T, C, bn_ticket_slot(),
                     rows, fin);
  return ocr_launch_status();
}

extern "C" int ocr_bn_inference_params(const void* gamma, const void* beta, const void* moving_mean,
                                       const void* moving_var, float eps, int C, void* scale,
                                       void* shift, void* stream) {
  OCR_CHECK_ARG(moving_mean && moving_var && scale && shift && C > 0);
  hipLaunchKernelGGL(bn_inference_params_kernel, dim3(ocr_cdiv(C, 64)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(gamma),
                     static_cast<const float*>(beta), static_cast<const float*>(moving_mean),
                     static_cast<const float*>(moving_var), eps, C, static_cast<float*>(scale),
                     static_cast<float*>(shift));
  return ocr_launch_status();
}

extern "C" int ocr_bn_relu_f16(const void* y, const void* scale, const void* shift, int n, int h,
                               int w, int c, int relu, int pool, void* a_full, void* a_pool,
                               void* stream) {
  OCR_CHECK_ARG(y && scale && shift && n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(c % 8 == 0);
  OCR_CHECK_ARG(pool == 0 || pool == 2);
  OCR_CHECK_ARG(pool ? a_pool != nullptr : a_full != nullptr);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int oh = pool ? (h + 1) / 2 : h, ow = pool ? (w + 1) / 2 : w;
  const size_t total = (size_t)n * oh * ow * (c / 8);
  dim3 grid(stream_grid(total));
  const half_t* yp = static_cast<const half_t*>(y);
  const float* sc = static_cast<const float*>(scale);
  const float* sh = static_cast<const float*>(shift);
  half_t* af = static_cast<half_t*>(a_full);
  half_t* ap = static_cast<half_t*>(a_pool);
  if (pool == 0) {
    if (relu) hipLaunchKernelGGL((bn_relu_kernel<true, 0>), grid, dim3(256), 0, st, yp, sc, sh, n, h, w, c, af, ap);
    else hipLaunchKernelGGL((bn_relu_kernel<false, 0>), grid, dim3(256), 0, st, yp, sc, sh, n, h, w, c, af, ap);
  } else {
    if (relu) hipLaunchKernelGGL((bn_relu_kernel<true, 2>), grid, dim3(256), 0, st, yp, sc, sh, n, h, w, c, af, ap);
    else hipLaunchKernelGGL((bn_relu_kernel<false, 2>), grid, dim3(256), 0, st, yp, sc, sh, n, h, w, c, af, ap);
  }
  return ocr_launch_status();
}

extern "C" int ocr_bn_bwd_num_partials(int n, int h, int w, int c, int pool) {
  if (n <= 0 || h <= 0 || w <= 0 || c % 8 || !pow2(c / 8) || c / 8 > 256) return OCR_ERR_UNSUPPORTED;
  return bwd_blocks(n, h, w, c, pool);
}

extern "C" int ocr_bn_relu_bwd_f16(const void* y, const void* scale, const void* shift,
                                   const void* save_mean, const void* save_invstd,
                                   const void* da_full, const void* da_pool, int n, int h, int w,
                                   int c, int relu, int pool, void* dgamma, void* dbeta, void* dy,
                                   void* partial, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && save_mean && save_invstd && dgamma && dbeta && dy);
  OCR_CHECK_ARG(partial && workspace);
  OCR_CHECK_ARG(pool == 0 || pool == 2);
  OCR_CHECK_ARG(pool ? da_pool != nullptr : da_full != nullptr);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2(c / 8) && c / 8 <= 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = bwd_blocks(n, h, w, c, pool);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  BnBwdP p{n, h, w, c, relu, pool, (float)(1.0 / ((double)n * h * w))};
  const half_t* yp = static_cast<const half_t*>(y);
  hipLaunchKernelGGL(bn_relu_bwd_kernel<0>, dim3(T), dim3(256), 0, st, p, yp,
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     (const float*)nullptr, (const float*)nullptr,
                     static_cast<const half_t*>(da_full), static_cast<const half_t*>(da_pool),
                     static_cast<float*>(partial), (half_t*)nullptr);
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFin>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<double*>(workspace), T, c, bn_ticket_slot(),
                     rows, BnBwdFin{static_cast<float*>(dgamma), static_cast<float*>(dbeta)});
  hipLaunchKernelGGL(bn_relu_bwd_kernel<1>, dim3(T), dim3(256), 0, st, p, yp,
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     static_cast<const float*>(dgamma), static_cast<const float*>(dbeta),
                     static_cast<const half_t*>(da_full), static_cast<const half_t*>(da_pool),
                     (float*)nullptr, static_cast<half_t*>(dy));
  return ocr_launch_status();
}

// Column sums of fused-reduction partial rows [T][2][c]: out0[c] = sum of row kind 0 (sum dz: a bias gradient, or
// dbeta), out1[c] = row kind 1 (sum dz*xhat: dgamma).  The finalize launch of the BN backward on its own.
extern "C" int ocr_bn_bwd_sums(const void* partial, int T, int c, void* out0, void* out1, void* workspace,
                               size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(partial && out0 && out1 && workspace && T > 0 && c > 0);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFin>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(partial),
                     static_cast<double*>(workspace), T, c, bn_ticket_slot(), rows,
                     BnBwdFin{static_cast<float*>(out1), static_cast<float*>(out0)});
  return ocr_launch_status();
}

// The reduction half of ocr_bn_relu_bwd_f16 with the apply step handed to a consumer as coefficients
// (dy = A*dz + B*y + C): for layers whose dy has a single reader that can apply it on load — the first / root
// convolutions' weight gradients (ocr_conv2d_first_wgrad_bn_f16, ocr_conv2d_stem_wgrad_bn_f16).
extern "C" int ocr_bn_relu_bwd_reduce_f16(const void* y, const void* scale, const void* shift, const void* save_mean,
                                          const void* save_invstd, const void* da_full, const void* da_pool, int n, int h,
                                          int w, int c, int relu, void* dgamma, void* dbeta, void* coef_a, void* coef_b,
                                          void* coef_c, void* partial, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && save_mean && save_invstd && da_full && dgamma && dbeta);
  OCR_CHECK_ARG(coef_a && coef_b && coef_c && partial && workspace);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2(c / 8) && c / 8 <= 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int pool = da_pool ? 2 : 0;          // da_pool: the gradient of the layer's 2x2/2 max-pool, routed to each window's first maximum
  const int T = bwd_blocks(n, h, w, c, pool);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  BnBwdP p{n, h, w, c, relu, pool, (float)(1.0 / ((double)n * h * w))};
  hipLaunchKernelGGL(bn_relu_bwd_kernel<0>, dim3(T), dim3(256), 0, st, p, static_cast<const half_t*>(y),
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     (const float*)nullptr, (const float*)nullptr, static_cast<const half_t*>(da_full),
                     static_cast<const half_t*>(da_pool), static_cast<float*>(partial), (half_t*)nullptr);
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFinC>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<double*>(workspace), T, c, bn_ticket_slot(), rows,
                     BnBwdFinC{static_cast<float*>(dgamma), static_cast<float*>(dbeta), static_cast<const float*>(scale),
                               static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd), p.inv_count,
                               static_cast<float*>(coef_a), static_cast<float*>(coef_b), static_cast<float*>(coef_c)});
  return ocr_launch_status();
}

extern "C" int ocr_bn_relu_bwd_reduce_pooled_f16(const void* y, const void* scale, const void* shift, const void* save_mean,
                                                 const void* save_invstd, const void* da_pooled, const void* argmax,
                                                 int n, int h, int w, int c, int k, int stride, int pad_top, int pad_left,
                                                 int oh, int ow, int relu, void* da_full_out, void* dgamma, void* dbeta,
                                                 void* coef_a, void* coef_b, void* coef_c, void* partial,
                                                 void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && save_mean && save_invstd && da_pooled && argmax && dgamma && dbeta);
  OCR_CHECK_ARG(coef_a && coef_b && coef_c && partial && workspace);
  OCR_CHECK_ARG(n > 0 && h > 0 && w > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0 && k * k <= 255);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2(c / 8) && c / 8 <= 256);
  OCR_CHECK_SHAPE((long long)n * h * w * c < (1ll << 31));        // 32-bit element offsets in the gather
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = bwd_blocks(n, h, w, c, 0);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  BnBwdP p{n, h, w, c, relu, 0, (float)(1.0 / ((double)n * h * w))};
  PoolGather pg{static_cast<const unsigned char*>(argmax), static_cast<const half_t*>(da_pooled), oh, ow, k, stride,
                pad_top, pad_left};
  auto kern = (k == 3 && stride == 2) ? bn_relu_bwd_gather_kernel<3, 2> : bn_relu_bwd_gather_kernel<0, 0>;
  hipLaunchKernelGGL(kern, dim3(T), dim3(256), 0, st, p, pg, static_cast<const half_t*>(y),
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     static_cast<float*>(partial), static_cast<half_t*>(da_full_out));
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFinC>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<double*>(workspace), T, c, bn_ticket_slot(), rows,
                     BnBwdFinC{static_cast<float*>(dgamma), static_cast<float*>(dbeta), static_cast<const float*>(scale),
                               static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd), p.inv_count,
                               static_cast<float*>(coef_a), static_cast<float*>(coef_b), static_cast<float*>(coef_c)});
  return ocr_launch_status();
}

extern "C" int ocr_maxpool_f16(const void* x, int n, int h, int w, int c, int k, int stride,
                               int pad_top, int pad_left, int oh, int ow, void* y, void* argmax,
                               void* stream) {
  OCR_CHECK_ARG(x && y && n > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0 && k * k <= 255);
  OCR_CHECK_SHAPE(c % 8 == 0);
  PoolP p{n, h, w, c, oh, ow, k, stride, pad_top, pad_left};
  const size_t total = (size_t)n * oh * ow * (c / 8);
  auto kern = k == 3 ? maxpool_fwd_kernel<false, 3> : k == 2 ? maxpool_fwd_kernel<false, 2> : maxpool_fwd_kernel<false, 0>;
  hipLaunchKernelGGL(kern, dim3(stream_grid(total)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const half_t*>(x), (const float*)nullptr,
                     (const float*)nullptr, 0, static_cast<half_t*>(y), static_cast<unsigned char*>(argmax));
  return ocr_launch_status();
}

extern "C" int ocr_bn_relu_maxpool_f16(const void* bn_y, const void* scale, const void* shift, int relu, int n, int h,
                                       int w, int c, int k, int stride, int pad_top, int pad_left, int oh, int ow,
                                       void* y, void* argmax, void* stream) {
  OCR_CHECK_ARG(bn_y && scale && shift && y && n > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0 && k * k <= 255);
  OCR_CHECK_SHAPE(c % 8 == 0);
  PoolP p{n, h, w, c, oh, ow, k, stride, pad_top, pad_left};
  const size_t total = (size_t)n * oh * ow * (c / 8);
  auto kern = k == 3 ? maxpool_fwd_kernel<true, 3> : k == 2 ? maxpool_fwd_kernel<true, 2> : maxpool_fwd_kernel<true, 0>;
  hipLaunchKernelGGL(kern, dim3(stream_grid(total)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const half_t*>(bn_y),
                     static_cast<const float*>(scale), static_cast<const float*>(shift), relu,
                     static_cast<half_t*>(y), static_cast<unsigned char*>(argmax));
  return ocr_launch_status();
}

extern "C" int ocr_maxpool_bwd_f16(const void* x, const void* argmax, const void* dy, int n, int h, int w,
                                   int c, int k, int stride, int pad_top, int pad_left, int oh, int ow,
                                   void* dx, int accumulate, void* stream) {
  OCR_CHECK_ARG((x || argmax) && dy && dx && n > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0);
  OCR_CHECK_SHAPE(c % 8 == 0);
  PoolP p{n, h, w, c, oh, ow, k, stride, pad_top, pad_left};
  const size_t total = (size_t)n * h * w * (c / 8);
  if (argmax) {
    const bool small = (long long)n * h * w * c < (1ll << 31) && (long long)n * oh * ow * c < (1ll << 31);
    auto kern = !small ? maxpool_bwd_idx_kernel<0, 0>
                : (k == 2 && stride == 2) ? maxpool_bwd_idx_kernel<2, 2>
                : (k == 3 && stride == 2) ? maxpool_bwd_idx_kernel<3, 2>
                : (k == 3 && stride == 1) ? maxpool_bwd_idx_kernel<3, 1>
                : (k == 1 && stride == 2) ? maxpool_bwd_idx_kernel<1, 2>
                                          : maxpool_bwd_idx_kernel<0, 0>;
    hipLaunchKernelGGL(kern, dim3(stream_grid(total)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p, static_cast<const unsigned char*>(argmax),
                       static_cast<const half_t*>(dy), static_cast<half_t*>(dx), accumulate);
    return ocr_launch_status();
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const half_t*>(x),
                     static_cast<const half_t*>(dy), static_cast<half_t*>(dx), accumulate);
  return ocr_launch_status();
}

extern "C" int ocr_bias_relu_bwd_num_partials(int64_t npix, int c) {
  if (npix <= 0 || c % 8 || !pow2(c / 8) || c / 8 > 256) return OCR_ERR_UNSUPPORTED;
  const int lanes = 256 / (c / 8);
  int64_t b = (npix + lanes - 1) / lanes;
  if (b > 2048) b = 2048;
  return (int)b;
}

extern "C" int ocr_bias_relu_bwd_f16(const void* a, const void* da, int64_t npix, int c, int relu,
                                     void* dz, void* dbias, void* partial, void* stream) {
  OCR_CHECK_ARG(a && da && dz && dbias && partial);
  const int T = ocr_bias_relu_bwd_num_partials(npix, c);
  if (T < 0) return T;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3(T), dim3(256), 0, st, static_cast<const half_t*>(a),
                     static_cast<const half_t*>(da), (size_t)npix, c, relu, static_cast<half_t*>(dz),
                     static_cast<float*>(partial));
  hipLaunchKernelGGL(ocr_sum_rows_kernel, dim3(sum_rows_grid(c)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<float*>(dbias), c, T, 1.f);
  return ocr_launch_status();
}

extern "C" int ocr_channel_stats_num_partials(int64_t npix, int c) {
  return ocr_bias_relu_bwd_num_partials(npix, c);
}

extern "C" int ocr_channel_stats_f16(const void* x, int64_t npix, int c, void* partial, void* stream) {
  OCR_CHECK_ARG(x && partial);
  const int T = ocr_channel_stats_num_partials(npix, c);
  if (T < 0) return T;
  hipLaunchKernelGGL(channel_stats_kernel, dim3(T), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const half_t*>(x), (size_t)npix, c, static_cast<float*>(partial));
  return ocr_launch_status();
}

extern "C" int ocr_bn_add_relu_f16(const void* y, const void* scale, const void* shift,
                                   const void* shortcut, const void* sc_scale, const void* sc_shift, int64_t npix,
                                   int c, void* out, void* mask_bits, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && shortcut && out && npix > 0);
  OCR_CHECK_ARG((sc_scale == nullptr) == (sc_shift == nullptr));
  OCR_CHECK_SHAPE(c % 8 == 0);
  const dim3 grid(stream_grid((size_t)npix * (c / 8)));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const half_t* yp = static_cast<const half_t*>(y);
  const half_t* sp = static_cast<const half_t*>(shortcut);
  const float *a = static_cast<const float*>(scale), *b = static_cast<const float*>(shift);
  const float *pa = static_cast<const float*>(sc_scale), *pb = static_cast<const float*>(sc_shift);
  half_t* op = static_cast<half_t*>(out);
  unsigned char* bp = static_cast<unsigned char*>(mask_bits);
  const size_t np = (size_t)npix;
  if (pa) {
    if (bp) hipLaunchKernelGGL((bn_add_relu_kernel<true, true>), grid, dim3(256), 0, st, yp, a, b, sp, pa, pb, np, c, op, bp);
    else hipLaunchKernelGGL((bn_add_relu_kernel<true, false>), grid, dim3(256), 0, st, yp, a, b, sp, pa, pb, np, c, op, bp);
  } else {
    if (bp) hipLaunchKernelGGL((bn_add_relu_kernel<false, true>), grid, dim3(256), 0, st, yp, a, b, sp, pa, pb, np, c, op, bp);
    else hipLaunchKernelGGL((bn_add_relu_kernel<false, false>), grid, dim3(256), 0, st, yp, a, b, sp, pa, pb, np, c, op, bp);
  }
  return ocr_launch_status();
}

// BN-backward sums -> dgamma, dbeta and the per-channel coefficients of the apply step as an affine map,
// dy = A*dz + B*y + C (conv_pwx_kernel applies it while staging the pixel operand of the input-gradient GEMM, so
// the separate apply pass over the widest tensors of a ResNet unit disappears).  Same reduction launch as
// ocr_bn_relu_bwd_apply_f16's first half.
extern "C" int ocr_bn_bwd_coefficients(const void* partial, int T, int c, double count, const void* scale,
                                       const void* save_mean, const void* save_invstd, void* dgamma, void* dbeta,
                                       void* coef_a, void* coef_b, void* coef_c, void* workspace, size_t ws_bytes,
                                       void* stream) {
  OCR_CHECK_ARG(partial && scale && save_mean && save_invstd && dgamma && dbeta && coef_a && coef_b && coef_c);
  OCR_CHECK_ARG(workspace && T > 0 && c > 0 && count > 0);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFinC>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(partial),
                     static_cast<double*>(workspace), T, c, bn_ticket_slot(), rows,
                     BnBwdFinC{static_cast<float*>(dgamma), static_cast<float*>(dbeta), static_cast<const float*>(scale),
                               static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                               (float)(1.0 / count), static_cast<float*>(coef_a), static_cast<float*>(coef_b),
                               static_cast<float*>(coef_c)});
  return ocr_launch_status();
}

extern "C" int ocr_relu_bwd_f16(const void* out, const void* dout, int64_t n, void* dz, void* stream) {
  OCR_CHECK_ARG(out && dout && dz && n > 0 && n % 8 == 0);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid((size_t)n / 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const half_t*>(out),
                     static_cast<const half_t*>(dout), (size_t)n / 8, static_cast<half_t*>(dz));
  return ocr_launch_status();
}

extern "C" int ocr_add_inplace_f16(void* a, const void* b, int64_t n, void* stream) {
  OCR_CHECK_ARG(a && b && n > 0 && n % 8 == 0);
  hipLaunchKernelGGL(add_inplace_kernel, dim3(stream_grid((size_t)n / 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<half_t*>(a),
                     static_cast<const half_t*>(b), (size_t)n / 8);
  return ocr_launch_status();
}

extern "C" int ocr_unpool_f16(const void* x, int n, int lh, int lw, int c, void* y, void* stream) {
  OCR_CHECK_ARG(x && y && n > 0 && lh > 0 && lw > 0);
  OCR_CHECK_SHAPE(c % 8 == 0);
  hipLaunchKernelGGL(unpool_f16_kernel, dim3(stream_grid((size_t)n * lh * lw * 4 * (c / 8))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const half_t*>(x), n, lh, lw, c,
                     static_cast<half_t*>(y));
  return ocr_launch_status();
}

extern "C" int ocr_unpool_add_stats_f16(const void* t_low, int n, int lh, int lw, int c, void* y, void* partial,
                                        void* stream) {
  OCR_CHECK_ARG(t_low && y && n > 0 && lh > 0 && lw > 0);
  const int T = ocr_channel_stats_num_partials((int64_t)n * lh * lw * 4, c);
  if (T < 0) return T;
  hipLaunchKernelGGL(unpool_add_stats_kernel, dim3(T), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const half_t*>(t_low), n, lh, lw, c, static_cast<half_t*>(y),
                     static_cast<float*>(partial));
  return ocr_launch_status();
}

extern "C" int ocr_unpool_bwd_f16(const void* dy, int n, int lh, int lw, int c, void* dx, int accumulate,
                                  void* stream) {
  OCR_CHECK_ARG(dy && dx && n > 0 && lh > 0 && lw > 0);
  OCR_CHECK_SHAPE(c % 8 == 0);
  hipLaunchKernelGGL(unpool_bwd_f16_kernel, dim3(stream_grid((size_t)n * lh * lw * (c / 8))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const half_t*>(dy), n, lh, lw, c,
                     static_cast<half_t*>(dx), accumulate);
  return ocr_launch_status();
}

// Second half of ocr_bn_relu_bwd_f16 when the (sum dz, sum dz*xhat) partials [T][2][c] already
// exist (produced by ocr_conv2d_bnred_f16): finalise dgamma / dbeta, then the apply pass.
extern "C" int ocr_bn_relu_bwd_apply_f16(const void* y, const void* scale, const void* shift,
                                         const void* save_mean, const void* save_invstd,
                                         const void* da_full, int n, int h, int w, int c, int relu,
                                         const void* partial, int T, void* dgamma, void* dbeta,
                                         void* dy, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && save_mean && save_invstd && da_full && partial && dgamma && dbeta && dy);
  OCR_CHECK_ARG(workspace && T > 0);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2(c / 8) && c / 8 <= 256);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFin>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<double*>(workspace), T, c, bn_ticket_slot(),
                     rows, BnBwdFin{static_cast<float*>(dgamma), static_cast<float*>(dbeta)});
  BnBwdP p{n, h, w, c, relu, 0, (float)(1.0 / ((double)n * h * w))};
  const int B = bwd_blocks(n, h, w, c, 0);
  hipLaunchKernelGGL(bn_relu_bwd_kernel<1>, dim3(B), dim3(256), 0, st, p, static_cast<const half_t*>(y),
                     static_cast<const float*>(scale), static_cast<const float*>(shift),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     static_cast<const float*>(dgamma), static_cast<const float*>(dbeta),
                     static_cast<const half_t*>(da_full), (const half_t*)nullptr, (float*)nullptr,
                     static_cast<half_t*>(dy));
  return ocr_launch_status();
}

extern "C" int ocr_bn_relu_pool_idx_f16(const void* y, const void* scale, const void* shift, int n, int h, int w,
                                        int c, int relu, void* a_full, void* a_pool, void* argmax_u8,
                                        void* y_pool, void* stream) {
  OCR_CHECK_ARG(y && scale && shift && a_pool && argmax_u8 && n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(c % 8 == 0);
  const int oh = (h + 1) / 2, ow = (w + 1) / 2;
  const size_t total = (size_t)n * oh * ow * (c / 8);
  dim3 grid(stream_grid(total));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const half_t* yp = static_cast<const half_t*>(y);
  const float* sc = static_cast<const float*>(scale);
  const float* sh = static_cast<const float*>(shift);
  half_t* af = static_cast<half_t*>(a_full);
  half_t* ap = static_cast<half_t*>(a_pool);
  unsigned char* am = static_cast<unsigned char*>(argmax_u8);
  half_t* ypl = static_cast<half_t*>(y_pool);
  if (relu) hipLaunchKernelGGL((bn_relu_kernel<true, 2>), grid, dim3(256), 0, st, yp, sc, sh, n, h, w, c, af, ap, am, ypl);
  else hipLaunchKernelGGL((bn_relu_kernel<false, 2>), grid, dim3(256), 0, st, yp, sc, sh, n, h, w, c, af, ap, am, ypl);
  return ocr_launch_status();
}

extern "C" int ocr_bn_relu_pool_bwd_idx_f16(const void* y, const void* scale, const void* save_mean,
                                            const void* save_invstd, const void* a_pool, const void* argmax_u8,
                                            const void* da_pool, int n, int h, int w, int c, int relu,
                                            void* dgamma, void* dbeta, void* dy, void* partial, void* workspace,
                                            size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(y && scale && save_mean && save_invstd && argmax_u8 && da_pool && dgamma && dbeta && dy);   // a_pool: unused (its sign travels in argmax)
  OCR_CHECK_ARG(partial && workspace && n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2(c / 8) && c / 8 <= 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = bwd_blocks(n, h, w, c, 2);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  BnBwdP p{n, h, w, c, relu, 2, (float)(1.0 / ((double)n * h * w))};
  const half_t* yp = static_cast<const half_t*>(y);
  hipLaunchKernelGGL(bn_pool_bwd_idx_kernel<0>, dim3(T), dim3(256), 0, st, p, yp, static_cast<const float*>(scale),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     (const float*)nullptr, (const float*)nullptr, static_cast<const half_t*>(a_pool),
                     static_cast<const unsigned char*>(argmax_u8), static_cast<const half_t*>(da_pool),
                     static_cast<float*>(partial), (half_t*)nullptr);
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFin>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<double*>(workspace), T, c, bn_ticket_slot(),
                     rows, BnBwdFin{static_cast<float*>(dgamma), static_cast<float*>(dbeta)});
  hipLaunchKernelGGL(bn_pool_bwd_idx_kernel<1>, dim3(T), dim3(256), 0, st, p, yp, static_cast<const float*>(scale),
                     static_cast<const float*>(save_mean), static_cast<const float*>(save_invstd),
                     static_cast<const float*>(dgamma), static_cast<const float*>(dbeta),
                     static_cast<const half_t*>(a_pool), static_cast<const unsigned char*>(argmax_u8),
                     static_cast<const half_t*>(da_pool), (float*)nullptr, static_cast<half_t*>(dy));
  return ocr_launch_status();
}

// ocr_bn_relu_pool_bwd_idx_f16 without its reduction pass: the (sum dz, sum dz*xhat) partials [T][2][c] come from the
// input-gradient kernel that produced da_pool (ocr_conv2d_bnred_f16 with bn_y = y_pool, the conv output at each window's
// first maximum: dz is zero everywhere else, so the sums over the pooled positions ARE the layer's sums).
extern "C" int ocr_bn_relu_pool_bwd_idx_apply_f16(const void* y, const void* scale, const void* save_mean,
                                                  const void* save_invstd, const void* argmax_u8, const void* da_pool,
                                                  int n, int h, int w, int c, int relu, const void* partial, int T,
                                                  void* dgamma, void* dbeta, void* dy, void* workspace,
                                                  size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(y && scale && save_mean && save_invstd && argmax_u8 && da_pool && partial && dgamma && dbeta && dy);
  OCR_CHECK_ARG(workspace && T > 0 && n > 0 && h > 0 && w > 0);
  OCR_CHECK_SHAPE(c % 8 == 0 && pow2(c / 8) && c / 8 <= 256);
  if (ws_bytes < ocr_bn_reduce_workspace(T, c)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = red_rows(T), R = ocr_cdiv(T, rows);
  OCR_CHECK_SHAPE(ocr_cdiv(c, 64) <= 32 && R <= kTicketGroup * kTicketGroups);
  hipLaunchKernelGGL(reduce_finalize_kernel<BnBwdFin>, dim3(R, ocr_cdiv(c, 64)), dim3(256), 0, st,
                     static_cast<const float*>(partial), static_cast<double*>(workspace), T, c, bn_ticket_slot(),
                     rows, BnBwdFin{static_cast<float*>(dgamma), static_cast<float*>(dbeta)});
  BnBwdP p{n, h, w, c, relu, 2, (float)(1.0 / ((double)n * h * w))};
  const int B = bwd_blocks(n, h, w, c, 2);
  hipLaunchKernelGGL(bn_pool_bwd_idx_kernel<1>, dim3(B), dim3(256), 0, st, p, static_cast<const half_t*>(y),
                     static_cast<const float*>(scale), static_cast<const float*>(save_mean),
                     static_cast<const float*>(save_invstd), static_cast<const float*>(dgamma),
                     static_cast<const float*>(dbeta), (const half_t*)nullptr,
                     static_cast<const unsigned char*>(argmax_u8), static_cast<const half_t*>(da_pool),
                     (float*)nullptr, static_cast<half_t*>(dy));
  return ocr_launch_status();
}

// ---- batched finalisation of small reductions (the fuse heads: up to 8 per launch) -----------------------------------
extern "C" int ocr_bn_finalize_batch(const ocr_bn_finalize_item* items, int count, float eps, float decay, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 8);
  BnBatchTab tab;
  for (int i = 0; i < count; ++i) {
    const ocr_bn_finalize_item& a = items[i];
    OCR_CHECK_ARG(a.partial && a.scale && a.shift && a.count > 0.0);
    OCR_CHECK_SHAPE(a.T > 0 && a.T <= 2048 && a.C > 0 && a.C <= 32);
    tab.it[i].partial = static_cast<const float*>(a.partial);
    tab.it[i].T = a.T;
    tab.it[i].C = a.C;
    tab.it[i].mode = 0;
    tab.it[i].fin = BnFin{a.count, static_cast<const float*>(a.gamma), static_cast<const float*>(a.beta), eps, decay,
                          static_cast<float*>(a.moving_mean), static_cast<float*>(a.moving_var),
                          static_cast<float*>(a.scale), static_cast<float*>(a.shift),
                          static_cast<float*>(a.save_mean), static_cast<float*>(a.save_invstd)};
    tab.it[i].bwd = BnBwdFin{nullptr, nullptr};
  }
  hipLaunchKernelGGL(bn_finalize_batch_kernel, dim3(count), dim3(1024), 0, static_cast<hipStream_t>(stream), tab);
  return ocr_launch_status();
}

extern "C" int ocr_bn_bwd_sums_batch(const ocr_bn_sums_item* items, int count, void* stream) {
  OCR_CHECK_ARG(items && count > 0 && count <= 8);
  BnBatchTab tab;
  for (int i = 0; i < count; ++i) {
    const ocr_bn_sums_item& a = items[i];
    OCR_CHECK_ARG(a.partial && a.out0 && a.out1);
    OCR_CHECK_SHAPE(a.T > 0 && a.T <= 2048 && a.C > 0 && a.C <= 32);
    tab.it[i].partial = static_cast<const float*>(a.partial);
    tab.it[i].T = a.T;
    tab.it[i].C = a.C;
    tab.it[i].mode = 1;
    tab.it[i].fin = BnFin{1.0, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    tab.it[i].bwd = BnBwdFin{static_cast<float*>(a.out1), static_cast<float*>(a.out0)};   // (dgamma <- kind 1, dbeta <- kind 0)
  }
  hipLaunchKernelGGL(bn_finalize_batch_kernel, dim3(count), dim3(1024), 0, static_cast<hipStream_t>(stream), tab);
  return ocr_launch_status();
}
