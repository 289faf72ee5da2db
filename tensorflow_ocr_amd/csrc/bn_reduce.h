// Per-channel reduction of partial rows + finalisation (batch-norm forward statistics, batch-norm backward sums and
// apply coefficients) — shared by the stand-alone launches of bn_pool.hip and by the CHAINED form: the same workgroups
// appended to the grid of the kernel that produces the partial rows (conv_igemm.hip), where they wait for the producers'
// arrival counter instead of for a kernel boundary.  Round 6: a dependent launch costs ~5 us of queue time whatever it does
// (profiles/r06_gap_probe.json) on top of the ~10 us latency chain of this reduction; 30 of them sit on the critical path of
// the VGG step (15 forward, 15 backward), ~120 on ResNet-50's.  Chained, the launch and its boundary disappear and the
// result is bit-identical (same code, same order).
#pragma once
#include "common.h"

namespace ocr_bn {

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
constexpr int kTicketWords = 16 * 32 * (1 + kTicketGroups);     // [slot][channel group][0: level 2 | 1 + g: level 1]

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
  static const int forced = [] { const char* e = getenv("OCR_BN_ROWS"); return e ? atoi(e) : 0; }();   // dev sweep
  if (forced > 0) return forced;
  if (T <= 1024) return (T + 63) / 64 * 64;
  int rows = ((T + 15) / 16 + 31) / 32 * 32;
  if (rows < 256) rows = 256;
  if (rows > 2048) rows = 2048;
  return rows;
}

// The launch's body as a device function, so that the SAME code also runs as the closing workgroups of the kernel that
// produced the partial rows (bn_chain_finalize below): block (bx, by) of an R x ceil(C / 64) grid, 256 working threads
// (threads beyond 256 of a larger workgroup only take part in the barriers), `lds` = 16 388 bytes of workgroup memory.
template <typename FIN>
__device__ __forceinline__ void reduce_finalize_body(const float* __restrict__ partial, double* __restrict__ stage, int T,
                                                     int C, int slot, int rows, const FIN& fin, unsigned* tickets,
                                                     char* lds, int bx, int by, int R) {
  // a block = one 64-channel group x `rows` partial rows; thread = 4 channels (one 16-byte load per
  // row and sum) x one of 16 row lanes
  typedef double RedT[2][64];
  RedT* const red = reinterpret_cast<RedT*>(lds);                       // [16][2][64]
  unsigned& s_ticket = *reinterpret_cast<unsigned*>(lds + 16 * 2 * 64 * 8);
  const bool on = threadIdx.x < 256;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c4 = by * 64 + cl * 4;        // first of this thread's 4 channels
  const int t0 = bx * rows;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  const bool live = on && c4 < C;
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
  if (on) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[rl][0][cl * 4 + e] = s[e];
      red[rl][1][cl * 4 + e] = q[e];
    }
  }
  __syncthreads();
  const int c = by * 64 + (threadIdx.x & 63);
  const int which = threadIdx.x >> 6;             // threads 0..63: sums, 64..127: second sums
  double tot = 0.0;
  if (which < 2) {
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k][which][threadIdx.x & 63];
    if (R > 1 && c < C) stage[((size_t)bx * 2 + which) * C + c] = tot;
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
    unsigned* tk = tickets + (size_t)(slot * 32 + by) * (1 + kTicketGroups);
    const int g = bx / kTicketGroup, ng = (R + kTicketGroup - 1) / kTicketGroup;
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
    const int cc = by * 64 + (col & 63);
    double acc = 0.0;
    if (on && cc < C) {
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
    if (on) red[lane2][col >> 6][col & 63] = acc;         // (every reader of the first use passed the barriers above)
  }
  __syncthreads();
  if (threadIdx.x < 64 && c < C)
    bn_fin_apply(fin, c, red[0][0][threadIdx.x] + red[1][0][threadIdx.x], red[0][1][threadIdx.x] + red[1][1][threadIdx.x]);
}


// ------------------------------------------------------------------------------------------- chained form
// Kernel parameter block of a producer launch that carries its own finalisation.  Producer workgroups are the first
// `nprod` of the grid; the R * ncg workgroups behind them are the reduction's blocks.  Every producer calls
// bn_chain_signal() after its partial-row stores; a closing workgroup waits until all `nprod` have, then runs
// reduce_finalize_body.  Producers never wait for anything, so no placement or dispatch order can deadlock the launch (a
// closing workgroup dispatched early merely spins on a CU until the producers are through; workgroups are dealt in index
// order in practice, so they arrive when the grid drains).  The arrival counters reset themselves: the last closing
// workgroup to finish (all of them have left their wait loops by then) zeroes both.
struct BnChain {
  int kind;                 // 0: none   1: BnFin   2: BnBwdFinC
  int nprod;                // producer workgroups
  int T, C, rows, R, ncg, slot;
  const float* partial;
  double* stage;
  unsigned* tickets;        // reduce_finalize_body's ticket table
  unsigned* arrive;         // [2]: producers arrived, closing workgroups done
  BnFin fin;
  BnBwdFinC finc;
};

// A partial-row element stored so that it needs NO release fence to be seen by the closing workgroups: an agent-scope
// (sc1, write-through) store, complete at the memory side once the wave's vmcnt has drained.  A release fence here
// (__threadfence = buffer_wbl2: write back the XCD's whole L2) in every producer workgroup cost the headline step 5.7 ms
// (18.0 -> 23.7: profiles/r06_ab_chain_fence.txt) — the L2 holds the convolution's own output lines.
__device__ __forceinline__ void bn_chain_store(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// after the workgroup's partial-row stores (bn_chain_store); every thread of the workgroup calls it (uniform control flow)
__device__ __forceinline__ void bn_chain_signal(const BnChain& ch) {
  __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's row stores have been written through
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ch.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the closing workgroup `fid` (0 .. R * ncg - 1); `lds`: 16 388 bytes; every thread of the workgroup calls it
__device__ __forceinline__ void bn_chain_finalize(const BnChain& ch, char* lds, int fid) {
  if (threadIdx.x == 0) {
    // bounded (~1-2 s: 2^21 polls of >= 1024 cycles); a launch that never gets there leaves NaN-free but stale outputs AND the
    // counters dirty, which the next launch on this slot turns into a visible failure (tests/test_gpu_chain.py)
    for (int spin = 0; spin < (1 << 21); ++spin) {
      if (__hip_atomic_load(ch.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)ch.nprod) break;
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the producers' rows, not this CU's stale lines
  const int bx = fid % ch.R, by = fid / ch.R;
  if (ch.kind == 1)
    reduce_finalize_body<BnFin>(ch.partial, ch.stage, ch.T, ch.C, ch.slot, ch.rows, ch.fin, ch.tickets, lds, bx, by, ch.R);
  else
    reduce_finalize_body<BnBwdFinC>(ch.partial, ch.stage, ch.T, ch.C, ch.slot, ch.rows, ch.finc, ch.tickets, lds, bx, by, ch.R);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ch.arrive + 1, 1u) == (unsigned)(ch.R * ch.ncg - 1)) {
      ch.arrive[0] = 0u;                             // ready for the next launch that uses this slot
      ch.arrive[1] = 0u;
    }
  }
}

}  // namespace ocr_bn

// host side (bn_pool.hip): the pending finalisation armed by ocr_bn_finalize_arm / ocr_bn_bwd_coefficients_arm
namespace ocr_detail {
// fills `out` and returns true when a finalisation is armed for exactly these partial rows
bool bn_chain_take(const void* partial, int T, int C, ocr_bn::BnChain* out);
// after the producer's launch: runs an armed finalisation that no kernel took as its own launch; passes `rc` through
int bn_chain_flush(int rc, const void* partial, hipStream_t st);
}  // namespace ocr_detail
