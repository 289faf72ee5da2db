// Locality-aware NMS (EAST, Zhou et al. CVPR 2017, Algorithm 1) for batches of quadrangle lists.
//
// The reference tree has NO NMS; the north star asks for one, so this implements the
// published algorithm: (1) row-major weighted merge of consecutive quads with IoU > thr,
// (2) standard score-ordered NMS.  Four launches:
//   * merge (one workgroup per image): the chain is inherently sequential only while the running quad
//     carries merged coordinates; verdicts between neighbouring INPUT quads are taken in parallel
//     first and whole runs of non-mergeable neighbours are skipped; wave 0 walks the rest;
//   * rank (grid = quad blocks x images): stable score ranking as a counting sort, one thread per quad;
//   * suppression matrix (grid = images x row blocks, fills the chip): IoU of every sorted pair by
//     convex-quad clipping, one 64-bit mask word per thread, upper triangle only;
//   * greedy sweep (one wave per image): ORs the row masks of kept quads, removed bits in registers.
// Float arithmetic is written without FMA contraction so that the kept INDICES are bit-identical
// to the plain-C oracle (oracle/lanms_oracle.c).
#include "common.h"

namespace {

#pragma clang fp contract(off)

struct pt { float x, y; };

__device__ float signed_area(const pt* p, int n) {
  float a = 0.f;
  for (int i = 0; i < n; ++i) {
    const pt u = p[i], v = p[(i + 1) % n];
    a += u.x * v.y - v.x * u.y;
  }
  return a * 0.5f;
}

__device__ float cross3(pt a, pt b, pt c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }

__device__ pt intersect(pt s, pt e, pt c1, pt c2) {
  const float d1 = cross3(c1, c2, s), d2 = cross3(c1, c2, e);
  const float t = d1 / (d1 - d2);
  pt r;
  r.x = s.x + t * (e.x - s.x);
  r.y = s.y + t * (e.y - s.y);
  return r;
}

__device__ void load_ccw(const float* q, pt* out) {
  for (int i = 0; i < 4; ++i) { out[i].x = q[2 * i]; out[i].y = q[2 * i + 1]; }
  if (signed_area(out, 4) < 0.f) {
    pt t = out[1]; out[1] = out[3]; out[3] = t;
  }
}

// IoU of two convex quads by Sutherland-Hodgman clipping (quad_iou of oracle/lanms_oracle.c, operation for
// operation), one pair per thread.  The two work polygons live in LDS (point i of thread t at [i * 256 + t]:
// one bank per lane): as private arrays the compiler places them in scratch = global memory, and a clip is a
// chain of dependent indexed reads and writes (~4x slower).  A quad clipped by four half-planes has at most
// 8 vertices.
constexpr int kClipPts = 10;
__device__ float quad_iou_lds(const float* qa, const float* qb, pt* cur, pt* nxt) {
  pt a[4], b[4];
  load_ccw(qa, a);
  load_ccw(qb, b);
  const float area_a = fabsf(signed_area(a, 4)), area_b = fabsf(signed_area(b, 4));
  int n = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) cur[i * 256] = a[i];
#pragma unroll
  for (int ce = 0; ce < 4; ++ce) {
    if (n > 0) {
      const pt c1 = b[ce], c2 = b[(ce + 1) % 4];
      int m = 0;
      for (int i = 0; i < n; ++i) {
        const pt s = cur[i * 256], e = cur[((i + 1) % n) * 256];
        const bool sin = cross3(c1, c2, s) >= 0.f, ein = cross3(c1, c2, e) >= 0.f;
        if (sin && ein) nxt[(m++) * 256] = e;
        else if (sin && !ein) nxt[(m++) * 256] = intersect(s, e, c1, c2);
        else if (!sin && ein) { nxt[(m++) * 256] = intersect(s, e, c1, c2); nxt[(m++) * 256] = e; }
      }
      n = m;
      pt* t = cur; cur = nxt; nxt = t;
    }
  }
  float inter = 0.f;
  if (n >= 3) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
      const pt u = cur[i * 256], v = cur[((i + 1) % n) * 256];
      acc += u.x * v.y - v.x * u.y;
    }
    inter = fabsf(acc * 0.5f);
  }
  const float uni = area_a + area_b - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

// Phase 1 (one workgroup per image): sequential weighted merge (lane 0), then the stable score
// ranking of the merged quads (all lanes).
// Early-out: quads whose axis-aligned bounding boxes are disjoint have an empty intersection (the
// clipper could at most leave a rounding-sized sliver), so `iou(a, b) > thr` is false for any
// thr >= 1e-3 without running the clipper.  Only that decision is consumed, so the kept indices stay
// bit-identical to the oracle; below 1e-3 the clipper always runs.
__device__ bool aabb_disjoint(const float* a, const float* b) {
  float ax0 = a[0], ax1 = a[0], ay0 = a[1], ay1 = a[1], bx0 = b[0], bx1 = b[0], by0 = b[1], by1 = b[1];
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    ax0 = fminf(ax0, a[2 * i]); ax1 = fmaxf(ax1, a[2 * i]);
    ay0 = fminf(ay0, a[2 * i + 1]); ay1 = fmaxf(ay1, a[2 * i + 1]);
    bx0 = fminf(bx0, b[2 * i]); bx1 = fmaxf(bx1, b[2 * i]);
    by0 = fminf(by0, b[2 * i + 1]); by1 = fmaxf(by1, b[2 * i + 1]);
  }
  return ax1 < bx0 || bx1 < ax0 || ay1 < by0 || by1 < ay0;
}

// Wave-cooperative quad IoU for the merge chain (all 64 lanes of one wave call it with UNIFORM
// arguments; the result is uniform).  The chain is sequential in the quads, so the parallelism has to
// come from inside one IoU: lane i owns vertex i of the polygon being clipped (a quad clipped by four
// half-planes has at most 8 vertices), every Sutherland-Hodgman stage is one step for all vertices —
// neighbour by shuffle, inside tests and the intersection per lane, output slots by an 8-lane prefix
// sum, compaction through 16 points of wave-private LDS.  Every lane performs exactly the scalar
// clipper's arithmetic for its vertex and the areas are summed in vertex order, so the value equals
// quad_iou_lds()'s (and the oracle's) bit for bit.
__device__ float quad_iou_wave(const float* qa, const float* qb, float* scratch, int lane) {
  pt a[4], b[4];
  load_ccw(qa, a);                               // uniform: computed redundantly by every lane
  load_ccw(qb, b);
  const float area_a = fabsf(signed_area(a, 4)), area_b = fabsf(signed_area(b, 4));
  // lane i holds vertex i (`cur`) and its successor (`e`) of the polygon being clipped
  pt cur = a[lane & 3], e = a[(lane + 1) & 3];
  int n = 4;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int ce = 0; ce < 4; ++ce) {
    if (n > 0) {
      pt* nxt = reinterpret_cast<pt*>(scratch) + 16 * (ce & 1);     // two buffers: no read/write overlap between stages
      const pt c1 = b[ce], c2 = b[(ce + 1) & 3];
      int cnt = 0;
      pt o0 = {0.f, 0.f}, o1 = {0.f, 0.f};
      if (lane < n) {
        const pt s = cur;
        const bool sin = cross3(c1, c2, s) >= 0.f, ein = cross3(c1, c2, e) >= 0.f;
        if (sin && ein) { o0 = e; cnt = 1; }
        else if (sin && !ein) { o0 = intersect(s, e, c1, c2); cnt = 1; }
        else if (!sin && ein) { o0 = intersect(s, e, c1, c2); o1 = e; cnt = 2; }
      }
      // output slots from two ballots (a shuffle prefix sum is four dependent LDS-crossbar trips)
      const unsigned long long b1 = __ballot(cnt >= 1), b2 = __ballot(cnt == 2);
      const int pos = __popcll(b1 & lt) + __popcll(b2 & lt);
      if (cnt >= 1) nxt[pos] = o0;
      if (cnt == 2) nxt[pos + 1] = o1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      n = __popcll(b1) + __popcll(b2);
      if (lane < n) {
        cur = nxt[lane];
        e = nxt[lane + 1 == n ? 0 : lane + 1];
      }
    }
  }
  float inter = 0.f;
  if (n >= 3) {
    // shoelace terms per lane, summed in vertex order like signed_area()
    float* terms = scratch + 64;
    if (lane < n) terms[lane] = cur.x * e.y - e.x * cur.y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = terms[i];   // one batch of broadcast reads, then the dependent adds
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < n) acc += t[i];
    inter = fabsf(acc * 0.5f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const float uni = area_a + area_b - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

// p <- (score_g * g + score_p * p) / (score_g + score_p), scores added (EAST's weighted_merge).  p is
// replicated in every lane; lane j computes coordinate j (one division deep instead of eight) and the
// eight results are broadcast back.
__device__ __forceinline__ void weighted_merge(const float* g, float* p, int lane) {
  const float sg = g[8], sp = p[8], sc = sg + sp;
  float pj = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if ((lane & 7) == j) pj = p[j];
  const float qj = (sg * g[lane & 7] + sp * pj) / sc;
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qj), j));
  p[8] = sc;
}

// Phase 0 (grid = quad blocks x images): verdict[i] = iou(g_{i+1}, g_i) > thr for every pair of
// neighbouring INPUT quads — what the chain decides whenever its running quad is still an unmerged input.
__global__ __launch_bounds__(256) void lanms_flags_kernel(const float* __restrict__ boxes,
                                                          const int* __restrict__ counts, int max_k, float thr,
                                                          unsigned char* __restrict__ flags) {
  __shared__ pt s_poly[2][kClipPts][256];
  const int img = blockIdx.y;
  int k = counts[img];
  if (k > max_k) k = max_k;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i + 1 >= k) return;
  const float* g = boxes + ((size_t)img * max_k + i + 1) * 9;
  const float* p = g - 9;
  bool v = false;
  if (!(thr >= 1e-3f && aabb_disjoint(g, p)))
    v = quad_iou_lds(g, p, &s_poly[0][0][threadIdx.x], &s_poly[1][0][threadIdx.x]) > thr;
  flags[(size_t)img * max_k + i] = v ? 1 : 0;
}

// STAGE: the image's quads are first copied into LDS by the whole workgroup (and the merged list is
// built there): the chain's per-step cost was one global-memory round trip (~1.4 us) for the next
// quad, not the clipping arithmetic.
template <bool STAGE>
__global__ __launch_bounds__(256) void lanms_merge_kernel(const float* __restrict__ boxes,
                                                          const int* __restrict__ counts, int max_k, float thr,
                                                          float* __restrict__ merged, int* __restrict__ n_merged,
                                                          const unsigned char* __restrict__ flags) {
  extern __shared__ float s_q[];                 // STAGE: [k][9] input quads, then [k][9] merged quads
  __shared__ int s_m;
  const int img = blockIdx.x;
  const float* bx = boxes + (size_t)img * max_k * 9;
  float* mg = merged + (size_t)img * max_k * 9;
  int k = counts[img];
  if (k > max_k) k = max_k;
  const float* src = bx;
  float* dst = mg;
  if (STAGE) {
    for (int i = threadIdx.x; i < k * 9; i += 256) s_q[i] = bx[i];
    __syncthreads();
    src = s_q;
    dst = s_q + (size_t)k * 9;
  }

  __shared__ float s_scratch[80];           // two 16-point clip buffers + 16 area terms
  // While the running quad p is an UNMERGED input quad g_r, the next decision is iou(g_{r+1}, g_r) > thr
  // between two inputs: those k-1 verdicts do not depend on the chain (lanms_flags_kernel).  The chain then jumps over whole runs of non-mergeable neighbours (bulk copy) and walks quad
  // by quad only while p carries merged coordinates.
  unsigned char* s_flag = reinterpret_cast<unsigned char*>(s_q + (size_t)2 * k * 9);
  if (STAGE) {
    for (int i = threadIdx.x; i + 1 < k; i += 256) s_flag[i] = flags[(size_t)img * max_k + i];
    __syncthreads();
  }
  if (threadIdx.x < 64) {                       // (1) sequential weighted merge: wave 0, cooperatively per IoU
    const int lane = threadIdx.x;
    int m = 0;
    bool have = false;
    int raw = -1;                               // p == input quad `raw` exactly (and i == raw + 1), or -1
    float p[9];                                 // the running merged quad, replicated in every lane
    int i = 0;
    while (i < k) {
      if (STAGE && have && raw >= 0) {
        // first j >= raw with flag[j] set (g_{j+1} merges into g_j), 64 verdicts per look
        int j = raw;
        bool hit = false;
        while (j + 1 < k) {
          const int t = j + lane;
          const unsigned long long vote = __ballot(t + 1 < k && s_flag[t] != 0);
          if (vote != 0ull) { j += __ffsll((long long)vote) - 1; hit = true; break; }
          j += 64;
        }
        if (!hit) j = k - 1;
        // quads raw .. j-1 are emitted unchanged; p becomes g_j
        for (int t = lane; t < (j - raw) * 9; t += 64) dst[9 * m + t] = src[9 * raw + t];
        m += j - raw;
        for (int t = 0; t < 9; ++t) p[t] = src[9 * j + t];
        raw = j;
        i = j + 1;
        if (!hit) break;
        // flag[j]: g_{j+1} merges into p = g_j
        const float* g = src + 9 * i;
        weighted_merge(g, p, lane);
        raw = -1;
        ++i;
        continue;
      }
      const float* g = src + 9 * i;
      bool mergeable = false;
      if (have && !(thr >= 1e-3f && aabb_disjoint(g, p))) mergeable = quad_iou_wave(g, p, s_scratch, lane) > thr;
      if (mergeable) {
        weighted_merge(g, p, lane);
        raw = -1;
      } else {
        if (have) {
#pragma unroll
          for (int j = 0; j < 9; ++j)
            if (lane == j) dst[9 * m + j] = p[j];
          ++m;
        }
        for (int j = 0; j < 9; ++j) p[j] = g[j];
        have = true;
        raw = i;
      }
      ++i;
    }
    if (have) {
#pragma unroll
      for (int j = 0; j < 9; ++j)
        if (lane == j) dst[9 * m + j] = p[j];
      ++m;
    }
    if (lane == 0) {
      s_m = m;
      n_merged[img] = m;
    }
    __threadfence_block();
  }
  __syncthreads();
  const int m = s_m;
  if (STAGE)
    for (int i = threadIdx.x; i < m * 9; i += 256) mg[i] = dst[i];
}

// Phase 1b (grid = quad blocks x images): stable rank by descending score as a counting sort, one thread
// per quad, the scores streamed through LDS 2048 at a time.  Inside the merge kernel this O(m^2) pass ran
// on ONE compute unit per image and cost more than the chain itself (169 of 277 us at m = 1024).
__global__ __launch_bounds__(256) void lanms_rank_kernel(const float* __restrict__ merged,
                                                         const int* __restrict__ n_merged, int max_k,
                                                         int* __restrict__ order_ws) {
  __shared__ float s_score[2048];
  const int img = blockIdx.y;
  const int m = n_merged[img];
  if ((int)blockIdx.x * 256 >= m) return;
  const float* mg = merged + (size_t)img * max_k * 9;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float si = i < m ? mg[9 * i + 8] : 0.f;
  int r = 0;
  for (int base = 0; base < m; base += 2048) {
    const int cnt = min(2048, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) s_score[t] = mg[9 * (base + t) + 8];
    __syncthreads();
#pragma unroll 8
    for (int t = 0; t < cnt; ++t) {
      const float sj = s_score[t];
      if (sj > si || (sj == si && base + t < i)) ++r;
    }
  }
  if (i < m) order_ws[(size_t)img * max_k + r] = i;
}

// Phase 2 (grid = images x row blocks): the suppression matrix of the score-sorted quads, one
// 64-bit word (64 pairs) per thread; only the strict upper triangle is evaluated.  Most pairs are
// settled by the bounding-box test; the few that need the clipper would make a wave wait on one or
// two busy lanes at a time, so candidates are queued per wave in LDS ((owner lane, bit) entries) and
// clipped 64 at a time, every lane busy, the verdicts OR-ed back into the owners' words.
__global__ __launch_bounds__(256) void lanms_mask_kernel(const float* __restrict__ merged,
                                                         const int* __restrict__ n_merged, int max_k, float thr,
                                                         const int* __restrict__ order_ws,
                                                         unsigned long long* __restrict__ mask_ws) {
  __shared__ unsigned short s_queue[4][128];
  __shared__ unsigned long long s_bits[4][64];
  __shared__ pt s_poly[2][kClipPts][256];
  const int img = blockIdx.y;
  const int m = n_merged[img];
  const int words = (max_k + 63) / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx0 = blockIdx.x * 256 + wave * 64;          // idx of this wave's lane 0
  if (idx0 >= m * words) return;                          // whole wave out of range
  const int idx = idx0 + lane;
  const bool valid = idx < m * words;
  const float* mg = merged + (size_t)img * max_k * 9;
  const int* order = order_ws + (size_t)img * max_k;
  unsigned long long* mask = mask_ws + (size_t)img * max_k * words;
  const int a = idx / words, wd = idx % words;
  const bool boxes_decide = thr >= 1e-3f;
  float ax0 = 0.f, ax1 = 0.f, ay0 = 0.f, ay1 = 0.f;
  const bool live = valid && wd * 64 + 63 > a;
  if (live) {
    const float* qa = mg + 9 * order[a];
    ax0 = fminf(fminf(qa[0], qa[2]), fminf(qa[4], qa[6])); ax1 = fmaxf(fmaxf(qa[0], qa[2]), fmaxf(qa[4], qa[6]));
    ay0 = fminf(fminf(qa[1], qa[3]), fminf(qa[5], qa[7])); ay1 = fmaxf(fmaxf(qa[1], qa[3]), fmaxf(qa[5], qa[7]));
  }
  s_bits[wave][lane] = 0ull;
  int qn = 0;                                             // wave-uniform queue length

  auto drain = [&](int first, int count) {                // entries [first, first+count): one per lane
    if (lane < count) {
      const int e = s_queue[wave][first + lane];
      const int owner = e >> 6, bit = e & 63;
      const int oi = idx0 + owner;
      const int oa = oi / words, ob = (oi % words) * 64 + bit;
      if (quad_iou_lds(mg + 9 * order[oa], mg + 9 * order[ob], &s_poly[0][0][threadIdx.x],
                       &s_poly[1][0][threadIdx.x]) > thr)
        atomicOr(&s_bits[wave][owner], 1ull << bit);
    }
  };

  for (int bb = 0; bb < 64; ++bb) {
    const int b = wd * 64 + bb;
    bool cand = live && b > a && b < m;
    if (cand && boxes_decide) {
      const float* qb = mg + 9 * order[b];
      const float bx0 = fminf(fminf(qb[0], qb[2]), fminf(qb[4], qb[6])), bx1 = fmaxf(fmaxf(qb[0], qb[2]), fmaxf(qb[4], qb[6]));
      const float by0 = fminf(fminf(qb[1], qb[3]), fminf(qb[5], qb[7])), by1 = fmaxf(fmaxf(qb[1], qb[3]), fmaxf(qb[5], qb[7]));
      cand = !(ax1 < bx0 || bx1 < ax0 || ay1 < by0 || by1 < ay0);
    }
    const unsigned long long vote = __ballot(cand);
    if (vote == 0ull) continue;
    if (cand) s_queue[wave][qn + __popcll(vote & ((1ull << lane) - 1ull))] = (unsigned short)((lane << 6) | bb);
    qn += __popcll(vote);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (qn >= 64) {
      drain(qn - 64, 64);
      qn -= 64;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (qn > 0) drain(0, qn);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (valid) mask[(size_t)a * words + wd] = s_bits[wave][lane];
}

__device__ __forceinline__ unsigned long long lane_word(unsigned long long v, int src_lane) {   // uniform src_lane
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src_lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src_lane);
  return ((unsigned long long)hi << 32) | lo;
}

// Phase 3 (one wave per image): greedy sweep over the row masks; lane w owns mask words w, w+64, ...
// A kept row's mask feeds the very next decision, so a load issued when the row is reached costs a
// full memory round trip per kept quad (~1 us x ~900).  With at most 64 words per row (max_k <= 4096)
// the rows are instead fetched 64 ahead, kept or not: one word per lane per row (16 ahead still left
// each group of rows waiting ~1 us for the next one's loads).
__global__ __launch_bounds__(64) void lanms_sweep64_kernel(const int* __restrict__ n_merged, int max_k,
                                                           const int* __restrict__ order_ws,
                                                           const unsigned long long* __restrict__ mask_ws,
                                                           int* __restrict__ keep_idx, int* __restrict__ n_keep) {
  constexpr int D = 64;                                   // rows in flight: a group must outlast one memory round trip
  __shared__ int s_keep[4096];                            // kept SORTED positions; mapped through order[] at the end
  const int img = blockIdx.x, lane = threadIdx.x;
  const int m = n_merged[img];
  const int words = (max_k + 63) / 64;                    // <= 64
  const int* order = order_ws + (size_t)img * max_k;
  const unsigned long long* mask = mask_ws + (size_t)img * max_k * words;
  unsigned long long removed = 0ull, rem = 0ull, kept = 0ull, cur[D], nxt[D];
#pragma unroll
  for (int j = 0; j < D; ++j) cur[j] = (j < m && lane < words) ? mask[(size_t)j * words + lane] : 0ull;
  int nk = 0;
  for (int a0 = 0; a0 < m; a0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int row = a0 + D + j;
      nxt[j] = (row < m && lane < words) ? mask[(size_t)row * words + lane] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int a = a0 + j;
      if (a < m) {
        // the decision chain stays in scalar registers: `rem` = the removed-word of the current 64-row
        // block, refreshed from its owner lane at block starts and updated from the kept row's own
        // word (already prefetched) — the vector OR below is off the critical path
        const int owner = __builtin_amdgcn_readfirstlane(a >> 6);
        if ((a & 63) == 0) rem = lane_word(removed, owner);
        if (!((rem >> (a & 63)) & 1ull)) {
          kept |= 1ull << (a & 63);
          removed |= cur[j];
          rem |= lane_word(cur[j], owner);
        }
        if ((a & 63) == 63 || a == m - 1) {               // block done: its kept positions, all lanes at once
          if ((kept >> lane) & 1ull) s_keep[nk + __popcll(kept & ((1ull << lane) - 1ull))] = (a & ~63) + lane;
          nk += __popcll(kept);
          kept = 0ull;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) cur[j] = nxt[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int i = lane; i < nk; i += 64) keep_idx[(size_t)img * max_k + i] = order[s_keep[i]];
  if (lane == 0) n_keep[img] = nk;
}

__global__ __launch_bounds__(64) void lanms_sweep_kernel(const int* __restrict__ n_merged, int max_k,
                                                         const int* __restrict__ order_ws,
                                                         const unsigned long long* __restrict__ mask_ws,
                                                         int* __restrict__ keep_idx, int* __restrict__ n_keep) {
  const int img = blockIdx.x;
  const int m = n_merged[img];
  const int words = (max_k + 63) / 64;
  const int* order = order_ws + (size_t)img * max_k;
  const unsigned long long* mask = mask_ws + (size_t)img * max_k * words;
  unsigned long long removed[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) removed[j] = 0ull;
  int nk = 0;
  for (int a = 0; a < m; ++a) {
    const int wd = a >> 6, owner = wd & 63, slot = wd >> 6;
    unsigned long long rv = 0ull;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == slot) rv = removed[j];
    rv = __shfl(rv, owner, 64);
    if ((rv >> (a & 63)) & 1ull) continue;
    if (threadIdx.x == 0) keep_idx[(size_t)img * max_k + nk] = order[a];
    ++nk;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int w2 = (int)threadIdx.x + 64 * j;
      if (w2 < words) removed[j] |= mask[(size_t)a * words + w2];
    }
  }
  if (threadIdx.x == 0) n_keep[img] = nk;
}

}  // namespace

extern "C" size_t ocr_lanms_workspace(int n_images, int max_k) {
  const size_t words = (size_t)(max_k + 63) / 64;
  return (size_t)n_images * max_k * sizeof(int) + (size_t)n_images * max_k * words * sizeof(unsigned long long) +
         (size_t)n_images * max_k + 512;
}

extern "C" int ocr_lanms(const void* boxes, const void* counts, int n_images, int max_k, float iou_thresh,
                         void* merged, void* n_merged, void* keep_idx, void* n_keep, void* workspace,
                         size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(boxes && counts && merged && n_merged && keep_idx && n_keep && workspace);
  OCR_CHECK_ARG(n_images > 0 && max_k > 0);
  OCR_CHECK_SHAPE(max_k <= 64 * 64 * 8);        // 8 mask words per lane in the sweep
  if (ws_bytes < ocr_lanms_workspace(n_images, max_k)) return OCR_ERR_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  int* order = reinterpret_cast<int*>(ws);
  size_t off = ((size_t)n_images * max_k * sizeof(int) + 255) / 256 * 256;
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws + off);
  const int words_ = (max_k + 63) / 64;
  unsigned char* flags = reinterpret_cast<unsigned char*>(ws + off) +
                         ((size_t)n_images * max_k * words_ * sizeof(unsigned long long) + 255) / 256 * 256;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int words = (max_k + 63) / 64;
  const size_t stage_bytes = (size_t)max_k * 9 * sizeof(float) * 2 + (size_t)max_k;     // two quad lists + verdict bytes
  if (stage_bytes <= 144 * 1024) {
    hipLaunchKernelGGL(lanms_flags_kernel, dim3(ocr_cdiv(max_k, 256), n_images), dim3(256), 0, st,
                       static_cast<const float*>(boxes), static_cast<const int*>(counts), max_k, iou_thresh, flags);
    auto kern = lanms_merge_kernel<true>;
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              144 * 1024) != hipSuccess)      // + the kernel's few static bytes <= 160 KB
        return OCR_ERR_HIP;
      configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_images), dim3(256), stage_bytes, st, static_cast<const float*>(boxes),
                       static_cast<const int*>(counts), max_k, iou_thresh, static_cast<float*>(merged),
                       static_cast<int*>(n_merged), flags);
  } else {
    hipLaunchKernelGGL(lanms_merge_kernel<false>, dim3(n_images), dim3(256), 0, st, static_cast<const float*>(boxes),
                       static_cast<const int*>(counts), max_k, iou_thresh, static_cast<float*>(merged),
                       static_cast<int*>(n_merged), flags);
  }
  hipLaunchKernelGGL(lanms_rank_kernel, dim3(ocr_cdiv(max_k, 256), n_images), dim3(256), 0, st,
                     static_cast<const float*>(merged), static_cast<const int*>(n_merged), max_k, order);
  hipLaunchKernelGGL(lanms_mask_kernel, dim3(ocr_cdiv(max_k * words, 256), n_images), dim3(256), 0, st,
                     static_cast<const float*>(merged), static_cast<const int*>(n_merged), max_k, iou_thresh,
                     order, mask);
  hipLaunchKernelGGL(words <= 64 ? lanms_sweep64_kernel : lanms_sweep_kernel, dim3(n_images), dim3(64), 0, st,
                     static_cast<const int*>(n_merged), max_k, order, mask, static_cast<int*>(keep_idx),
                     static_cast<int*>(n_keep));
  return ocr_launch_status();
}
