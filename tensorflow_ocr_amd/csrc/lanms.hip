// Locality-aware NMS (EAST, Zhou et al. CVPR 2017, Algorithm 1) for batches of quadrangle lists.
//
// The reference tree has NO NMS; the north star asks for one, so this implements the
// published algorithm: (1) row-major weighted merge of consecutive quads with IoU > thr,
// (2) standard score-ordered NMS.  Three launches:
//   * merge (one workgroup per image): the chain is inherently sequential (each step compares with
//     the running merged quad), lane 0 walks it; then the stable score ranking as a parallel
//     counting sort (one thread per quad);
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

__device__ float quad_iou(const float* qa, const float* qb) {
  pt a[4], b[4], cur[16], nxt[16];
  load_ccw(qa, a);
  load_ccw(qb, b);
  const float area_a = fabsf(signed_area(a, 4)), area_b = fabsf(signed_area(b, 4));
  int n = 4;
  for (int i = 0; i < 4; ++i) cur[i] = a[i];
  for (int ce = 0; ce < 4 && n > 0; ++ce) {
    const pt c1 = b[ce], c2 = b[(ce + 1) % 4];
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const pt s = cur[i], e = cur[(i + 1) % n];
      const bool sin = cross3(c1, c2, s) >= 0.f, ein = cross3(c1, c2, e) >= 0.f;
      if (sin && ein) nxt[m++] = e;
      else if (sin && !ein) nxt[m++] = intersect(s, e, c1, c2);
      else if (!sin && ein) { nxt[m++] = intersect(s, e, c1, c2); nxt[m++] = e; }
    }
    n = m;
    for (int i = 0; i < m; ++i) cur[i] = nxt[i];
  }
  const float inter = n >= 3 ? fabsf(signed_area(cur, n)) : 0.f;
  const float uni = area_a + area_b - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

// Phase 1 (one workgroup per image): sequential weighted merge (lane 0), then the stable score
// ranking of the merged quads (all lanes).
// Early-out: quads whose axis-aligned bounding boxes are disjoint have an empty intersection (the
// clipper could at most leave a rounding-sized sliver), so `quad_iou(a, b) > thr` is false for any
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

__device__ bool iou_above(const float* a, const float* b, float thr) {
  if (thr >= 1e-3f && aabb_disjoint(a, b)) return false;
  return quad_iou(a, b) > thr;
}

// Wave-cooperative quad IoU for the merge chain (all 64 lanes of one wave call it with UNIFORM
// arguments; the result is uniform).  The chain is sequential in the quads, so the parallelism has to
// come from inside one IoU: lane i owns vertex i of the polygon being clipped (a quad clipped by four
// half-planes has at most 8 vertices), every Sutherland-Hodgman stage is one step for all vertices —
// neighbour by shuffle, inside tests and the intersection per lane, output slots by an 8-lane prefix
// sum, compaction through 16 points of wave-private LDS.  Every lane performs exactly the scalar
// clipper's arithmetic for its vertex and the areas are summed in vertex order, so the value equals
// quad_iou()'s bit for bit.
__device__ float quad_iou_wave(const float* qa, const float* qb, float* scratch, int lane) {
  pt a[4], b[4];
  load_ccw(qa, a);                               // uniform: computed redundantly by every lane
  load_ccw(qb, b);
  const float area_a = fabsf(signed_area(a, 4)), area_b = fabsf(signed_area(b, 4));
  pt cur;
  cur.x = lane < 4 ? a[lane & 3].x : 0.f;
  cur.y = lane < 4 ? a[lane & 3].y : 0.f;
  int n = 4;
  pt* nxt = reinterpret_cast<pt*>(scratch);      // [16]
  for (int ce = 0; ce < 4 && n > 0; ++ce) {
    const pt c1 = b[ce], c2 = b[(ce + 1) & 3];
    const int nb = (lane + 1 == n) ? 0 : lane + 1;
    pt e;
    e.x = __shfl(cur.x, nb, 64);
    e.y = __shfl(cur.y, nb, 64);
    int cnt = 0;
    pt o0 = {0.f, 0.f}, o1 = {0.f, 0.f};
    if (lane < n) {
      const pt s = cur;
      const bool sin = cross3(c1, c2, s) >= 0.f, ein = cross3(c1, c2, e) >= 0.f;
      if (sin && ein) { o0 = e; cnt = 1; }
      else if (sin && !ein) { o0 = intersect(s, e, c1, c2); cnt = 1; }
      else if (!sin && ein) { o0 = intersect(s, e, c1, c2); o1 = e; cnt = 2; }
    }
    int pos = cnt;                                // inclusive prefix sum over lanes 0..7
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const int t = __shfl_up(pos, o, 64);
      if (lane >= o) pos += t;
    }
    const int m = __shfl(pos, 7, 64);
    pos -= cnt;
    if (cnt >= 1) nxt[pos] = o0;
    if (cnt == 2) nxt[pos + 1] = o1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    n = m;
    if (lane < n) cur = nxt[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  float inter = 0.f;
  if (n >= 3) {
    // shoelace terms per lane, summed in vertex order like signed_area()
    const int nb = (lane + 1 == n) ? 0 : lane + 1;
    const float vx = __shfl(cur.x, nb, 64), vy = __shfl(cur.y, nb, 64);
    float* terms = scratch + 32;
    if (lane < n) terms[lane] = cur.x * vy - vx * cur.y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += terms[i];
    inter = fabsf(acc * 0.5f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const float uni = area_a + area_b - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

// STAGE: the image's quads are first copied into LDS by the whole workgroup (and the merged list is
// built there): the chain's per-step cost was one global-memory round trip (~1.4 us) for the next
// quad, not the clipping arithmetic.
template <bool STAGE>
__global__ __launch_bounds__(256) void lanms_merge_kernel(const float* __restrict__ boxes,
                                                          const int* __restrict__ counts, int max_k, float thr,
                                                          float* __restrict__ merged, int* __restrict__ n_merged,
                                                          int* __restrict__ order_ws) {
  extern __shared__ float s_q[];                 // STAGE: [k][9] input quads, then [k][9] merged quads
  __shared__ int s_m;
  const int img = blockIdx.x;
  const float* bx = boxes + (size_t)img * max_k * 9;
  float* mg = merged + (size_t)img * max_k * 9;
  int* order = order_ws + (size_t)img * max_k;
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

  __shared__ float s_scratch[48];
  if (threadIdx.x < 64) {                       // (1) sequential weighted merge: wave 0, cooperatively per IoU
    const int lane = threadIdx.x;
    int m = 0;
    bool have = false;
    float p[9], q[9];                           // the running merged quad, replicated in every lane
    for (int i = 0; i < k; ++i) {
      const float* g = src + 9 * i;
      bool mergeable = false;
      if (have && !(thr >= 1e-3f && aabb_disjoint(g, p))) mergeable = quad_iou_wave(g, p, s_scratch, lane) > thr;
      if (mergeable) {
        const float sg = g[8], sp = p[8], sc = sg + sp;
        for (int j = 0; j < 8; ++j) q[j] = (sg * g[j] + sp * p[j]) / sc;
        q[8] = sc;
        for (int j = 0; j < 9; ++j) p[j] = q[j];
      } else {
        if (have) {
#pragma unroll
          for (int j = 0; j < 9; ++j)
            if (lane == j) dst[9 * m + j] = p[j];
          ++m;
        }
        for (int j = 0; j < 9; ++j) p[j] = g[j];
        have = true;
      }
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
  for (int i = threadIdx.x; i < m; i += 256) {  // (2) stable rank by descending score
    const float si = dst[9 * i + 8];
    int r = 0;
    for (int j = 0; j < m; ++j) {
      const float sj = dst[9 * j + 8];
      if (sj > si || (sj == si && j < i)) ++r;
    }
    order[r] = i;
  }
}

// Phase 2 (grid = images x row blocks): the suppression matrix of the score-sorted quads, one
// 64-bit word (64 IoUs) per thread; only the strict upper triangle is evaluated.
__global__ __launch_bounds__(256) void lanms_mask_kernel(const float* __restrict__ merged,
                                                         const int* __restrict__ n_merged, int max_k, float thr,
                                                         const int* __restrict__ order_ws,
                                                         unsigned long long* __restrict__ mask_ws) {
  const int img = blockIdx.y;
  const int m = n_merged[img];
  const int words = (max_k + 63) / 64;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= m * words) return;
  const float* mg = merged + (size_t)img * max_k * 9;
  const int* order = order_ws + (size_t)img * max_k;
  unsigned long long* mask = mask_ws + (size_t)img * max_k * words;
  const int a = idx / words, wd = idx % words;
  unsigned long long bits = 0ull;
  if (wd * 64 + 63 > a) {
    const float* qa = mg + 9 * order[a];
    for (int bb = 0; bb < 64; ++bb) {
      const int b = wd * 64 + bb;
      if (b > a && b < m && iou_above(qa, mg + 9 * order[b], thr)) bits |= 1ull << bb;
    }
  }
  mask[(size_t)a * words + wd] = bits;
}

// Phase 3 (one wave per image): greedy sweep over the row masks; lane w owns mask words w, w+64, ...
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
  return (size_t)n_images * max_k * sizeof(int) + (size_t)n_images * max_k * words * sizeof(unsigned long long) + 256;
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
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int words = (max_k + 63) / 64;
  const size_t stage_bytes = (size_t)max_k * 9 * sizeof(float) * 2;
  if (stage_bytes <= 144 * 1024) {
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
                       static_cast<int*>(n_merged), order);
  } else {
    hipLaunchKernelGGL(lanms_merge_kernel<false>, dim3(n_images), dim3(256), 0, st, static_cast<const float*>(boxes),
                       static_cast<const int*>(counts), max_k, iou_thresh, static_cast<float*>(merged),
                       static_cast<int*>(n_merged), order);
  }
  hipLaunchKernelGGL(lanms_mask_kernel, dim3(ocr_cdiv(max_k * words, 256), n_images), dim3(256), 0, st,
                     static_cast<const float*>(merged), static_cast<const int*>(n_merged), max_k, iou_thresh,
                     order, mask);
  hipLaunchKernelGGL(lanms_sweep_kernel, dim3(n_images), dim3(64), 0, st, static_cast<const int*>(n_merged),
                     max_k, order, mask, static_cast<int*>(keep_idx), static_cast<int*>(n_keep));
  return ocr_launch_status();
}
