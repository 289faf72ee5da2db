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

__global__ __launch_bounds__(256) void lanms_merge_kernel(const float* __restrict__ boxes,
                                                          const int* __restrict__ counts, int max_k, float thr,
                                                          float* __restrict__ merged, int* __restrict__ n_merged,
                                                          int* __restrict__ order_ws) {
  __shared__ int s_m;
  const int img = blockIdx.x;
  const float* bx = boxes + (size_t)img * max_k * 9;
  float* mg = merged + (size_t)img * max_k * 9;
  int* order = order_ws + (size_t)img * max_k;
  int k = counts[img];
  if (k > max_k) k = max_k;

  if (threadIdx.x == 0) {                       // (1) sequential weighted merge
    int m = 0;
    bool have = false;
    float p[9], q[9];
    for (int i = 0; i < k; ++i) {
      const float* g = bx + 9 * i;
      if (have && iou_above(g, p, thr)) {
        const float sg = g[8], sp = p[8], s = sg + sp;
        for (int j = 0; j < 8; ++j) q[j] = (sg * g[j] + sp * p[j]) / s;
        q[8] = s;
        for (int j = 0; j < 9; ++j) p[j] = q[j];
      } else {
        if (have) { for (int j = 0; j < 9; ++j) mg[9 * m + j] = p[j]; ++m; }
        for (int j = 0; j < 9; ++j) p[j] = g[j];
        have = true;
      }
    }
    if (have) { for (int j = 0; j < 9; ++j) mg[9 * m + j] = p[j]; ++m; }
    s_m = m;
    n_merged[img] = m;
    __threadfence_block();
  }
  __syncthreads();
  const int m = s_m;
  for (int i = threadIdx.x; i < m; i += 256) {  // (2) stable rank by descending score
    const float si = mg[9 * i + 8];
    int r = 0;
    for (int j = 0; j < m; ++j) {
      const float sj = mg[9 * j + 8];
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
  hipLaunchKernelGGL(lanms_merge_kernel, dim3(n_images), dim3(256), 0, st, static_cast<const float*>(boxes),
                     static_cast<const int*>(counts), max_k, iou_thresh, static_cast<float*>(merged),
                     static_cast<int*>(n_merged), order);
  hipLaunchKernelGGL(lanms_mask_kernel, dim3(ocr_cdiv(max_k * words, 256), n_images), dim3(256), 0, st,
                     static_cast<const float*>(merged), static_cast<const int*>(n_merged), max_k, iou_thresh,
                     order, mask);
  hipLaunchKernelGGL(lanms_sweep_kernel, dim3(n_images), dim3(64), 0, st, static_cast<const int*>(n_merged),
                     max_k, order, mask, static_cast<int*>(keep_idx), static_cast<int*>(n_keep));
  return ocr_launch_status();
}
