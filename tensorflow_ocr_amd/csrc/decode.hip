// PixelLink decode: per-direction softmax, the threshold mask of pixel_detect, and the
// link-gated connected-component labelling that the reference does with a Python neighbour
// graph + DFS.
//
//   ocr_softmax_pairs   slim.softmax / tf.nn.softmax over [.., 2] pairs (nets/pixellink.py:71,
//                       test_pixellink_fast.py:53-64)
//   ocr_pixel_detect    tool/pixellink_fn.py:120-154: mask = score[0,y,x,0] > t_s and, for all 8
//                       directions, link[i,0,y,x,1] >= t_l   (batch element 0, like the reference)
//   ocr_link_cc         test_pixellink_fast.py:110-178: pixel_seg = score > t_p; for INTERIOR pixels
//                       (1 <= x <= w-2, 1 <= y <= h-2) an edge to neighbour d exists iff
//                       link_score[d][y][x] > t_l and the neighbour is in pixel_seg; components of
//                       that graph with more than `min_size` pixels get a label.
//                       Direction order: left, left_down, left_up, right, right_down, right_up,
//                       up, down.  The reference walks DIRECTED edges in dict order (order
//                       dependent); the kernel labels the weakly-connected components with a
//                       lock-free union-find whose result is order independent: the root of a
//                       component is its smallest pixel index, ids are dense in ascending root
//                       order, so labels are bit-reproducible.
#include "common.h"

namespace {

__global__ void softmax_pairs_kernel(const float* __restrict__ x, float* __restrict__ y, size_t m) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const float mx = fmaxf(a, b);
    const float ea = expf(a - mx), eb = expf(b - mx);
    const float s = ea + eb;
    y[2 * i] = ea / s;
    y[2 * i + 1] = eb / s;
  }
}

__global__ void pixel_detect_kernel(const float* __restrict__ score, const float* __restrict__ link,
                                    int n, int h, int w, float ts, float tl,
                                    unsigned char* __restrict__ mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= h * w) return;
  bool ok = score[i] > ts;                      // score [n,h,w,1], batch element 0
  const size_t dir_stride = (size_t)n * h * w * 2;
#pragma unroll
  for (int d = 0; d < 8; ++d) ok = ok && !(link[d * dir_stride + (size_t)i * 2 + 1] < tl);
  mask[i] = ok ? 1 : 0;
}

// ---- union-find on pixel indices (roots = smallest index of the set) -----------------------
__device__ __forceinline__ int uf_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(int* parent, int i) {
  for (;;) {
    const int p = uf_load(parent + i);
    if (p == i) return i;
    i = p;
  }
}
__device__ __forceinline__ void uf_unite(int* parent, int a, int b) {
  for (int guard = 0; guard < (1 << 22); ++guard) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }       // attach the larger root under the smaller
    const int old = atomicMin(parent + a, b);
    if (old == a) return;
    a = old;
  }
}

struct CcP {
  int n, h, w, min_size;
  float tp, tl;
  int ls, lo;   // element stride / offset inside link_score (1,0: [8][n][h][w]; 2,1: [8][n][h][w][2])
};

// link_logits [m][16] -> out [8][m][2]: tf.stack([softmax(link_cls[..., 2i:2i+2]) for i in 0..7])
__global__ void link_softmax_stack_kernel(const float* __restrict__ x, float* __restrict__ y, size_t m) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m * 8; i += (size_t)gridDim.x * 256) {
    const size_t px = i >> 3;
    const int d = (int)(i & 7);
    const float a = x[px * 16 + 2 * d], b = x[px * 16 + 2 * d + 1];
    const float mx = fmaxf(a, b);
    const float ea = expf(a - mx), eb = expf(b - mx);
    const float s = ea + eb;
    y[((size_t)d * m + px) * 2] = ea / s;
    y[((size_t)d * m + px) * 2 + 1] = eb / s;
  }
}

__device__ __constant__ int kDx[8] = {-1, -1, -1, 1, 1, 1, 0, 0};
__device__ __constant__ int kDy[8] = {0, 1, -1, 0, 1, -1, -1, 1};

__global__ void cc_init_kernel(CcP p, const float* __restrict__ score, int* __restrict__ parent,
                               int* __restrict__ size) {
  const size_t total = (size_t)p.n * p.h * p.w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int local = (int)(i % ((size_t)p.h * p.w));
    parent[i] = score[i] > p.tp ? local : -1;
    size[i] = 0;
  }
}

__global__ void cc_root_kernel(CcP p, int* __restrict__ parent, int* __restrict__ root,
                               int* __restrict__ size, const int* __restrict__ tcnt) {
  const size_t hw = (size_t)p.h * p.w, total = (size_t)p.n * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int img = (int)(i / hw), local = (int)(i % hw);
    int* par = parent + (size_t)img * hw;
    int r = -1;
    if (par[local] >= 0) {
      r = uf_find(par, local);
      const int c = tcnt[i];                      // > 0 only at the root of a tile-local set
      if (c > 0) atomicAdd(size + (size_t)img * hw + r, c);
    }
    root[i] = r;
  }
}

// Dense ids (ascending root index) for roots with size > min_size, component table [id] = {root, size},
// then labels — three grid-wide launches (1024-pixel blocks x images).  One workgroup per image (the
// first version) spent ~95 us per 256x256 map walking it 1024 pixels at a time on ONE compute unit,
// most of it in the dependent label = id[root[pixel]] gather.
__device__ __forceinline__ bool cc_is_kept_root(const CcP& p, const int* rt, const int* sz, int i, int hw) {
  return i < hw && rt[i] == i && sz[i] > p.min_size;
}

__global__ __launch_bounds__(1024) void cc_count_kernel(CcP p, const int* __restrict__ root,
                                                        const int* __restrict__ size, int* __restrict__ blkcnt) {
  __shared__ int wsum[16];
  const int img = blockIdx.y, hw = p.h * p.w;
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const bool flag = cc_is_kept_root(p, root + (size_t)img * hw, size + (size_t)img * hw, i, hw);
  const int c = __popcll(__ballot(flag));
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += wsum[k];
    blkcnt[(size_t)img * gridDim.x + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void cc_assign_kernel(CcP p, const int* __restrict__ root,
                                                         const int* __restrict__ size,
                                                         const int* __restrict__ blkcnt, int* __restrict__ ids,
                                                         int* __restrict__ ncomp, int* __restrict__ comps,
                                                         int max_comps) {
  __shared__ int wsum[16];
  __shared__ int s_red[16];
  const int img = blockIdx.y, hw = p.h * p.w;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int* rt = root + (size_t)img * hw;
  const int* sz = size + (size_t)img * hw;
  // ids given out by the blocks before this one (and, for the image's count, by all of them)
  int before = 0, all = 0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 1024) {
    const int c = blkcnt[(size_t)img * gridDim.x + b];
    all += c;
    if (b < (int)blockIdx.x) before += c;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_down(before, o, 64);
    all += __shfl_down(all, o, 64);
  }
  if (lane == 0) { wsum[wave] = before; s_red[wave] = all; }
  __syncthreads();
  before = 0; all = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { before += wsum[k]; all += s_red[k]; }
  __syncthreads();
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const bool flag = cc_is_kept_root(p, rt, sz, i, hw);
  const unsigned long long vote = __ballot(flag);
  if (lane == 0) wsum[wave] = __popcll(vote);
  __syncthreads();
  int off = before;
  for (int k = 0; k < wave; ++k) off += wsum[k];
  if (i < hw) {
    const int myid = flag ? off + __popcll(vote & ((1ull << lane) - 1ull)) + 1 : 0;       // 1-based dense id
    ids[(size_t)img * hw + i] = myid;
    if (flag && myid <= max_comps) {
      comps[((size_t)img * max_comps + myid - 1) * 2 + 0] = i;
      comps[((size_t)img * max_comps + myid - 1) * 2 + 1] = sz[i];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) ncomp[img] = all;
}

__global__ void cc_relabel_kernel(CcP p, const int* __restrict__ root, const int* __restrict__ ids,
                                  int* __restrict__ label) {
  const size_t hw = (size_t)p.h * p.w, total = (size_t)p.n * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = root[i];
    label[i] = r >= 0 ? ids[(i / hw) * hw + r] : 0;
  }
}

// ---- plain connected components of a binary mask (test.py:182: cv2.findContours' regions) -------
// value-pixels of `mask` joined with 4- or 8-connectivity, same union-find and dense numbering as
// above (ids ascend with each component's smallest pixel index).
__global__ void cc_init_mask_kernel(CcP p, const unsigned char* __restrict__ mask, int value,
                                    int* __restrict__ parent, int* __restrict__ size) {
  const size_t total = (size_t)p.n * p.h * p.w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int local = (int)(i % ((size_t)p.h * p.w));
    parent[i] = (mask[i] != 0) == (value != 0) ? local : -1;
    size[i] = 0;
  }
}

// ---- tiled linking -----------------------------------------------------------------------------
// The kernels above unite every edge with global atomics.  Most edges join pixels of the same 32x32 tile:
// those are united in LDS first (same union-find, LDS atomics), each pixel then points at its tile's root
// (the smallest index of its tile-local set, written as a global index), and a second launch unites only
// the edges that cross a tile border.  Roots are still the smallest pixel index of each component, so
// labels do not change.  MODE 0: link-gated 8 directions on interior pixels; 1 / 2: 4- / 8-connected mask.
__device__ __forceinline__ int lds_find(volatile int* lp, int i) {
  for (;;) {
    const int p = lp[i];
    if (p == i) return i;
    i = p;
  }
}
__device__ __forceinline__ void lds_unite(int* lp, int a, int b) {
  for (int guard = 0; guard < (1 << 20); ++guard) {
    a = lds_find(lp, a);
    b = lds_find(lp, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(lp + a, b);
    if (old == a) return;
    a = old;
  }
}

template <int MODE, bool BORDER>
__global__ __launch_bounds__(1024) void cc_union_tile_kernel(CcP p, const float* __restrict__ link,
                                                             int* __restrict__ parent, int* __restrict__ tcnt) {
  __shared__ int lp[1024];
  __shared__ int lcnt[1024];
  const int hw = p.h * p.w;
  const int img = blockIdx.y;
  const int tiles_x = (p.w + 31) >> 5;
  const int tx0 = (blockIdx.x % tiles_x) << 5, ty0 = (blockIdx.x / tiles_x) << 5;
  const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;
  const int x = tx0 + lx, y = ty0 + ly;
  const bool inimg = x < p.w && y < p.h;
  const int local = y * p.w + x;
  int* par = parent + (size_t)img * hw;
  const bool seg = inimg && par[local] >= 0;
  if (!BORDER) {
    lp[tid] = seg ? tid : -1;
    lcnt[tid] = 0;
    __syncthreads();
  }
  if (seg) {
    constexpr int ND = MODE == 0 ? 8 : (MODE == 1 ? 2 : 4);
    const bool active = MODE != 0 || (x >= 1 && x <= p.w - 2 && y >= 1 && y <= p.h - 2);
    if (active) {
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        int dx, dy;
        if (MODE == 0) { dx = kDx[d]; dy = kDy[d]; }
        else { dx = d == 0 ? 1 : (d == 1 ? 0 : (d == 2 ? -1 : 1)); dy = d == 0 ? 0 : 1; }   // right, down, down-left, down-right
        const int nx = x + dx, ny = y + dy;
        if (MODE != 0 && (nx < 0 || nx >= p.w || ny >= p.h)) continue;
        const bool inside = (nx >> 5) == (x >> 5) && (ny >> 5) == (y >> 5);
        if (inside == BORDER) continue;
        if (MODE == 0 && !(link[(((size_t)d * p.n + img) * hw + local) * p.ls + p.lo] > p.tl)) continue;
        const int q = ny * p.w + nx;
        if (BORDER) {
          if (par[q] >= 0) uf_unite(par, local, q);
        } else if (lp[(ly + dy) * 32 + lx + dx] >= 0) {
          lds_unite(lp, tid, (ly + dy) * 32 + lx + dx);
        }
      }
    }
  }
  if (!BORDER) {
    __syncthreads();
    int r = -1;
    if (seg) {
      r = lds_find(lp, tid);
      par[local] = (ty0 + (r >> 5)) * p.w + tx0 + (r & 31);
      atomicAdd(lcnt + r, 1);
    }
    __syncthreads();
    // pixels per tile-local set, kept at the set's root: the size pass then needs one global atomic per
    // tile-local set instead of one per pixel
    if (inimg) tcnt[(size_t)img * hw + local] = lcnt[tid];
  }
}

// test.py:45-74 `pixel_detect` (the EAST-script twin of tool/pixellink_fn.pixel_detect):
//   res = score > t_s;  for each of the 8 link channels: link_text = argwhere(link[..., 2i+1] < t_l);
//   res[link_text[0], link_text[1]] = 0
// i.e. only the FIRST TWO below-threshold pixels (raster order) of each channel are looked at, and
// they are used as (row list, column list).  The kernels find those two indices per channel; the
// O(16) index arithmetic and the IndexError cases stay on the host, as in the reference.
__global__ void east_mask_first_kernel(const float* __restrict__ score, const float* __restrict__ link16, int hw,
                                       float ts, float tl, unsigned char* __restrict__ mask, int* __restrict__ first) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
    mask[i] = score[i] > ts ? 1 : 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (link16[(size_t)i * 16 + 2 * c + 1] < tl && i < __hip_atomic_load(first + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(first + c, i);          // the minimum only falls: a read that already covers i proves the atomic a no-op
  }
}
__global__ void east_second_kernel(const float* __restrict__ link16, int hw, float tl, const int* __restrict__ first,
                                   int* __restrict__ second) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (i > first[c] && link16[(size_t)i * 16 + 2 * c + 1] < tl &&
          i < __hip_atomic_load(second + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(second + c, i);
  }
}
__global__ void zero_pixels_kernel(unsigned char* __restrict__ mask, const int* __restrict__ idx, int count) {
  if ((int)threadIdx.x < count) mask[idx[threadIdx.x]] = 0;
}

// ---- the reference's DIRECTED grouping (test_pixellink_fast.py:153-178) ------------------------------------
// The script walks its neighbour graph with a depth-first search along DIRECTED edges: for every key (interior
// segment pixel; Python-2 dict order restated as ascending pixel index) whose group is still 0 it
// collects everything reachable from the key through pixels whose group is 0, and gives the set a new gid
// only if it has more than min_size members — otherwise the members stay 0 and later keys may collect them
// again.  Result per weakly-connected component C (ocr_link_cc's label; sets never leave C):
//     repeat: seed = smallest key of C that is unassigned and not yet known to fail;
//             R = pixels reachable from seed along edges through unassigned pixels;
//             |R| > min_size ? assign R to the seed's group : every key of R is known to fail
//             (a later search from k in R runs on a subset of the unassigned pixels and stays inside R).
// Components are independent, so ONE round serves one seed of every component at once; a round is a frontier
// expansion to its fixed point.  gid order = ascending seed index (the order the script meets its seeds).
// One workgroup per image: the sequential depth is the number of rounds, not the image size.
__global__ __launch_bounds__(1024) void cc_directed_kernel(CcP p, const float* __restrict__ score,
                                                           const float* __restrict__ link,
                                                           const int* __restrict__ ulabel_all,
                                                           const int* __restrict__ uncomp,
                                                           const int* __restrict__ order_all,
                                                           unsigned char* __restrict__ ws8,
                                                           int* __restrict__ ws32, int* __restrict__ labels_all,
                                                           int* __restrict__ ncomp_out, int* __restrict__ comps_out,
                                                           int max_comps) {
  __shared__ int s_scan[16];
  __shared__ int s_base;
  const int img = blockIdx.x, hw = p.h * p.w, tid = threadIdx.x;
  const int* ulabel = ulabel_all + (size_t)img * hw;
  unsigned char* edges = ws8 + (size_t)img * hw * 4;
  unsigned char* state = edges + hw;      // 0 outside every kept component, 1 unassigned, 2 assigned
  unsigned char* reach = state + hw;      // 0, 1 = reached, not expanded yet, 2 = expanded
  unsigned char* dead = reach + hw;       // key that cannot seed a group any more
  int* group = ws32 + (size_t)img * hw * 5;   // seed index + 1 of the pixel's group (0: none)
  int* seed = group + hw;                 // per union label: RANK of this round's seed; later: dense ids per pixel
  int* cnt = seed + hw;                   // per union label: |R| of this round
  int* gsize = cnt + hw;                  // per seed pixel: size of its group
  int* rank = gsize + hw;                 // per pixel: position of the key in the script's key order (INT_MAX: not a key)
  // order[r] = the r-th key the script's `for i in graph.keys()` meets (-1 padded), NULL = ascending pixel index
  const int* order = order_all ? order_all + (size_t)img * hw : nullptr;
#define CCD_PIX(r) (order ? order[r] : (r))
  int* labels = labels_all + (size_t)img * hw;
  const float* sc = score + (size_t)img * hw;
  const int K = min(uncomp[img], hw - 1);
  for (int i = tid; i < hw; i += 1024) {
    const int L = ulabel[i];
    const int x = i % p.w, y = i / p.w;
    unsigned e = 0;
    if (L > 0 && x >= 1 && x <= p.w - 2 && y >= 1 && y <= p.h - 2) {
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int q = (y + kDy[d]) * p.w + x + kDx[d];
        if (link[(((size_t)d * p.n + img) * hw + i) * p.ls + p.lo] > p.tl && sc[q] > p.tp) e |= 1u << d;
      }
    }
    edges[i] = (unsigned char)e;
    state[i] = L > 0 ? 1 : 0;
    reach[i] = 0;
    dead[i] = 0;
    group[i] = 0;
    gsize[i] = 0;
    const bool interior = x >= 1 && x <= p.w - 2 && y >= 1 && y <= p.h - 2;
    rank[i] = (order == nullptr && interior) ? i : 0x7fffffff;
  }
  for (int c = tid; c <= K; c += 1024) { seed[c] = 0x7fffffff; cnt[c] = 0; }
  __syncthreads();
  if (order) {
    for (int r = tid; r < hw; r += 1024) {
      const int k = order[r];
      if (k >= 0 && k < hw) rank[k] = r;
    }
    __syncthreads();
  }
  for (int round = 0; round < hw; ++round) {
    for (int i = tid; i < hw; i += 1024) {
      if (state[i] == 1 && !dead[i]) {
        const int rk = rank[i];
        if (rk < seed[ulabel[i]]) atomicMin(seed + ulabel[i], rk);
      }
    }
    __syncthreads();
    int any = 0;
    for (int c = 1 + tid; c <= K; c += 1024)
      if (seed[c] != 0x7fffffff) { reach[CCD_PIX(seed[c])] = 1; any = 1; }
    if (!__syncthreads_or(any)) break;
    for (int sweep = 0; sweep < hw; ++sweep) {
      int changed = 0;
      for (int i = tid; i < hw; i += 1024) {
        if (reach[i] != 1) continue;
        reach[i] = 2;
        const unsigned e = edges[i];
        const int x = i % p.w, y = i / p.w;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          if (!(e >> d & 1)) continue;
          const int q = (y + kDy[d]) * p.w + x + kDx[d];
          if (state[q] == 1 && reach[q] == 0) { reach[q] = 1; changed = 1; }
        }
      }
      if (!__syncthreads_or(changed)) break;
    }
    for (int i = tid; i < hw; i += 1024)
      if (reach[i]) atomicAdd(cnt + ulabel[i], 1);
    __syncthreads();
    for (int i = tid; i < hw; i += 1024) {
      if (!reach[i]) continue;
      const int c = ulabel[i];
      if (cnt[c] > p.min_size) {
        const int sp = CCD_PIX(seed[c]);
        group[i] = sp + 1;
        state[i] = 2;
        if (i == sp) gsize[i] = cnt[c];
      } else {
        dead[i] = 1;
      }
      reach[i] = 0;
    }
    __syncthreads();
    for (int c = 1 + tid; c <= K; c += 1024) { seed[c] = 0x7fffffff; cnt[c] = 0; }
    __syncthreads();
  }
  __syncthreads();
  // dense ids in the order the script meets its successful seeds (ascending rank): a pixel is one iff group[i] == i + 1
  if (tid == 0) s_base = 0;
  __syncthreads();
  int* ids = seed;                          // the per-label slots are dead now
  for (int i0 = 0; i0 < hw; i0 += 1024) {
    const int r = i0 + tid;
    const int i = r < hw ? CCD_PIX(r) : -1;
    const bool flag = i >= 0 && i < hw && group[i] == i + 1;
    const unsigned long long vote = __ballot(flag);
    if ((tid & 63) == 0) s_scan[tid >> 6] = __popcll(vote);
    __syncthreads();
    int off = s_base;
    for (int k = 0; k < (tid >> 6); ++k) off += s_scan[k];
    if (flag) {
      const int id = off + __popcll(vote & ((1ull << (tid & 63)) - 1ull)) + 1;
      ids[i] = id;
      if (id <= max_comps) {
        comps_out[((size_t)img * max_comps + id - 1) * 2 + 0] = i;
        comps_out[((size_t)img * max_comps + id - 1) * 2 + 1] = gsize[i];
      }
    }
    __syncthreads();
    if (tid == 0) {
      int t = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += s_scan[k];
      s_base += t;
    }
    __syncthreads();
  }
  for (int i = tid; i < hw; i += 1024) labels[i] = group[i] ? ids[group[i] - 1] : 0;
  if (tid == 0) ncomp_out[img] = s_base;
#undef CCD_PIX
}

unsigned dgrid(size_t items) {
  size_t b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

void cc_number(const CcP& p, const int* root, const int* size, int* ids, int* blkcnt, int* labels, int* ncomp,
               int* comps, int max_comps, hipStream_t st) {
  const unsigned bpi = (unsigned)(((size_t)p.h * p.w + 1023) / 1024);
  hipLaunchKernelGGL(cc_count_kernel, dim3(bpi, p.n), dim3(1024), 0, st, p, root, size, blkcnt);
  hipLaunchKernelGGL(cc_assign_kernel, dim3(bpi, p.n), dim3(1024), 0, st, p, root, size, blkcnt, ids, ncomp, comps,
                     max_comps);
  hipLaunchKernelGGL(cc_relabel_kernel, dim3(dgrid((size_t)p.n * p.h * p.w)), dim3(256), 0, st, p, root, ids, labels);
}

}  // namespace

extern "C" int ocr_softmax_pairs(const void* logits, int64_t pairs, void* probs, void* stream) {
  OCR_CHECK_ARG(logits && probs && pairs > 0);
  hipLaunchKernelGGL(softmax_pairs_kernel, dim3(dgrid((size_t)pairs)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(logits),
                     static_cast<float*>(probs), (size_t)pairs);
  return ocr_launch_status();
}

extern "C" int ocr_pixel_detect(const void* score_map, const void* link_scores, int n, int h, int w,
                                float score_map_thresh, float link_thresh, void* mask_u8,
                                void* stream) {
  OCR_CHECK_ARG(score_map && link_scores && mask_u8 && n > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(pixel_detect_kernel, dim3(ocr_cdiv(h * w, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(score_map),
                     static_cast<const float*>(link_scores), n, h, w, score_map_thresh, link_thresh,
                     static_cast<unsigned char*>(mask_u8));
  return ocr_launch_status();
}

extern "C" size_t ocr_link_cc_workspace(int n, int h, int w) {
  // parent, size, root, ids + one counter per 1024-pixel block
  return (size_t)n * h * w * sizeof(int) * 4 + (size_t)n * (((size_t)h * w + 1023) / 1024) * sizeof(int);
}

extern "C" int ocr_link_softmax_stack(const void* link_logits, int64_t m, void* out, void* stream) {
  OCR_CHECK_ARG(link_logits && out && m > 0);
  hipLaunchKernelGGL(link_softmax_stack_kernel, dim3(dgrid((size_t)m * 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float*>(link_logits),
                     static_cast<float*>(out), (size_t)m);
  return ocr_launch_status();
}

extern "C" int ocr_link_cc(const void* pixel_score, const void* link_score, int link_elem_stride,
                           int link_elem_offset, int n, int h, int w,
                           float pixel_thresh, float link_thresh, int min_size, void* labels_i32,
                           void* ncomp_i32, void* comps_i32, int max_comps, void* workspace,
                           size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(pixel_score && link_score && labels_i32 && ncomp_i32 && comps_i32 && workspace);
  OCR_CHECK_ARG(n > 0 && h > 2 && w > 2 && max_comps > 0);
  if (ws_bytes < ocr_link_cc_workspace(n, h, w)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t total = (size_t)n * h * w;
  int* parent = static_cast<int*>(workspace);
  int* size = parent + total;
  int* root = size + total;
  int* ids = root + total;
  int* blkcnt = ids + total;
  OCR_CHECK_ARG(link_elem_stride >= 1 && link_elem_offset >= 0 && link_elem_offset < link_elem_stride);
  CcP p{n, h, w, min_size, pixel_thresh, link_thresh, link_elem_stride, link_elem_offset};
  hipLaunchKernelGGL(cc_init_kernel, dim3(dgrid(total)), dim3(256), 0, st, p,
                     static_cast<const float*>(pixel_score), parent, size);
  const dim3 tgrid((unsigned)(((w + 31) / 32) * ((h + 31) / 32)), (unsigned)n);
  hipLaunchKernelGGL((cc_union_tile_kernel<0, false>), tgrid, dim3(1024), 0, st, p,
                     static_cast<const float*>(link_score), parent, ids);
  hipLaunchKernelGGL((cc_union_tile_kernel<0, true>), tgrid, dim3(1024), 0, st, p,
                     static_cast<const float*>(link_score), parent, ids);
  hipLaunchKernelGGL(cc_root_kernel, dim3(dgrid(total)), dim3(256), 0, st, p, parent, root, size, ids);
  cc_number(p, root, size, ids, blkcnt, static_cast<int*>(labels_i32), static_cast<int*>(ncomp_i32),
            static_cast<int*>(comps_i32), max_comps, st);
  return ocr_launch_status();
}

extern "C" size_t ocr_link_cc_directed_workspace(int n, int h, int w) {
  return (size_t)n * h * w * (4 + 5 * sizeof(int));
}

// HOST routine.  The script fills a Python-2 dict with the keys y*w + x of the interior segment pixels, x outer / y inner
// (test_pixellink_fast.py:119-150), and meets its seeds in `graph.keys()` order (:171): the slot order of CPython 2.7's
// open-addressing table for int keys (hash = the int; first slot hash & mask, then i = 5*i + perturb + 1 with perturb =
// hash, shifted right by 5 after each probe; rebuilt at the smallest power of two > (used > 50000 ? 2 : 4) * used once
// fill*3 >= size*2, old slots re-inserted in slot order).  Restated from Objects/dictobject.c (2.7); the same algorithm in
// Python is oracle/ocr_oracle.py: py27_dict_key_order, and tests hold the two equal.
static void py27_place(int32_t* tab, size_t mask, int32_t key) {
  size_t i = (size_t)key & mask;
  if (tab[i] >= 0) {
    size_t perturb = (size_t)key;
    for (;;) {
      i = (i << 2) + i + perturb + 1;
      perturb >>= 5;
      if (tab[i & mask] < 0) { i &= mask; break; }
    }
  }
  tab[i] = key;
}

extern "C" int ocr_py27_dict_order(const float* pixel_score_host, float pixel_thresh, int h, int w,
                                   int32_t* order_host) {
  if (!pixel_score_host || !order_host || h <= 2 || w <= 2 || (int64_t)h * w > (1 << 30)) return OCR_ERR_INVALID_ARG;
  size_t size = 8, used = 0;
  int32_t* tab = static_cast<int32_t*>(malloc(size * sizeof(int32_t)));
  if (!tab) return OCR_ERR_INVALID_ARG;
  for (size_t i = 0; i < size; ++i) tab[i] = -1;
  for (int x = 1; x < w - 1; ++x)
    for (int y = 1; y < h - 1; ++y) {
      if (!(pixel_score_host[(size_t)y * w + x] > pixel_thresh)) continue;   // `pixel_seg = pixel_score > threshold`, f32
      py27_place(tab, size - 1, y * w + x);
      ++used;
      if (used * 3 >= size * 2) {
        const size_t minused = (used > 50000 ? 2 : 4) * used;
        size_t nsize = 8;
        while (nsize <= minused) nsize <<= 1;
        int32_t* nt = static_cast<int32_t*>(malloc(nsize * sizeof(int32_t)));
        if (!nt) { free(tab); return OCR_ERR_INVALID_ARG; }
        for (size_t i = 0; i < nsize; ++i) nt[i] = -1;
        for (size_t i = 0; i < size; ++i)
          if (tab[i] >= 0) py27_place(nt, nsize - 1, tab[i]);
        free(tab);
        tab = nt;
        size = nsize;
      }
    }
  size_t r = 0;
  for (size_t i = 0; i < size; ++i)
    if (tab[i] >= 0) order_host[r++] = tab[i];
  free(tab);
  for (size_t i = r; i < (size_t)h * w; ++i) order_host[i] = -1;
  return (int)r;
}

extern "C" int ocr_link_cc_directed(const void* pixel_score, const void* link_score, int link_elem_stride,
                                    int link_elem_offset, int n, int h, int w, float pixel_thresh, float link_thresh,
                                    int min_size, const void* union_labels_i32, const void* union_ncomp_i32,
                                    const void* seed_order_i32, void* labels_i32, void* ncomp_i32, void* comps_i32,
                                    int max_comps, void* workspace, size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(pixel_score && link_score && union_labels_i32 && union_ncomp_i32 && labels_i32 && ncomp_i32 && comps_i32 &&
                workspace);
  OCR_CHECK_ARG(n > 0 && h > 2 && w > 2 && max_comps > 0 && labels_i32 != union_labels_i32);
  OCR_CHECK_ARG(link_elem_stride >= 1 && link_elem_offset >= 0 && link_elem_offset < link_elem_stride);
  if (ws_bytes < ocr_link_cc_directed_workspace(n, h, w)) return OCR_ERR_WORKSPACE;
  CcP p{n, h, w, min_size, pixel_thresh, link_thresh, link_elem_stride, link_elem_offset};
  const size_t total = (size_t)n * h * w;
  int* ws32 = static_cast<int*>(workspace);                                  // 16-byte aligned part first
  unsigned char* ws8 = reinterpret_cast<unsigned char*>(ws32 + total * 5);
  hipLaunchKernelGGL(cc_directed_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), p,
                     static_cast<const float*>(pixel_score), static_cast<const float*>(link_score),
                     static_cast<const int*>(union_labels_i32), static_cast<const int*>(union_ncomp_i32),
                     static_cast<const int*>(seed_order_i32), ws8, ws32,
                     static_cast<int*>(labels_i32), static_cast<int*>(ncomp_i32), static_cast<int*>(comps_i32), max_comps);
  return ocr_launch_status();
}

extern "C" int ocr_mask_cc(const void* mask_u8, int value, int connectivity, int n, int h, int w, void* labels_i32,
                           void* ncomp_i32, void* comps_i32, int max_comps, void* workspace, size_t ws_bytes,
                           void* stream) {
  OCR_CHECK_ARG(mask_u8 && labels_i32 && ncomp_i32 && comps_i32 && workspace);
  OCR_CHECK_ARG(n > 0 && h > 0 && w > 0 && max_comps > 0 && (connectivity == 4 || connectivity == 8));
  if (ws_bytes < ocr_link_cc_workspace(n, h, w)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t total = (size_t)n * h * w;
  int* parent = static_cast<int*>(workspace);
  int* size = parent + total;
  int* root = size + total;
  int* ids = root + total;
  int* blkcnt = ids + total;
  CcP p{n, h, w, 0, 0.f, 0.f, 1, 0};
  hipLaunchKernelGGL(cc_init_mask_kernel, dim3(dgrid(total)), dim3(256), 0, st, p,
                     static_cast<const unsigned char*>(mask_u8), value, parent, size);
  const dim3 tgrid((unsigned)(((w + 31) / 32) * ((h + 31) / 32)), (unsigned)n);
  if (connectivity == 8) {
    hipLaunchKernelGGL((cc_union_tile_kernel<2, false>), tgrid, dim3(1024), 0, st, p, (const float*)nullptr, parent, ids);
    hipLaunchKernelGGL((cc_union_tile_kernel<2, true>), tgrid, dim3(1024), 0, st, p, (const float*)nullptr, parent, ids);
  } else {
    hipLaunchKernelGGL((cc_union_tile_kernel<1, false>), tgrid, dim3(1024), 0, st, p, (const float*)nullptr, parent, ids);
    hipLaunchKernelGGL((cc_union_tile_kernel<1, true>), tgrid, dim3(1024), 0, st, p, (const float*)nullptr, parent, ids);
  }
  hipLaunchKernelGGL(cc_root_kernel, dim3(dgrid(total)), dim3(256), 0, st, p, parent, root, size, ids);
  cc_number(p, root, size, ids, blkcnt, static_cast<int*>(labels_i32), static_cast<int*>(ncomp_i32),
            static_cast<int*>(comps_i32), max_comps, st);
  return ocr_launch_status();
}

extern "C" int ocr_east_pixel_detect(const void* score_f32, const void* link16_f32, int h, int w, float score_thresh,
                                     float link_thresh, void* mask_u8, void* first_second_i32, void* stream) {
  OCR_CHECK_ARG(score_f32 && link16_f32 && mask_u8 && first_second_i32 && h > 0 && w > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int* fs = static_cast<int*>(first_second_i32);      // [2][8], pre-filled with INT_MAX by the caller
  const int hw = h * w;
  hipLaunchKernelGGL(east_mask_first_kernel, dim3(dgrid((size_t)hw)), dim3(256), 0, st,
                     static_cast<const float*>(score_f32), static_cast<const float*>(link16_f32), hw, score_thresh,
                     link_thresh, static_cast<unsigned char*>(mask_u8), fs);
  hipLaunchKernelGGL(east_second_kernel, dim3(dgrid((size_t)hw)), dim3(256), 0, st,
                     static_cast<const float*>(link16_f32), hw, link_thresh, fs, fs + 8);
  return ocr_launch_status();
}

namespace {
// The region to the LEFT of each region's first pixel: for a component that is the 0-region its outer
// contour lies in, for a 0-region the component whose pixels carry the hole contour — the parent
// links of cv2.findContours(RETR_TREE) (0 = the first pixel is in column 0: the frame).
__global__ void contour_parents_kernel(const int* __restrict__ labels, const int* __restrict__ zlabels,
                                       const int* __restrict__ comps, int k1, const int* __restrict__ zcomps,
                                       int k0, int w, int* __restrict__ parent_c, int* __restrict__ parent_z) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < k1) {
    const int first = comps[2 * i];
    parent_c[i] = first % w ? zlabels[first - 1] : 0;
  } else if (i < k1 + k0) {
    const int first = zcomps[2 * (i - k1)];
    parent_z[i - k1] = first % w ? labels[first - 1] : 0;
  }
}
}  // namespace

extern "C" int ocr_contour_parents(const void* labels_i32, const void* zlabels_i32, const void* comps_i32,
                                   int ncomp, const void* zcomps_i32, int nregions, int h, int w,
                                   void* parent_c_i32, void* parent_z_i32, void* stream) {
  OCR_CHECK_ARG(labels_i32 && zlabels_i32 && comps_i32 && zcomps_i32 && parent_c_i32 && parent_z_i32);
  OCR_CHECK_ARG(ncomp >= 0 && nregions >= 0 && h > 0 && w > 0);
  if (ncomp + nregions == 0) return OCR_OK;
  hipLaunchKernelGGL(contour_parents_kernel, dim3((ncomp + nregions + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const int*>(labels_i32),
                     static_cast<const int*>(zlabels_i32), static_cast<const int*>(comps_i32), ncomp,
                     static_cast<const int*>(zcomps_i32), nregions, w, static_cast<int*>(parent_c_i32),
                     static_cast<int*>(parent_z_i32));
  return ocr_launch_status();
}

extern "C" int ocr_zero_pixels_u8(void* mask_u8, const void* idx_i32, int count, void* stream) {
  OCR_CHECK_ARG(mask_u8 && idx_i32 && count >= 0 && count <= 256);
  if (count == 0) return OCR_OK;
  hipLaunchKernelGGL(zero_pixels_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<unsigned char*>(mask_u8), static_cast<const int*>(idx_i32), count);
  return ocr_launch_status();
}
