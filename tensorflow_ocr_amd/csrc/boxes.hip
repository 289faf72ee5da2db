// One oriented box per labelled component: the GPU side of
//     rectangle = cv2.minAreaRect(show_xy); box = np.int0(cv2.boxPoints(rectangle))
// (test_pixellink_fast.py:193-202, test_pixellink.py:207-216), for every component of a batch of
// label maps at once.
//
//   ocr_min_area_rects   labels int32 [n][h][w] (0 = background, ids 1..K from ocr_link_cc) ->
//                        per component the convex hull size, its first two hull points and the
//                        rotating-calipers result (corner + two edge vectors, float32).
//
// A component's point set is `show_xy`: X = (int)(x * sx), Y = (int)(y * sy) (the reference assigns
// the scaled floats into an integer array: truncation).  Only a component's leftmost and rightmost
// pixel of each row can be hull vertices, so the pixels are first reduced to a per-row (min x, max x)
// table (rows of a component are contiguous in the table: prefix sum over the components' row
// extents), then one wave per component builds the two monotone chains in LDS with exact integer
// orientation tests and walks the calipers.  The hull is emitted exactly as OpenCV's convexHull
// (clockwise=true) orders it — strict vertices, starting at the (min X, then min Y) point, towards
// increasing Y — because rotatingCalipers breaks area ties by visiting order.  Float arithmetic
// mirrors rotatingCalipers operation by operation; this file is built with -ffp-contract=off so
// the result equals oracle/cvgeom_oracle.c bit for bit.
#include <float.h>
#include <limits.h>
#include "common.h"

namespace {

constexpr int kMaxRows = 1024;     // rows of one component staged in LDS (56 KB in all)

struct MarP {
  int n, h, w, max_comps;
  double sx, sy;
  int rows_per_img;      // row-table entries per image (h*w for components, 3*h*w for hole borders)
};

// An atomic only where it can change the extent: a minimum only falls and a maximum only rises, so a
// (possibly stale) read that already covers v proves the update a no-op.  Without the test every run of a
// component queues up on the same two addresses (16 x 256^2 noisy maps: 141 -> 12 us for the row extents).
__device__ __forceinline__ void extent_min(int* e, int v) {
  if (v < __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(e, v);
}
__device__ __forceinline__ void extent_max(int* e, int v) {
  if (v > __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(e, v);
}

__global__ void mar_init_kernel(int* __restrict__ yext, size_t n_ext, int* __restrict__ rows, size_t n_rows) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (size_t k = i; k < n_ext; k += step) yext[k] = (k & 1) ? -1 : INT_MAX;      // (ymin, ymax) pairs
  for (size_t k = i; k < n_rows; k += step) rows[k] = (k & 1) ? -1 : INT_MAX;     // (xmin, xmax) pairs
}

__global__ void mar_yext_kernel(MarP p, const int* __restrict__ labels, int* __restrict__ yext) {
  const size_t hw = (size_t)p.h * p.w, total = (size_t)p.n * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int L = labels[i];
    if (L < 1 || L > p.max_comps) continue;
    const int img = (int)(i / hw), local = (int)(i % hw);
    const int y = local / p.w, x = local - y * p.w;
    if (x > 0 && labels[i - 1] == L) continue;        // one atomic pair per horizontal run, not per pixel
    int* e = yext + ((size_t)img * (p.max_comps + 1) + L) * 2;
    extent_min(e, y);
    extent_max(e + 1, y);
  }
}

// one workgroup per image: rowoff[L] = sum of the row extents of components 1..L-1
__global__ __launch_bounds__(1024) void mar_scan_kernel(MarP p, const int* __restrict__ yext,
                                                        int* __restrict__ rowoff) {
  __shared__ int wsum[16];
  __shared__ int s_carry;
  const int img = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int* e = yext + (size_t)img * (p.max_comps + 1) * 2;
  int* ro = rowoff + (size_t)img * (p.max_comps + 1);
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 1; base <= p.max_comps; base += 1024) {
    const int L = base + threadIdx.x;
    int ext = 0;
    if (L <= p.max_comps) {
      const int y0 = e[2 * L], y1 = e[2 * L + 1];
      ext = y1 >= y0 ? y1 - y0 + 1 : 0;
    }
    int v = ext;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int off = s_carry;
    for (int k = 0; k < wave; ++k) off += wsum[k];
    if (L <= p.max_comps) ro[L] = off + v - ext;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = off + v;
    __syncthreads();
  }
}

__global__ void mar_rows_kernel(MarP p, const int* __restrict__ labels, const int* __restrict__ yext,
                                const int* __restrict__ rowoff, int* __restrict__ rows) {
  const size_t hw = (size_t)p.h * p.w, total = (size_t)p.n * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int L = labels[i];
    if (L < 1 || L > p.max_comps) continue;
    const int img = (int)(i / hw), local = (int)(i % hw);
    const int y = local / p.w, x = local - y * p.w;
    const bool run_start = x == 0 || labels[i - 1] != L, run_end = x == p.w - 1 || labels[i + 1] != L;
    if (!run_start && !run_end) continue;             // only the ends of a horizontal run can be row extremes
    const size_t c = (size_t)img * (p.max_comps + 1) + L;
    const int r = rowoff[c] + y - yext[2 * c];
    int* e = rows + ((size_t)img * p.rows_per_img + r) * 2;
    if (run_start) extent_min(e, x);
    if (run_end) extent_max(e + 1, x);
  }
}

// ---- hole borders (cv2.findContours RETR_TREE inner contours, test.py:182) -------------------
// zlabels: 4-connected components of the 0-pixels.  A 0-region that touches the image edge is
// background (OpenCV pads the image with zeros), every other one is a hole whose border is the set
// of 1-pixels with a 4-neighbour inside it (Suzuki's border points for 8-connected foreground).
__global__ void mar_edge_kernel(MarP p, const int* __restrict__ zlabels, unsigned char* __restrict__ edge) {
  const int per = 2 * (p.h + p.w);
  const size_t total = (size_t)p.n * per;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int img = (int)(i / per), k = (int)(i % per);
    int x, y;
    if (k < p.w) { x = k; y = 0; }
    else if (k < 2 * p.w) { x = k - p.w; y = p.h - 1; }
    else if (k < 2 * p.w + p.h) { x = 0; y = k - 2 * p.w; }
    else { x = p.w - 1; y = k - 2 * p.w - p.h; }
    const int L = zlabels[((size_t)img * p.h + y) * p.w + x];
    if (L >= 1 && L <= p.max_comps) edge[(size_t)img * (p.max_comps + 1) + L] = 1;
  }
}

template <int PASS>   // 0: y extents, 1: row extremes
__global__ void mar_border_kernel(MarP p, const unsigned char* __restrict__ mask, const int* __restrict__ zlabels,
                                  const unsigned char* __restrict__ edge, int* __restrict__ yext,
                                  const int* __restrict__ rowoff, int* __restrict__ rows) {
  const size_t hw = (size_t)p.h * p.w, total = (size_t)p.n * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    if (!mask[i]) continue;
    const int img = (int)(i / hw), local = (int)(i % hw);
    const int y = local / p.w, x = local - y * p.w;
    const int* zl = zlabels + (size_t)img * hw;
    int nb[4] = {x > 0 ? zl[local - 1] : 0, x + 1 < p.w ? zl[local + 1] : 0, y > 0 ? zl[local - p.w] : 0,
                 y + 1 < p.h ? zl[local + p.w] : 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int L = nb[k];
      if (L < 1 || L > p.max_comps) continue;
      bool dup = false;
      for (int j = 0; j < k; ++j) dup |= nb[j] == L;
      if (dup) continue;
      const size_t c = (size_t)img * (p.max_comps + 1) + L;
      if (edge[c]) continue;
      if (PASS == 0) {
        extent_min(yext + 2 * c, y);
        extent_max(yext + 2 * c + 1, y);
      } else {
        int* e = rows + ((size_t)img * p.rows_per_img + rowoff[c] + y - yext[2 * c]) * 2;
        extent_min(e, x);
        extent_max(e + 1, x);
      }
    }
  }
}

__device__ __forceinline__ long long cross3(int ax, int ay, int bx, int by, int cx, int cy) {
  return (long long)(bx - ax) * (cy - ay) - (long long)(by - ay) * (cx - ax);
}

// one wave per (image, component)
__global__ __launch_bounds__(64) void mar_hull_calipers_kernel(
    MarP p, const int* __restrict__ ncomp, const int* __restrict__ yext, const int* __restrict__ rowoff,
    const int* __restrict__ rows, int* __restrict__ hull_n, int* __restrict__ hull_head,
    float* __restrict__ cal) {
  __shared__ int lx[kMaxRows], ly[kMaxRows], rx[kMaxRows], ry[kMaxRows];
  __shared__ float hx[2 * kMaxRows], hy[2 * kMaxRows];
  __shared__ float vx[2 * kMaxRows], vy[2 * kMaxRows], ivl[2 * kMaxRows];
  __shared__ int s_nl, s_nr, s_nh;
  const int img = blockIdx.y, L = blockIdx.x + 1;
  const int lane = threadIdx.x;
  int K = ncomp[img];
  if (K > p.max_comps) K = p.max_comps;
  const size_t c = (size_t)img * (p.max_comps + 1) + L;
  const size_t o = (size_t)img * p.max_comps + (L - 1);
  if (L > K) return;
  const int y0 = yext[2 * c], y1 = yext[2 * c + 1];
  const int R = y1 >= y0 ? y1 - y0 + 1 : 0;
  if (R == 0) {                        // label id without pixels
    if (lane == 0) hull_n[o] = 0;
    return;
  }
  const int* rw = rows + ((size_t)img * p.rows_per_img + rowoff[c]) * 2;
  // scaled candidates of every row (rows are distinct Y because sy >= 1)
  for (int r = lane; r < R; r += 64) {
    const int xl = rw[2 * r], xr = rw[2 * r + 1];
    const int Y = (int)((double)(y0 + r) * p.sy);
    lx[r] = xr >= xl ? (int)((double)xl * p.sx) : INT_MAX;     // INT_MAX: empty row
    rx[r] = xr >= xl ? (int)((double)xr * p.sx) : INT_MAX;
    ly[r] = Y;
    ry[r] = Y;
  }
  __syncthreads();
  if (lane == 0) {
    // left chain, top -> bottom, in place (stack never outruns the read index): keep b iff it lies
    // strictly left of a->c (cross < 0 with y pointing down)
    int nl = 0;
    for (int r = 0; r < R; ++r) {
      const int X = lx[r], Y = ly[r];
      if (X == INT_MAX) continue;
      while (nl >= 2 && cross3(lx[nl - 2], ly[nl - 2], lx[nl - 1], ly[nl - 1], X, Y) >= 0) --nl;
      lx[nl] = X;
      ly[nl] = Y;
      ++nl;
    }
    int nr = 0;
    for (int r = 0; r < R; ++r) {
      const int X = rx[r], Y = ry[r];
      if (X == INT_MAX) continue;
      while (nr >= 2 && cross3(rx[nr - 2], ry[nr - 2], rx[nr - 1], ry[nr - 1], X, Y) <= 0) --nr;
      rx[nr] = X;
      ry[nr] = Y;
      ++nr;
    }
    s_nl = nl;
    s_nr = nr;
  }
  __syncthreads();
  const int nl = s_nl, nr = s_nr;
  if (lane == 0) {
    // cycle: L_0..L_{nl-1}, then the right chain bottom -> top without the points it shares with L
    const bool share_bot = lx[nl - 1] == rx[nr - 1] && ly[nl - 1] == ry[nr - 1];
    const bool share_top = lx[0] == rx[0] && ly[0] == ry[0];
    const int r_hi = share_bot ? nr - 2 : nr - 1;
    const int r_lo = share_top ? 1 : 0;
    const int nrt = r_hi >= r_lo ? r_hi - r_lo + 1 : 0;
    int nh = nl + nrt;
    if (nl == 2 && nr == 2 && share_bot && share_top) nh = 2;            // collinear set
    // start = first point of L with the smallest X (Y ascends along L)
    int s = 0;
    for (int i = 1; i < nl; ++i)
      if (lx[i] < lx[s]) s = i;
    for (int i = 0; i < nh; ++i) {
      int k = s + i;
      if (k >= nh) k -= nh;
      int X, Y;
      if (k < nl) { X = lx[k]; Y = ly[k]; }
      else { const int q = r_hi - (k - nl); X = rx[q]; Y = ry[q]; }
      hx[i] = (float)X;
      hy[i] = (float)Y;
    }
    s_nh = nh;
    hull_n[o] = nh;
    hull_head[o * 4 + 0] = (int)hx[0];
    hull_head[o * 4 + 1] = (int)hy[0];
    hull_head[o * 4 + 2] = nh > 1 ? (int)hx[1] : 0;
    hull_head[o * 4 + 3] = nh > 1 ? (int)hy[1] : 0;
  }
  __syncthreads();
  const int n = s_nh;
  float* out = cal + o * 6;
  if (n <= 2) {
    if (lane < 6) out[lane] = 0.f;
    return;
  }
  // edge vectors and inverse lengths (rotatingCalipers' first loop), lanes in parallel
  for (int i = lane; i < n; i += 64) {
    const int j = i + 1 < n ? i + 1 : 0;
    const double dx = hx[j] - hx[i], dy = hy[j] - hy[i];
    vx[i] = (float)dx;
    vy[i] = (float)dy;
    ivl[i] = (float)(1. / sqrt(dx * dx + dy * dy));
  }
  __syncthreads();
  if (lane != 0) return;
  int left = 0, bottom = 0, right = 0, top = 0;
  {
    float left_x = hx[0], right_x = hx[0], top_y = hy[0], bottom_y = hy[0];
    for (int i = 0; i < n; ++i) {
      const float px = hx[i], py = hy[i];
      if (px < left_x) left_x = px, left = i;
      if (px > right_x) right_x = px, right = i;
      if (py > top_y) top_y = py, top = i;
      if (py < bottom_y) bottom_y = py, bottom = i;
    }
  }
  float orientation = 0.f;
  {
    double ax = vx[n - 1], ay = vy[n - 1];
    for (int i = 0; i < n; ++i) {
      const double bx = vx[i], by = vy[i];
      const double convexity = ax * by - ay * bx;
      if (convexity != 0) {
        orientation = convexity > 0 ? 1.f : -1.f;
        break;
      }
      ax = bx;
      ay = by;
    }
  }
  float base_a = orientation, base_b = 0.f;
  int seq[4] = {bottom, right, top, left};
  float minarea = FLT_MAX;
  int b_left = 0, b_bottom = 0;
  float b_a = 0.f, b_b = 0.f, b_width = 0.f, b_height = 0.f;
  for (int k = 0; k < n; ++k) {
    const float dp0 = +base_a * vx[seq[0]] + base_b * vy[seq[0]];
    const float dp1 = -base_b * vx[seq[1]] + base_a * vy[seq[1]];
    const float dp2 = -base_a * vx[seq[2]] - base_b * vy[seq[2]];
    const float dp3 = +base_b * vx[seq[3]] - base_a * vy[seq[3]];
    float maxcos = dp0 * ivl[seq[0]];
    int main_element = 0;
    float cs = dp1 * ivl[seq[1]];
    if (cs > maxcos) { main_element = 1; maxcos = cs; }
    cs = dp2 * ivl[seq[2]];
    if (cs > maxcos) { main_element = 2; maxcos = cs; }
    cs = dp3 * ivl[seq[3]];
    if (cs > maxcos) { main_element = 3; maxcos = cs; }
    const int pindex = main_element == 0 ? seq[0] : main_element == 1 ? seq[1] : main_element == 2 ? seq[2] : seq[3];
    const float lead_x = vx[pindex] * ivl[pindex];
    const float lead_y = vy[pindex] * ivl[pindex];
    int nxt = pindex + 1;
    if (nxt == n) nxt = 0;
    if (main_element == 0) { base_a = lead_x; base_b = lead_y; seq[0] = nxt; }
    else if (main_element == 1) { base_a = lead_y; base_b = -lead_x; seq[1] = nxt; }
    else if (main_element == 2) { base_a = -lead_x; base_b = -lead_y; seq[2] = nxt; }
    else { base_a = -lead_y; base_b = lead_x; seq[3] = nxt; }
    float dx = hx[seq[1]] - hx[seq[3]];
    float dy = hy[seq[1]] - hy[seq[3]];
    const float width = dx * base_a + dy * base_b;
    dx = hx[seq[2]] - hx[seq[0]];
    dy = hy[seq[2]] - hy[seq[0]];
    const float height = -dx * base_b + dy * base_a;
    const float area = width * height;
    if (area <= minarea) {
      minarea = area;
      b_left = seq[3];
      b_a = base_a;
      b_width = width;
      b_b = base_b;
      b_height = height;
      b_bottom = seq[0];
    }
  }
  const float A1 = b_a, B1 = b_b, A2 = -b_b, B2 = b_a;
  const float C1 = A1 * hx[b_left] + hy[b_left] * B1;
  const float C2 = A2 * hx[b_bottom] + hy[b_bottom] * B2;
  const float idet = 1.f / (A1 * B2 - A2 * B1);
  out[0] = (C1 * B2 - C2 * B1) * idet;
  out[1] = (A1 * C2 - A2 * C1) * idet;
  out[2] = A1 * b_width;
  out[3] = B1 * b_width;
  out[4] = A2 * b_height;
  out[5] = B2 * b_height;
}

unsigned bgrid(size_t items) {
  size_t b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" size_t ocr_min_area_rects_workspace(int n, int h, int w, int max_comps) {
  // (ymin,ymax) [n][max_comps+1][2], rowoff [n][max_comps+1], rows [n][h*w][2]
  return ((size_t)n * (max_comps + 1) * 3 + (size_t)n * h * w * 2) * sizeof(int);
}

extern "C" int ocr_min_area_rects(const void* labels_i32, const void* ncomp_i32, int n, int h, int w,
                                  int max_comps, double scale_x, double scale_y, void* hull_n_i32,
                                  void* hull_head_i32, void* calipers_f32, void* workspace,
                                  size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(labels_i32 && ncomp_i32 && hull_n_i32 && hull_head_i32 && calipers_f32 && workspace);
  OCR_CHECK_ARG(n > 0 && h > 0 && w > 0 && max_comps > 0);
  // strictly increasing integer maps (rows stay distinct) and LDS staging of one component's rows
  OCR_CHECK_SHAPE(scale_x >= 1.0 && scale_y >= 1.0 && h <= kMaxRows && max_comps <= 65535 && n <= 65535);
  OCR_CHECK_SHAPE((double)w * scale_x < 16777216.0 && (double)h * scale_y < 16777216.0);   // exact in f32
  if (ws_bytes < ocr_min_area_rects_workspace(n, h, w, max_comps)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MarP p{n, h, w, max_comps, scale_x, scale_y, h * w};
  int* yext = static_cast<int*>(workspace);
  const size_t n_ext = (size_t)n * (max_comps + 1) * 2;
  int* rowoff = yext + n_ext;
  int* rows = rowoff + (size_t)n * (max_comps + 1);
  const size_t n_rows = (size_t)n * h * w * 2;
  const size_t total = (size_t)n * h * w;
  hipLaunchKernelGGL(mar_init_kernel, dim3(bgrid(n_rows)), dim3(256), 0, st, yext, n_ext, rows, n_rows);
  hipLaunchKernelGGL(mar_yext_kernel, dim3(bgrid(total)), dim3(256), 0, st, p,
                     static_cast<const int*>(labels_i32), yext);
  hipLaunchKernelGGL(mar_scan_kernel, dim3(n), dim3(1024), 0, st, p, yext, rowoff);
  hipLaunchKernelGGL(mar_rows_kernel, dim3(bgrid(total)), dim3(256), 0, st, p,
                     static_cast<const int*>(labels_i32), yext, rowoff, rows);
  hipLaunchKernelGGL(mar_hull_calipers_kernel, dim3(max_comps, n), dim3(64), 0, st, p,
                     static_cast<const int*>(ncomp_i32), yext, rowoff, rows,
                     static_cast<int*>(hull_n_i32), static_cast<int*>(hull_head_i32),
                     static_cast<float*>(calipers_f32));
  return ocr_launch_status();
}

extern "C" size_t ocr_hole_border_rects_workspace(int n, int h, int w, int max_regions) {
  // (ymin,ymax), rowoff per region; rows [n][3*h*w][2] (a hole of e rows has a border of e + 2 rows
  // and at least e pixels); edge flags
  return ((size_t)n * (max_regions + 1) * 3 + (size_t)n * h * w * 3 * 2) * sizeof(int) +
         (((size_t)n * (max_regions + 1) + 15) & ~(size_t)15);
}

extern "C" int ocr_hole_border_rects(const void* mask_u8, const void* zlabels_i32, const void* nregions_i32, int n,
                                     int h, int w, int max_regions, double scale_x, double scale_y,
                                     void* hull_n_i32, void* hull_head_i32, void* calipers_f32, void* workspace,
                                     size_t ws_bytes, void* stream) {
  OCR_CHECK_ARG(mask_u8 && zlabels_i32 && nregions_i32 && hull_n_i32 && hull_head_i32 && calipers_f32 && workspace);
  OCR_CHECK_ARG(n > 0 && h > 0 && w > 0 && max_regions > 0);
  OCR_CHECK_SHAPE(scale_x >= 1.0 && scale_y >= 1.0 && h <= kMaxRows && max_regions <= 65535 && n <= 65535);
  OCR_CHECK_SHAPE((double)w * scale_x < 16777216.0 && (double)h * scale_y < 16777216.0);
  if (ws_bytes < ocr_hole_border_rects_workspace(n, h, w, max_regions)) return OCR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MarP p{n, h, w, max_regions, scale_x, scale_y, 3 * h * w};
  int* yext = static_cast<int*>(workspace);
  const size_t n_ext = (size_t)n * (max_regions + 1) * 2;
  int* rowoff = yext + n_ext;
  int* rows = rowoff + (size_t)n * (max_regions + 1);
  const size_t n_rows = (size_t)n * p.rows_per_img * 2;
  unsigned char* edge = reinterpret_cast<unsigned char*>(rows + n_rows);
  const size_t total = (size_t)n * h * w;
  const unsigned char* mask = static_cast<const unsigned char*>(mask_u8);
  const int* zl = static_cast<const int*>(zlabels_i32);
  if (hipMemsetAsync(edge, 0, (size_t)n * (max_regions + 1), st) != hipSuccess) return OCR_ERR_HIP;
  hipLaunchKernelGGL(mar_init_kernel, dim3(bgrid(n_rows)), dim3(256), 0, st, yext, n_ext, rows, n_rows);
  hipLaunchKernelGGL(mar_edge_kernel, dim3(bgrid((size_t)n * 2 * (h + w))), dim3(256), 0, st, p, zl, edge);
  hipLaunchKernelGGL(mar_border_kernel<0>, dim3(bgrid(total)), dim3(256), 0, st, p, mask, zl, edge, yext, rowoff, rows);
  hipLaunchKernelGGL(mar_scan_kernel, dim3(n), dim3(1024), 0, st, p, yext, rowoff);
  hipLaunchKernelGGL(mar_border_kernel<1>, dim3(bgrid(total)), dim3(256), 0, st, p, mask, zl, edge, yext, rowoff, rows);
  hipLaunchKernelGGL(mar_hull_calipers_kernel, dim3(max_regions, n), dim3(64), 0, st, p,
                     static_cast<const int*>(nregions_i32), yext, rowoff, rows, static_cast<int*>(hull_n_i32),
                     static_cast<int*>(hull_head_i32), static_cast<float*>(calipers_f32));
  return ocr_launch_status();
}
