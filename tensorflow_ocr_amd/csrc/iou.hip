// Rasterised IoU of detected quadrangles against ground-truth quadrangles — the inner loop of the
// reference's evaluation (tool/bboxes.py: np_bboxes_jaccard :246-283 called per detection from
// bboxes_matching :171-240): both polygons are drawn filled into 0/1 masks
// (cv2.drawContours(..., thickness=-1): the same CollectPolyEdges + FillEdgeCollection raster as
// cv2.fillPoly) and IoU = sum(a*b) / sum(a+b >= 1).
//
//   ocr_quad_iou   dets int32 [nd][V][2], gts int32 [ng][V][2] -> inter, uni int32 [nd][ng]
//
// One workgroup per (gt, det) pair walks the pair's bounding box with the closed-form raster
// predicate of raster.h (no masks are materialised); integer counts out, so the result is bit-exact
// and the float division stays where the reference does it (NumPy, on the host).
#include "raster.h"

namespace {

constexpr int kMaxV = 8;

__global__ __launch_bounds__(256) void quad_iou_kernel(const int* __restrict__ dets, const int* __restrict__ gts,
                                                       int nd, int ng, int V, int mask_h, int mask_w,
                                                       int* __restrict__ inter, int* __restrict__ uni) {
  __shared__ raster::Seg s_seg[2][kMaxV];
  __shared__ raster::FillEdge s_edge[2][kMaxV];
  __shared__ int s_fill[2];
  __shared__ int s_box[4];
  __shared__ int s_cnt[2][4];
  const int gi = blockIdx.x, di = blockIdx.y;
  const int* pv[2] = {dets + (size_t)di * V * 2, gts + (size_t)gi * V * 2};
  if (threadIdx.x < 2 * kMaxV) {
    const int which = threadIdx.x / kMaxV, e = threadIdx.x % kMaxV;
    raster::Seg sg{0, 0, 0, 0, 0};
    raster::FillEdge fe{0, 0, 0, 0};
    if (e < V) {
      const int* v = pv[which];
      const int e0 = e == 0 ? V - 1 : e - 1;
      raster::setup_edge(v[2 * e0], v[2 * e0 + 1], v[2 * e], v[2 * e + 1], mask_w, mask_h, sg, fe);
    }
    s_seg[which][e] = sg;
    s_edge[which][e] = fe;
  }
  __syncthreads();
  if (threadIdx.x < 2) s_fill[threadIdx.x] = raster::fill_enabled(s_edge[threadIdx.x], V, mask_w, mask_h) ? 1 : 0;
  if (threadIdx.x == 2) {
    int x0 = INT_MAX, y0 = INT_MAX, x1 = INT_MIN, y1 = INT_MIN;
    for (int w = 0; w < 2; ++w)
      for (int k = 0; k < V; ++k) {
        x0 = min(x0, pv[w][2 * k]); x1 = max(x1, pv[w][2 * k]);
        y0 = min(y0, pv[w][2 * k + 1]); y1 = max(y1, pv[w][2 * k + 1]);
      }
    s_box[0] = max(x0 - 1, 0);
    s_box[1] = max(y0 - 1, 0);
    s_box[2] = min(x1 + 1, mask_w - 1);
    s_box[3] = min(y1 + 1, mask_h - 1);
  }
  __syncthreads();
  const int bx0 = s_box[0], by0 = s_box[1], bw = s_box[2] - bx0 + 1, bh = s_box[3] - by0 + 1;
  int ci = 0, cu = 0;
  if (bw > 0 && bh > 0) {
    const long long total = (long long)bw * bh;
    for (long long i = threadIdx.x; i < total; i += 256) {
      const int y = by0 + (int)(i / bw), x = bx0 + (int)(i % bw);
      const bool a = raster::covers(s_seg[0], s_edge[0], V, s_fill[0] != 0, x, y);
      const bool b = raster::covers(s_seg[1], s_edge[1], V, s_fill[1] != 0, x, y);
      ci += a && b;
      cu += a || b;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ci += __shfl_xor(ci, o, 64);
    cu += __shfl_xor(cu, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_cnt[0][threadIdx.x >> 6] = ci;
    s_cnt[1][threadIdx.x >> 6] = cu;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    inter[(size_t)di * ng + gi] = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
    uni[(size_t)di * ng + gi] = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
  }
}

}  // namespace

extern "C" int ocr_quad_iou(const void* dets_i32, int nd, const void* gts_i32, int ng, int verts, int mask_h,
                            int mask_w, void* inter_i32, void* union_i32, void* stream) {
  OCR_CHECK_ARG(dets_i32 && gts_i32 && inter_i32 && union_i32 && nd > 0 && ng > 0 && mask_h > 0 && mask_w > 0);
  OCR_CHECK_SHAPE(verts >= 3 && verts <= kMaxV && nd <= 65535 && mask_h <= 32768 && mask_w <= 32768);
  hipLaunchKernelGGL(quad_iou_kernel, dim3(ng, nd), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const int*>(dets_i32), static_cast<const int*>(gts_i32), nd, ng, verts, mask_h,
                     mask_w, static_cast<int*>(inter_i32), static_cast<int*>(union_i32));
  return ocr_launch_status();
}
