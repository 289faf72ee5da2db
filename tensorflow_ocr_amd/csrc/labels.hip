// Ground-truth label maps on the GPU: the two generators of the reference,
//   datasets/icdar.py:486-539      generate_rbox  (score / 8 link maps / training mask, then [::4,::4])
//   tool/pixellink_fn.py:53-110    generate_rbox  (score / 8 link maps at 1/4 resolution)
// both built on cv2.fillPoly of the text quadrangles.
//
//   ocr_poly_cover        per full-resolution pixel: smallest and largest index of the polygons whose
//                         cv2.fillPoly raster contains it, and whether an ignored polygon covers it
//   ocr_icdar_labels      icdar.generate_rbox + the generator's subsample, from the cover map
//   ocr_pixellink_labels  pixellink_fn.generate_rbox, from the cover map
//
// The reference draws the polygons one after another and evaluates `valid_link` between draws; all
// of it is a function of (first, last) covering index per pixel:
//   score_map        = covered by any polygon;  training_mask = not covered by an ignored polygon
//   poly_mask        = last covering index + 1
//   icdar link c     = border rule, else "neighbour already drawn when MY last polygon was drawn"
//                      = first(neighbour) <= last(me)        (transposed directions, -1 wraps)
//   pixellink link c = border rule, else last(neighbour) == last(me)
// so every pixel is independent: no atomics, no ordering, bitwise reproducible.
//
// fillPoly's raster (OpenCV CollectPolyEdges + FillEdgeCollection, restated literally in
// oracle/cvgeom_oracle.c) in closed form per pixel (x, y), X = x << 16:
//   * outline: each edge is an 8-connected Bresenham line walked from its LEFT end (LineIterator,
//     left_to_right) after clipLine; after j major-axis steps the minor offset is
//     floor((2*minor*j + major - 1) / (2*major))
//   * interior: the edges active on row y (y0 <= y < y1) cross it at x_e = x_top + (y - y0) * dx,
//     dx = ((x1 - x0) << 16) / (y1 - y0) truncated; OpenCV fills [ceil(xs[2k]), floor(xs[2k+1])] of
//     the sorted crossings, which is  (some x_e == X)  or  (#{x_e < X} is odd).
#include "raster.h"

namespace {

constexpr int kChunk = 32;     // polygons staged per pass
constexpr int kMaxV = 8;       // vertices per polygon
constexpr int kTileW = 64, kTileH = 4;

struct CoverP {
  int n, P, V, h, w;
};

using raster::FillEdge;
using raster::Seg;

// grid (tiles_x, tiles_y, n), 256 threads = one 64 x 4 pixel tile
__global__ __launch_bounds__(256) void poly_cover_kernel(CoverP p, const int* __restrict__ polys,
                                                         const int* __restrict__ counts,
                                                         const unsigned char* __restrict__ ignore,
                                                         unsigned* __restrict__ cover) {
  __shared__ Seg s_seg[kChunk * kMaxV];
  __shared__ FillEdge s_edge[kChunk * kMaxV];
  __shared__ int s_fill[kChunk];          // interior fill enabled
  __shared__ int s_ign[kChunk];
  __shared__ int s_hit[kChunk];
  __shared__ int s_nhit;
  const int img = blockIdx.z;
  const int tx0 = blockIdx.x * kTileW, ty0 = blockIdx.y * kTileH;
  const int x = tx0 + (threadIdx.x & 63), y = ty0 + (threadIdx.x >> 6);
  int count = counts[img];
  if (count > p.P) count = p.P;
  unsigned mn = 0, mx = 0, ign = 0;
  for (int base = 0; base < count; base += kChunk) {
    __syncthreads();       // previous chunk fully consumed
    {
      // edge setup: thread -> (polygon base + t/8, edge t%8); edge e runs from vertex e-1 to vertex e
      const int pl = threadIdx.x >> 3, e = threadIdx.x & 7;
      const int pi = base + pl;
      Seg sg{0, 0, 0, 0, 0};
      FillEdge fe{0, 0, 0, 0};
      if (pi < count && e < p.V) {
        const int* v = polys + (((size_t)img * p.P + pi) * p.V) * 2;
        const int e0 = e == 0 ? p.V - 1 : e - 1;
        raster::setup_edge(v[2 * e0], v[2 * e0 + 1], v[2 * e], v[2 * e + 1], p.w, p.h, sg, fe);
      }
      s_seg[threadIdx.x] = sg;
      s_edge[threadIdx.x] = fe;
    }
    __syncthreads();
    if (threadIdx.x < kChunk) {
      const int pi = base + threadIdx.x;
      int hit = 0;
      if (pi < count) {
        const int* v = polys + (((size_t)img * p.P + pi) * p.V) * 2;
        int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
        for (int k = 0; k < p.V; ++k) {
          bx0 = min(bx0, v[2 * k]); bx1 = max(bx1, v[2 * k]);
          by0 = min(by0, v[2 * k + 1]); by1 = max(by1, v[2 * k + 1]);
        }
        s_fill[threadIdx.x] = raster::fill_enabled(&s_edge[threadIdx.x * kMaxV], p.V, p.w, p.h) ? 1 : 0;
        s_ign[threadIdx.x] = ignore[(size_t)img * p.P + pi] ? 1 : 0;
        // tile culling on the vertex bounding box (+1 px slack; the exact test follows per pixel)
        hit = !(bx1 + 1 < tx0 || bx0 - 1 > tx0 + kTileW - 1 || by1 + 1 < ty0 || by0 - 1 > ty0 + kTileH - 1);
      }
      const unsigned long long m = __ballot(hit);       // wave 0, lanes 0..31
      if (hit) s_hit[__popcll(m & ((1ull << threadIdx.x) - 1))] = threadIdx.x;
      if (threadIdx.x == 0) s_nhit = __popcll(m);
    }
    __syncthreads();
    const int nhit = s_nhit;
    for (int q = 0; q < nhit; ++q) {
      const int pl = s_hit[q];
      const bool in = raster::covers(&s_seg[pl * kMaxV], &s_edge[pl * kMaxV], p.V, s_fill[pl] != 0, x, y);
      if (in) {
        const unsigned id = (unsigned)(base + pl + 1);
        if (mn == 0) mn = id;
        mx = id;
        ign |= (unsigned)s_ign[pl];
      }
    }
  }
  if (x < p.w && y < p.h) cover[((size_t)img * p.h + y) * p.w + x] = mn | (mx << 8) | (ign << 16);
}

__device__ __constant__ int kIcdarDx[8] = {0, 1, -1, 0, 1, -1, -1, 1};
__device__ __constant__ int kIcdarDy[8] = {-1, -1, -1, 1, 1, 1, 0, 0};
__device__ __constant__ int kPlDx[8] = {-1, -1, -1, 1, 1, 1, 0, 0};
__device__ __constant__ int kPlDy[8] = {0, 1, -1, 0, 1, -1, -1, 1};

// one thread per (output pixel, channel 0..7); h == w (checked by the caller)
__global__ void icdar_labels_kernel(const unsigned* __restrict__ cover, int n, int h, int w, int step, int oh,
                                    int ow, float* __restrict__ score, float* __restrict__ geo,
                                    float* __restrict__ mask) {
  const size_t total = (size_t)n * oh * ow * 8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i & 7);
    const size_t px = i >> 3;
    const int ox = (int)(px % ow), oy = (int)((px / ow) % oh), img = (int)(px / ((size_t)ow * oh));
    const int x = ox * step, y = oy * step;
    const unsigned* cv = cover + (size_t)img * h * w;
    const unsigned me = cv[(size_t)y * w + x];
    const unsigned last = (me >> 8) & 255u;
    float v = 0.f;
    if (last) {
      if (x == h - 1 || y == w - 1) {
        v = 1.f;
      } else {
        int nx = x + kIcdarDx[c], ny = y + kIcdarDy[c];
        if (nx < 0) nx += w;                      // numpy index -1
        if (ny < 0) ny += h;
        const unsigned first_nb = cv[(size_t)ny * w + nx] & 255u;
        v = (first_nb != 0 && first_nb <= last) ? 1.f : 0.f;
      }
    }
    geo[i] = v;
    if (c == 0) {
      score[px] = (me & 255u) ? 1.f : 0.f;
      mask[px] = (me >> 16) & 1u ? 0.f : 1.f;
    }
  }
}

__global__ void pixellink_labels_kernel(const unsigned* __restrict__ cover, int n, int h, int w, int nh, int nw,
                                        double ifx, double ify, float* __restrict__ score,
                                        float* __restrict__ link) {
  const size_t total = (size_t)n * nh * nw * 8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i & 7);
    const size_t px = i >> 3;
    const int ox = (int)(px % nw), oy = (int)((px / nw) % nh), img = (int)(px / ((size_t)nw * nh));
    const unsigned* cv = cover + (size_t)img * h * w;
    // cv2.resize INTER_NEAREST source index
    const int sx = min((int)floor(ox * ifx), w - 1), sy = min((int)floor(oy * ify), h - 1);
    const unsigned me = cv[(size_t)sy * w + sx];
    const unsigned lab = (me >> 8) & 255u;
    float v = 0.f;
    if (lab) {
      if (ox == nw - 1 || oy == nh - 1 || ox == 0 || oy == 0) {
        v = 1.f;
      } else {
        const int qx = ox + kPlDx[c], qy = oy + kPlDy[c];
        const int tx = min((int)floor(qx * ifx), w - 1), ty = min((int)floor(qy * ify), h - 1);
        v = ((cv[(size_t)ty * w + tx] >> 8) & 255u) == lab ? 1.f : 0.f;
      }
    }
    link[i] = v;
    if (c == 0) score[px] = (me & 255u) ? 1.f : 0.f;
  }
}

unsigned lgrid(size_t items) {
  size_t b = (items + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int ocr_poly_cover(const void* polys_i32, const void* counts_i32, const void* ignore_u8, int n,
                              int max_polys, int verts, int h, int w, void* cover_u32, void* stream) {
  OCR_CHECK_ARG(polys_i32 && counts_i32 && ignore_u8 && cover_u32 && n > 0 && h > 0 && w > 0);
  // poly_mask is a uint8 image in the reference: index + 1 must stay below the saturation value
  OCR_CHECK_SHAPE(max_polys >= 1 && max_polys <= 254 && verts >= 3 && verts <= kMaxV && n <= 65535);
  OCR_CHECK_SHAPE(h <= 16384 && w <= 16384);
  CoverP p{n, max_polys, verts, h, w};
  hipLaunchKernelGGL(poly_cover_kernel, dim3(ocr_cdiv(w, kTileW), ocr_cdiv(h, kTileH), n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, static_cast<const int*>(polys_i32),
                     static_cast<const int*>(counts_i32), static_cast<const unsigned char*>(ignore_u8),
                     static_cast<unsigned*>(cover_u32));
  return ocr_launch_status();
}

extern "C" int ocr_icdar_labels(const void* cover_u32, int n, int h, int w, int step, void* score_f32,
                                void* geo_f32, void* mask_f32, void* stream) {
  OCR_CHECK_ARG(cover_u32 && score_f32 && geo_f32 && mask_f32 && n > 0 && h > 0 && w > 0 && step > 0);
  OCR_CHECK_SHAPE(h == w);       // the reference's transposed border rule indexes out of range otherwise
  const int oh = ocr_cdiv(h, step), ow = ocr_cdiv(w, step);
  hipLaunchKernelGGL(icdar_labels_kernel, dim3(lgrid((size_t)n * oh * ow * 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned*>(cover_u32), n, h, w, step,
                     oh, ow, static_cast<float*>(score_f32), static_cast<float*>(geo_f32),
                     static_cast<float*>(mask_f32));
  return ocr_launch_status();
}

extern "C" int ocr_pixellink_labels(const void* cover_u32, int n, int h, int w, int new_h, int new_w,
                                    void* score_f32, void* link_f32, void* stream) {
  OCR_CHECK_ARG(cover_u32 && score_f32 && link_f32 && n > 0 && h > 0 && w > 0 && new_h > 0 && new_w > 0);
  const double ifx = 1.0 / ((double)new_w / (double)w), ify = 1.0 / ((double)new_h / (double)h);
  hipLaunchKernelGGL(pixellink_labels_kernel, dim3(lgrid((size_t)n * new_h * new_w * 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned*>(cover_u32), n, h, w, new_h,
                     new_w, ifx, ify, static_cast<float*>(score_f32), static_cast<float*>(link_f32));
  return ocr_launch_status();
}
