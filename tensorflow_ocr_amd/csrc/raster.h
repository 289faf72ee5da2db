// cv2.fillPoly's raster in closed form, per pixel (shared by the label kernels and the evaluation
// IoU kernel).  OpenCV draws a polygon as (a) every edge as an 8-connected Bresenham line walked from
// its LEFT end after clipLine (CollectPolyEdges -> Line -> LineIterator(left_to_right)) and (b) an
// even-odd scanline fill between sorted 16.16 fixed-point edge crossings (FillEdgeCollection).
// Restated literally in oracle/cvgeom_oracle.c; here as order-free predicates:
//   outline  after j major-axis steps the minor offset is floor((2*minor*j + major - 1) / (2*major))
//   interior the edges active on row y (y0 <= y < y1) cross it at x_e = x_top + (y - y0) * dx,
//            dx = ((x1 - x0) << 16) / (y1 - y0) truncated; OpenCV fills [ceil(xs[2k]), floor(xs[2k+1])]
//            of the sorted crossings  <=>  (some x_e == x << 16) or (#{x_e < x << 16} is odd)
#pragma once
#include <limits.h>
#include "common.h"

namespace raster {

struct Seg {      // clipped outline segment, start = left end
  int x1, y1, major, minor, flags;     // flags: 1 ok, 2 steep, 4 y decreasing
};
struct FillEdge {
  int y0, y1;
  long long x, dx;
};

__device__ __forceinline__ bool clip_line(long long width, long long height, long long& x1, long long& y1,
                                          long long& x2, long long& y2) {
  const long long right = width - 1, bottom = height - 1;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

// edge from vertex a to vertex b of a polygon drawn into a w x h image
__device__ __forceinline__ void setup_edge(long long ax, long long ay, long long bx, long long by, int w, int h,
                                           Seg& sg, FillEdge& fe) {
  sg = Seg{0, 0, 0, 0, 0};
  fe = FillEdge{0, 0, 0, 0};     // y0 == y1: never active
  long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
  if (clip_line(w, h, x1, y1, x2, y2)) {
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int sx1 = (int)x1, sy1 = (int)y1;
    if (dx < 0) { dx = -dx; dy = -dy; sx1 = (int)x2; sy1 = (int)y2; }
    int fl = 1;
    if (dy < 0) { dy = -dy; fl |= 4; }
    if (dy > dx) { fl |= 2; sg.major = dy; sg.minor = dx; }
    else { sg.major = dx; sg.minor = dy; }
    sg.x1 = sx1;
    sg.y1 = sy1;
    sg.flags = fl;
  }
  if (ay != by) {
    const long long fax = ax << 16, fbx = bx << 16;
    if (ay < by) { fe.y0 = (int)ay; fe.y1 = (int)by; fe.x = fax; }
    else { fe.y0 = (int)by; fe.y1 = (int)ay; fe.x = fbx; }
    fe.dx = (fbx - fax) / (by - ay);
  }
}

// FillEdgeCollection's early outs: fewer than two non-horizontal edges, or entirely outside
__device__ __forceinline__ bool fill_enabled(const FillEdge* e, int V, int w, int h) {
  int total = 0, y_min = INT_MAX, y_max = INT_MIN;
  long long x_min = LLONG_MAX, x_max = -1;
  for (int k = 0; k < V; ++k) {
    const FillEdge fe = e[k];
    if (fe.y0 == fe.y1) continue;
    ++total;
    const long long xe = fe.x + (long long)(fe.y1 - fe.y0) * fe.dx;
    y_min = min(y_min, fe.y0);
    y_max = max(y_max, fe.y1);
    x_min = min(x_min, min(fe.x, xe));
    x_max = max(x_max, max(fe.x, xe));
  }
  return total >= 2 && !(y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((long long)w << 16));
}

// is pixel (x, y) in the polygon's raster?
__device__ __forceinline__ bool covers(const Seg* segs, const FillEdge* edges, int V, bool fill, int x, int y) {
  const long long X = (long long)x << 16;
  bool in = false, eq = false;
  int below = 0;
  for (int e = 0; e < V; ++e) {
    const Seg sg = segs[e];
    if (sg.flags & 1) {
      const int ys = (sg.flags & 4) ? -1 : 1;
      if (sg.flags & 2) {           // steep: major axis y
        const int j = (y - sg.y1) * ys;
        if (j >= 0 && j <= sg.major) {
          const int m = (int)((2ll * sg.minor * j + sg.major - 1) / (2ll * sg.major));
          in |= x == sg.x1 + m;
        }
      } else {
        const int j = x - sg.x1;
        if (j >= 0 && j <= sg.major) {
          const int m = sg.major > 0 ? (int)((2ll * sg.minor * j + sg.major - 1) / (2ll * sg.major)) : 0;
          in |= y == sg.y1 + ys * m;
        }
      }
    }
    const FillEdge fe = edges[e];
    if (fill && fe.y0 <= y && y < fe.y1) {
      const long long xe = fe.x + (long long)(y - fe.y0) * fe.dx;
      below += xe < X;
      eq |= xe == X;
    }
  }
  return in || eq || (below & 1);
}

}  // namespace raster
