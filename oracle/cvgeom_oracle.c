/* ORACLE (test infrastructure only; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline may call this).
 *
 * Plain-C restatement of the OpenCV geometry the reference calls through `cv2`, a third-party
 * dependency that is NOT under /root/reference and not installed here (no requirements file pins it;
 * the reference is Python-2 / TF-1.4 era, i.e. OpenCV 3.x — the 3.4 algorithms are restated):
 *
 *   cv2.minAreaRect + cv2.boxPoints   test_pixellink_fast.py:199-200, test_pixellink.py:213-214,
 *                                     test.py:188-190
 *       = convexHull (Sklansky on the (x,y)-sorted points, clockwise output) + rotatingCalipers
 *         (float32) + RotatedRect::points
 *   cv2.fillPoly                      datasets/icdar.py:507-515, tool/pixellink_fn.py
 *       = per edge an 8-connected Bresenham line (LineIterator) + even-odd scanline fill with
 *         16.16 fixed-point edges (CollectPolyEdges / FillEdgeCollection)
 *   cv2.resize(INTER_LINEAR, 8UC3)    datasets/icdar.py:615
 *       = fixed-point (11-bit coefficient) separable bilinear, half-pixel centres
 *
 * PARITY UNPINNED against cv2 itself (it cannot run here and the reference holds no vectors); the
 * HIP kernels are pinned against THIS file bit for bit (float32 paths: build with
 * -ffp-contract=off).  Known-answer cases in tests/test_oracle_cvgeom.py pin this file against
 * brute force (minimum-area over all hull edges, point-in-polygon fills, float bilinear).
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ *
 * convexHull(points int32 [n][2], clockwise=true, returnPoints=true)
 * ------------------------------------------------------------------------------------------ */
typedef struct { int x, y; } ipt;

static int cmp_pts(const void* a, const void* b) {
  const ipt* p = *(const ipt* const*)a;
  const ipt* q = *(const ipt* const*)b;
  if (p->x != q->x) return p->x < q->x ? -1 : 1;
  if (p->y != q->y) return p->y < q->y ? -1 : 1;
  return p < q ? -1 : (p > q ? 1 : 0);      /* OpenCV: pointer order breaks the tie */
}

static int sgn(long long v) { return (v > 0) - (v < 0); }

static int sklansky(ipt** array, int start, int end, int* stack, int nsign, int sign2) {
  int incr = end > start ? 1 : -1;
  int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
  int stacksize = 3;
  if (start == end || (array[start]->x == array[end]->x && array[start]->y == array[end]->y)) {
    stack[0] = start;
    return 1;
  }
  stack[0] = pprev;
  stack[1] = pcur;
  stack[2] = pnext;
  end += incr;
  while (pnext != end) {
    const long long cury = array[pcur]->y, nexty = array[pnext]->y;
    const long long by = nexty - cury;
    if (sgn(by) != nsign) {
      const long long ax = (long long)array[pcur]->x - array[pprev]->x;
      const long long bx = (long long)array[pnext]->x - array[pcur]->x;
      const long long ay = cury - array[pprev]->y;
      const long long convexity = ay * bx - ax * by;
      if (sgn(convexity) == sign2 && (ax != 0 || ay != 0)) {
        pprev = pcur;
        pcur = pnext;
        pnext += incr;
        stack[stacksize] = pnext;
        stacksize++;
      } else {
        if (pprev == start) {
          pcur = pnext;
          stack[1] = pcur;
          pnext += incr;
          stack[2] = pnext;
        } else {
          stack[stacksize - 2] = pnext;
          pcur = pprev;
          pprev = stack[stacksize - 4];
          stacksize--;
        }
      }
    } else {
      pnext += incr;
      stack[stacksize - 1] = pnext;
    }
  }
  return --stacksize;
}

/* returns the number of hull points written to hull_xy [<= n][2] */
int cvgeom_convex_hull(const int* pts_xy, int n, int* hull_xy) {
  if (n <= 0) return 0;
  const ipt* data = (const ipt*)pts_xy;
  ipt** pointer = (ipt**)malloc(sizeof(ipt*) * (size_t)n);
  int* stack = (int*)malloc(sizeof(int) * (size_t)(n + 2) * 2);
  int nout = 0, i;
  for (i = 0; i < n; ++i) pointer[i] = (ipt*)&data[i];
  qsort(pointer, (size_t)n, sizeof(ipt*), cmp_pts);
  int miny_ind = 0, maxy_ind = 0;
  for (i = 1; i < n; ++i) {
    const int y = pointer[i]->y;
    if (pointer[miny_ind]->y > y) miny_ind = i;
    if (pointer[maxy_ind]->y < y) maxy_ind = i;
  }
  if (pointer[0]->x == pointer[n - 1]->x && pointer[0]->y == pointer[n - 1]->y) {
    hull_xy[0] = pointer[0]->x;
    hull_xy[1] = pointer[0]->y;
    nout = 1;
  } else {
    const int clockwise = 1;
    /* upper half */
    int* tl_stack = stack;
    int tl_count = sklansky(pointer, 0, maxy_ind, tl_stack, -1, 1);
    int* tr_stack = stack + tl_count;
    int tr_count = sklansky(pointer, n - 1, maxy_ind, tr_stack, -1, -1);
    if (!clockwise) {
      int* t = tl_stack; tl_stack = tr_stack; tr_stack = t;
      int c = tl_count; tl_count = tr_count; tr_count = c;
    }
    int* hull_idx = (int*)malloc(sizeof(int) * (size_t)(n + 2) * 2);
    for (i = 0; i < tl_count - 1; ++i) hull_idx[nout++] = tl_stack[i];
    for (i = tr_count - 1; i > 0; --i) hull_idx[nout++] = tr_stack[i];
    const int stop_idx = tr_count > 2 ? tr_stack[1] : (tl_count > 2 ? tl_stack[tl_count - 2] : -1);
    /* the stacks are about to be overwritten: resolve stop_idx's coordinates now */
    /* lower half */
    int* bl_stack = stack;
    int bl_count = sklansky(pointer, 0, miny_ind, bl_stack, 1, -1);
    int* br_stack = stack + bl_count;
    int br_count = sklansky(pointer, n - 1, miny_ind, br_stack, 1, 1);
    if (clockwise) {
      int* t = bl_stack; bl_stack = br_stack; br_stack = t;
      int c = bl_count; bl_count = br_count; br_count = c;
    }
    if (stop_idx >= 0) {
      const int check_idx = bl_count > 2 ? bl_stack[1]
                            : (bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1);
      if (check_idx == stop_idx ||
          (check_idx >= 0 && pointer[check_idx]->x == pointer[stop_idx]->x &&
           pointer[check_idx]->y == pointer[stop_idx]->y)) {
        /* all points on one line: the bottom part mirrors the top part */
        if (bl_count > 2) bl_count = 2;
        if (br_count > 2) br_count = 2;
      }
    }
    for (i = 0; i < bl_count - 1; ++i) hull_idx[nout++] = bl_stack[i];
    for (i = br_count - 1; i > 0; --i) hull_idx[nout++] = br_stack[i];
    for (i = 0; i < nout; ++i) {
      hull_xy[2 * i] = pointer[hull_idx[i]]->x;
      hull_xy[2 * i + 1] = pointer[hull_idx[i]]->y;
    }
    free(hull_idx);
  }
  free(stack);
  free(pointer);
  return nout;
}

/* ------------------------------------------------------------------------------------------ *
 * rotatingCalipers(points f32 [n][2] (a convex polygon), CALIPERS_MINAREARECT) -> out[6] =
 * (corner x, corner y, edge-1 vector, edge-2 vector)
 * ------------------------------------------------------------------------------------------ */
void cvgeom_rotating_calipers(const float* pts, int n, float* out) {
  float minarea = FLT_MAX;
  int b_left = 0, b_bottom = 0;
  float b_a = 0.f, b_width = 0.f, b_b = 0.f, b_height = 0.f;
  int i, k;
  float* inv_vect_length = (float*)malloc(sizeof(float) * (size_t)n * 3);
  float* vect = inv_vect_length + n;     /* [n][2] */
  int left = 0, bottom = 0, right = 0, top = 0;
  int seq[4] = {-1, -1, -1, -1};
  float orientation = 0.f;
  float base_a, base_b = 0.f;
  float left_x, right_x, top_y, bottom_y;
  float p0x = pts[0], p0y = pts[1];
  left_x = right_x = p0x;
  top_y = bottom_y = p0y;
  for (i = 0; i < n; ++i) {
    double dx, dy;
    if (p0x < left_x) left_x = p0x, left = i;
    if (p0x > right_x) right_x = p0x, right = i;
    if (p0y > top_y) top_y = p0y, top = i;
    if (p0y < bottom_y) bottom_y = p0y, bottom = i;
    const int j = (i + 1 < n) ? i + 1 : 0;
    const float px = pts[2 * j], py = pts[2 * j + 1];
    dx = px - p0x;
    dy = py - p0y;
    vect[2 * i] = (float)dx;
    vect[2 * i + 1] = (float)dy;
    inv_vect_length[i] = (float)(1. / sqrt(dx * dx + dy * dy));
    p0x = px;
    p0y = py;
  }
  {
    double ax = vect[2 * (n - 1)], ay = vect[2 * (n - 1) + 1];
    for (i = 0; i < n; ++i) {
      const double bx = vect[2 * i], by = vect[2 * i + 1];
      const double convexity = ax * by - ay * bx;
      if (convexity != 0) {
        orientation = (convexity > 0) ? 1.f : -1.f;
        break;
      }
      ax = bx;
      ay = by;
    }
  }
  base_a = orientation;
  seq[0] = bottom;
  seq[1] = right;
  seq[2] = top;
  seq[3] = left;
  for (k = 0; k < n; ++k) {
    const float dp[4] = {
        +base_a * vect[2 * seq[0]] + base_b * vect[2 * seq[0] + 1],
        -base_b * vect[2 * seq[1]] + base_a * vect[2 * seq[1] + 1],
        -base_a * vect[2 * seq[2]] - base_b * vect[2 * seq[2] + 1],
        +base_b * vect[2 * seq[3]] - base_a * vect[2 * seq[3] + 1],
    };
    float maxcos = dp[0] * inv_vect_length[seq[0]];
    int main_element = 0;
    for (i = 1; i < 4; ++i) {
      const float cosalpha = dp[i] * inv_vect_length[seq[i]];
      if (cosalpha > maxcos) {
        main_element = i;
        maxcos = cosalpha;
      }
    }
    {
      const int pindex = seq[main_element];
      const float lead_x = vect[2 * pindex] * inv_vect_length[pindex];
      const float lead_y = vect[2 * pindex + 1] * inv_vect_length[pindex];
      switch (main_element) {
        case 0: base_a = lead_x; base_b = lead_y; break;
        case 1: base_a = lead_y; base_b = -lead_x; break;
        case 2: base_a = -lead_x; base_b = -lead_y; break;
        default: base_a = -lead_y; base_b = lead_x; break;
      }
    }
    seq[main_element] += 1;
    seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
    {
      float dx = pts[2 * seq[1]] - pts[2 * seq[3]];
      float dy = pts[2 * seq[1] + 1] - pts[2 * seq[3] + 1];
      const float width = dx * base_a + dy * base_b;
      dx = pts[2 * seq[2]] - pts[2 * seq[0]];
      dy = pts[2 * seq[2] + 1] - pts[2 * seq[0] + 1];
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
  }
  {
    const float A1 = b_a, B1 = b_b, A2 = -b_b, B2 = b_a;
    const float C1 = A1 * pts[2 * b_left] + pts[2 * b_left + 1] * B1;
    const float C2 = A2 * pts[2 * b_bottom] + pts[2 * b_bottom + 1] * B2;
    const float idet = 1.f / (A1 * B2 - A2 * B1);
    out[0] = (C1 * B2 - C2 * B1) * idet;
    out[1] = (A1 * C2 - A2 * C1) * idet;
    out[2] = A1 * b_width;
    out[3] = B1 * b_width;
    out[4] = A2 * b_height;
    out[5] = B2 * b_height;
  }
  free(inv_vect_length);
}

/* RotatedRect from the calipers' corner + edge vectors, or from a degenerate hull (n <= 2):
 * rect = (cx, cy, width, height, angle in degrees) — minAreaRect's tail. */
void cvgeom_rect_from_hull(const int* hull_xy, int nh, const float* cal6, float* rect) {
  float cx = 0.f, cy = 0.f, w = 0.f, h = 0.f, angle = 0.f;
  if (nh > 2) {
    cx = cal6[0] + (cal6[2] + cal6[4]) * 0.5f;
    cy = cal6[1] + (cal6[3] + cal6[5]) * 0.5f;
    w = (float)sqrt((double)cal6[2] * cal6[2] + (double)cal6[3] * cal6[3]);
    h = (float)sqrt((double)cal6[4] * cal6[4] + (double)cal6[5] * cal6[5]);
    angle = (float)atan2((double)cal6[3], (double)cal6[2]);
  } else if (nh == 2) {
    const float x0 = (float)hull_xy[0], y0 = (float)hull_xy[1], x1 = (float)hull_xy[2], y1 = (float)hull_xy[3];
    cx = (x0 + x1) * 0.5f;
    cy = (y0 + y1) * 0.5f;
    const double dx = x1 - x0, dy = y1 - y0;
    w = (float)sqrt(dx * dx + dy * dy);
    h = 0.f;
    angle = (float)atan2(dy, dx);
  } else if (nh == 1) {
    cx = (float)hull_xy[0];
    cy = (float)hull_xy[1];
  }
  rect[0] = cx;
  rect[1] = cy;
  rect[2] = w;
  rect[3] = h;
  rect[4] = (float)(angle * 180 / 3.1415926535897932384626433832795);
}

/* cv2.boxPoints: RotatedRect::points -> pts[4][2] */
void cvgeom_box_points(const float* rect, float* pts) {
  const double ang = rect[4] * 3.1415926535897932384626433832795 / 180.;
  const float b = (float)cos(ang) * 0.5f;
  const float a = (float)sin(ang) * 0.5f;
  const float cx = rect[0], cy = rect[1], w = rect[2], h = rect[3];
  pts[0] = cx - a * h - b * w;
  pts[1] = cy + b * h - a * w;
  pts[2] = cx + a * h - b * w;
  pts[3] = cy - b * h - a * w;
  pts[4] = 2 * cx - pts[0];
  pts[5] = 2 * cy - pts[1];
  pts[6] = 2 * cx - pts[2];
  pts[7] = 2 * cy - pts[3];
}

/* cv2.minAreaRect(points int32 [n][2]): hull_out (may be NULL) receives the hull, cal6 the calipers
 * output (zeros when the hull has <= 2 points), rect = (cx,cy,w,h,angle).  Returns hull size. */
int cvgeom_min_area_rect(const int* pts_xy, int n, float* rect, float* cal6, int* hull_out) {
  int* hull = (int*)malloc(sizeof(int) * 2 * (size_t)(n > 0 ? n : 1));
  const int nh = cvgeom_convex_hull(pts_xy, n, hull);
  float c6[6] = {0, 0, 0, 0, 0, 0};
  if (nh > 2) {
    float* hp = (float*)malloc(sizeof(float) * 2 * (size_t)nh);
    for (int i = 0; i < 2 * nh; ++i) hp[i] = (float)hull[i];
    cvgeom_rotating_calipers(hp, nh, c6);
    free(hp);
  }
  cvgeom_rect_from_hull(hull, nh, c6, rect);
  if (cal6) memcpy(cal6, c6, sizeof(c6));
  if (hull_out) memcpy(hull_out, hull, sizeof(int) * 2 * (size_t)nh);
  free(hull);
  return nh;
}

/* ------------------------------------------------------------------------------------------ *
 * cv2.fillPoly(img uint8 [h][w], [pts int32 [count][2]], color), lineType 8, shift 0:
 * CollectPolyEdges (draws every edge with Line -> clipLine + LineIterator(left_to_right)) and
 * FillEdgeCollection (sorted edge table, active list, even-odd spans with 16.16 fixed point).
 * ------------------------------------------------------------------------------------------ */
#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

static int clip_line(int64_t width, int64_t height, int64_t* px1, int64_t* py1, int64_t* px2, int64_t* py2) {
  int c1, c2;
  const int64_t right = width - 1, bottom = height - 1;
  int64_t x1 = *px1, y1 = *py1, x2 = *px2, y2 = *py2;
  if (width <= 0 || height <= 0) return 0;
  c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int64_t a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (int64_t)((double)(a - y1) * (x2 - x1) / (y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (int64_t)((double)(a - y2) * (x2 - x1) / (y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (int64_t)((double)(a - x1) * (y2 - y1) / (x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (int64_t)((double)(a - x2) * (y2 - y1) / (x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  *px1 = x1; *py1 = y1; *px2 = x2; *py2 = y2;
  return (c1 | c2) == 0;
}

/* Line(): 8-connected LineIterator from the left end point to the right one */
static void draw_line(unsigned char* img, int h, int w, int64_t X1, int64_t Y1, int64_t X2, int64_t Y2,
                      unsigned char color) {
  if (!clip_line(w, h, &X1, &Y1, &X2, &Y2)) return;
  int x1 = (int)X1, y1 = (int)Y1, x2 = (int)X2, y2 = (int)Y2;
  int dx = x2 - x1, dy = y2 - y1;
  int xstep = 1, ystep = 1;
  if (dx < 0) {            /* left_to_right: start from the left end */
    dx = -dx;
    dy = -dy;
    x1 = x2;
    y1 = y2;
  }
  if (dy < 0) { dy = -dy; ystep = -1; }
  int x = x1, y = y1;
  const int steep = dy > dx;
  int major = steep ? dy : dx, minor = steep ? dx : dy;
  int err = major - (minor + minor);
  const int plusDelta = major + major, minusDelta = -(minor + minor);
  const int count = major + 1;
  for (int i = 0; i < count; ++i) {
    img[(size_t)y * w + x] = color;
    const int mask = err < 0 ? -1 : 0;
    err += minusDelta + (plusDelta & mask);
    if (steep) { y += ystep; if (mask) x += xstep; }
    else { x += xstep; if (mask) y += ystep; }
  }
}

typedef struct PolyEdge {
  int y0, y1;
  int64_t x, dx;
  struct PolyEdge* next;
} PolyEdge;

static int cmp_edges(const void* a, const void* b) {
  const PolyEdge* e1 = (const PolyEdge*)a;
  const PolyEdge* e2 = (const PolyEdge*)b;
  if (e1->y0 != e2->y0) return e1->y0 < e2->y0 ? -1 : 1;
  if (e1->x != e2->x) return e1->x < e2->x ? -1 : 1;
  if (e1->dx != e2->dx) return e1->dx < e2->dx ? -1 : 1;
  return 0;
}

static void hline(unsigned char* row, int x1, int x2, unsigned char color) {
  for (int x = x1; x <= x2; ++x) row[x] = color;
}

void cvgeom_fill_poly(unsigned char* img, int h, int w, const int* pts_xy, int count, int color) {
  if (count <= 0) return;
  const unsigned char col = (unsigned char)(color < 0 ? 0 : (color > 255 ? 255 : color));
  PolyEdge* edges = (PolyEdge*)malloc(sizeof(PolyEdge) * (size_t)(count + 1));
  int total = 0, i;
  /* CollectPolyEdges, shift 0, offset 0 */
  int64_t p0x = (int64_t)pts_xy[2 * (count - 1)] << XY_SHIFT, p0y = pts_xy[2 * (count - 1) + 1];
  for (i = 0; i < count; ++i) {
    const int64_t p1x = (int64_t)pts_xy[2 * i] << XY_SHIFT, p1y = pts_xy[2 * i + 1];
    const int64_t t0x = (p0x + (XY_ONE >> 1)) >> XY_SHIFT, t1x = (p1x + (XY_ONE >> 1)) >> XY_SHIFT;
    draw_line(img, h, w, t0x, p0y, t1x, p1y, col);
    if (p0y != p1y) {
      PolyEdge e;
      if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = p0x; }
      else { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = p1x; }
      e.dx = (p1x - p0x) / (p1y - p0y);
      e.next = 0;
      edges[total++] = e;
    }
    p0x = p1x;
    p0y = p1y;
  }
  /* FillEdgeCollection */
  if (total >= 2) {
    int y_max = INT_MIN, y_min = INT_MAX;
    int64_t x_max = -1, x_min = 0x7FFFFFFFFFFFFFFFLL;
    for (i = 0; i < total; ++i) {
      const PolyEdge* e1 = &edges[i];
      const int64_t x1 = e1->x + (e1->y1 - e1->y0) * e1->dx;
      if (e1->y0 < y_min) y_min = e1->y0;
      if (e1->y1 > y_max) y_max = e1->y1;
      if (e1->x < x_min) x_min = e1->x;
      if (e1->x > x_max) x_max = e1->x;
      if (x1 < x_min) x_min = x1;
      if (x1 > x_max) x_max = x1;
    }
    if (!(y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((int64_t)w << XY_SHIFT))) {
      qsort(edges, (size_t)total, sizeof(PolyEdge), cmp_edges);
      PolyEdge tmp;
      tmp.y0 = INT_MAX;
      tmp.next = 0;
      edges[total] = tmp;
      edges[total].y0 = INT_MAX;
      i = 0;
      PolyEdge* e = &edges[i];
      if (y_max > h) y_max = h;
      for (int y = e->y0; y < y_max; ++y) {
        PolyEdge *last, *prelast, *keep_prelast;
        int sort_flag = 0, draw = 0;
        const int clipline = y < 0;
        prelast = &tmp;
        last = tmp.next;
        while (last || e->y0 == y) {
          if (last && last->y1 == y) {
            prelast->next = last->next;
            last = last->next;
            continue;
          }
          keep_prelast = prelast;
          if (last && (e->y0 > y || last->x < e->x)) {
            prelast = last;
            last = last->next;
          } else if (i < total) {
            prelast->next = e;
            e->next = last;
            prelast = e;
            e = &edges[++i];
          } else
            break;
          if (draw) {
            if (!clipline) {
              int x1, x2;
              if (keep_prelast->x > prelast->x) {
                x1 = (int)((prelast->x + XY_ONE - 1) >> XY_SHIFT);
                x2 = (int)(keep_prelast->x >> XY_SHIFT);
              } else {
                x1 = (int)((keep_prelast->x + XY_ONE - 1) >> XY_SHIFT);
                x2 = (int)(prelast->x >> XY_SHIFT);
              }
              if (x1 < w && x2 >= 0) {
                if (x1 < 0) x1 = 0;
                if (x2 >= w) x2 = w - 1;
                hline(img + (size_t)y * w, x1, x2, col);
              }
            }
            keep_prelast->x += keep_prelast->dx;
            prelast->x += prelast->dx;
          }
          draw ^= 1;
        }
        /* bubble-sort the active list by x */
        keep_prelast = 0;
        do {
          prelast = &tmp;
          last = tmp.next;
          while (last != keep_prelast && last->next != 0) {
            PolyEdge* te = last->next;
            if (last->x > te->x) {
              prelast->next = te;
              last->next = te->next;
              te->next = last;
              prelast = te;
              sort_flag = 1;
            } else {
              prelast = last;
              last = te;
            }
          }
          keep_prelast = prelast;
        } while (sort_flag && keep_prelast != tmp.next && keep_prelast != &tmp);
      }
    }
  }
  free(edges);
}

/* ------------------------------------------------------------------------------------------ *
 * cv2.resize(src uint8 [H][W][cn], dsize=(dw, dh)), default INTER_LINEAR (datasets/icdar.py:615):
 * resizeGeneric_ with HResizeLinear<uchar,int,short,2048> / VResizeLinear<uchar,int,short> —
 * 11-bit fixed-point coefficients (cvRound of float weights), half-pixel centres, horizontal pass
 * into int, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.  An exact 2x2 downscale
 * is rerouted to INTER_AREA by resize() itself ((a+b+c+d+2)>>2).
 * ------------------------------------------------------------------------------------------ */
static int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static short sat_short_round(float v) {
  long r = lrintf(v);           /* cvRound: round half to even */
  if (r < -32768) r = -32768;
  if (r > 32767) r = 32767;
  return (short)r;
}

void cvgeom_resize_linear_u8(const unsigned char* src, int H, int W, int cn, unsigned char* dst, int dh, int dw) {
  const double inv_scale_x = (double)dw / W, inv_scale_y = (double)dh / H;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
  const int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x)
        for (int k = 0; k < cn; ++k) {
          const unsigned char* s0 = src + ((size_t)(2 * y) * W + 2 * x) * cn + k;
          const unsigned char* s1 = s0 + (size_t)W * cn;
          dst[((size_t)y * dw + x) * cn + k] = (unsigned char)((s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2);
        }
    return;
  }
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= W - 1) { fx = 0; sx = W - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short_round((1.f - fx) * 2048);
    ialpha[2 * dx + 1] = sat_short_round(fx * 2048);
  }
  int* hbuf0 = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  int* hbuf1 = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = cv_floor(fy);
    fy -= sy;
    const short b0 = sat_short_round((1.f - fy) * 2048), b1 = sat_short_round(fy * 2048);
    int r0 = sy, r1 = sy + 1;
    if (r0 < 0) r0 = 0;
    if (r0 > H - 1) r0 = H - 1;
    if (r1 < 0) r1 = 0;
    if (r1 > H - 1) r1 = H - 1;
    for (int pass = 0; pass < 2; ++pass) {
      const unsigned char* S = src + (size_t)(pass ? r1 : r0) * W * cn;
      int* D = pass ? hbuf1 : hbuf0;
      for (int dx = 0; dx < dw; ++dx) {
        const int sx = xofs[dx];
        const int sx1 = sx + 1 < W ? sx + 1 : sx;      /* dx >= xmax: D = S[sx] * ONE (alpha1 is 0) */
        for (int k = 0; k < cn; ++k)
          D[dx * cn + k] = S[sx * cn + k] * ialpha[2 * dx] + S[sx1 * cn + k] * ialpha[2 * dx + 1];
      }
    }
    for (int x = 0; x < dw * cn; ++x) {
      const int v = (((b0 * (hbuf0[x] >> 4)) >> 16) + ((b1 * (hbuf1[x] >> 4)) >> 16) + 2) >> 2;
      dst[(size_t)dy * dw * cn + x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  free(hbuf0); free(hbuf1); free(ialpha); free(xofs);
}


/* cv2.resize(float32 plane, INTER_CUBIC): OpenCV imgproc resize.cpp — interpolateCubic (A = -0.75),
 * the xofs/alpha tables of cv::resize (fx at the half-pixel centre, scale = 1/(dsize/ssize)),
 * HResizeCubic (border taps walked back into the row), VResizeCubic over rows clipped to the image.
 * src [h][w] float -> dst [dh][dw] float.  test_pixellink.py:97-98,108-109. */
static void interpolate_cubic(float x, float* coeffs) {
  const float A = -0.75f;
  coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

void cvgeom_resize_cubic_f32(const float* src, int h, int w, float* dst, int dh, int dw) {
  const double scale_x = 1. / ((double)dw / w), scale_y = 1. / ((double)dh / h);
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
  float* alpha = (float*)malloc(sizeof(float) * 4 * (size_t)dw);
  int xmin = 0, xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 1) xmin = dx + 1;
    if (sx + 2 >= w && dx < xmax) xmax = dx;
    xofs[dx] = sx;
    interpolate_cubic(fx, alpha + 4 * dx);
  }
  float* rows = (float*)malloc(sizeof(float) * 4 * (size_t)dw);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = cv_floor(fy);
    fy -= sy;
    float beta[4];
    interpolate_cubic(fy, beta);
    for (int k = 0; k < 4; ++k) {
      int r = sy - 1 + k;
      if (r < 0) r = 0;
      if (r > h - 1) r = h - 1;
      const float* S = src + (size_t)r * w;
      float* D = rows + (size_t)k * dw;
      for (int dx = 0; dx < dw; ++dx) {
        const float* a = alpha + 4 * dx;
        const int sx = xofs[dx] - 1;
        if (dx >= xmin && dx < xmax) {
          D[dx] = S[sx] * a[0] + S[sx + 1] * a[1] + S[sx + 2] * a[2] + S[sx + 3] * a[3];
        } else {
          float v = 0;
          for (int j = 0; j < 4; ++j) {
            int sxj = sx + j;
            while (sxj < 0) sxj += 1;
            while (sxj >= w) sxj -= 1;
            v += S[sxj] * a[j];
          }
          D[dx] = v;
        }
      }
    }
    for (int x = 0; x < dw; ++x)
      dst[(size_t)dy * dw + x] = rows[x] * beta[0] + rows[dw + x] * beta[1] + rows[2 * dw + x] * beta[2] +
                                 rows[3 * dw + x] * beta[3];
  }
  free(rows); free(alpha); free(xofs);
}
