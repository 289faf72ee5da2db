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
