/* ORACLE (test infrastructure only): plain-C restatement of locality-aware NMS.
 *
 * The reference tree contains no NMS of any kind (tool/bboxes.py holds only crop /
 * eval helpers), so this follows the published algorithm the north star names: EAST (Zhou et al.,
 * CVPR 2017) Algorithm 1 — row-major weighted merge of consecutive quadrangles whose IoU exceeds the
 * threshold, then standard score-ordered NMS.  PARITY UNPINNED against the reference; the HIP
 * kernel (tensorflow_ocr_amd/csrc/lanms.hip) is pinned against THIS file, bit for bit
 * (float32, no FMA contraction: build with -ffp-contract=off).
 *
 * Quads are 9 floats: x1,y1,x2,y2,x3,y3,x4,y4,score.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } pt;

static float signed_area(const pt* p, int n) {
  float a = 0.f;
  for (int i = 0; i < n; ++i) {
    const pt u = p[i], v = p[(i + 1) % n];
    a += u.x * v.y - v.x * u.y;
  }
  return a * 0.5f;
}

static float cross3(pt a, pt b, pt c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }

static pt intersect(pt s, pt e, pt c1, pt c2) {
  /* intersection of segment s-e with the infinite line c1-c2 */
  const float d1 = cross3(c1, c2, s), d2 = cross3(c1, c2, e);
  const float t = d1 / (d1 - d2);
  pt r;
  r.x = s.x + t * (e.x - s.x);
  r.y = s.y + t * (e.y - s.y);
  return r;
}

static void load_ccw(const float* q, pt* out) {
  for (int i = 0; i < 4; ++i) { out[i].x = q[2 * i]; out[i].y = q[2 * i + 1]; }
  if (signed_area(out, 4) < 0.f) {
    pt t = out[1]; out[1] = out[3]; out[3] = t;
  }
}

float lanms_oracle_iou(const float* qa, const float* qb) {
  pt a[4], b[4], cur[16], nxt[16];
  load_ccw(qa, a);
  load_ccw(qb, b);
  const float area_a = fabsf(signed_area(a, 4)), area_b = fabsf(signed_area(b, 4));
  int n = 4;
  memcpy(cur, a, sizeof(a));
  for (int ce = 0; ce < 4 && n > 0; ++ce) {          /* Sutherland-Hodgman, clip polygon b */
    const pt c1 = b[ce], c2 = b[(ce + 1) % 4];
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const pt s = cur[i], e = cur[(i + 1) % n];
      const int sin = cross3(c1, c2, s) >= 0.f, ein = cross3(c1, c2, e) >= 0.f;
      if (sin && ein) nxt[m++] = e;
      else if (sin && !ein) nxt[m++] = intersect(s, e, c1, c2);
      else if (!sin && ein) { nxt[m++] = intersect(s, e, c1, c2); nxt[m++] = e; }
    }
    n = m;
    memcpy(cur, nxt, sizeof(pt) * (size_t)m);
  }
  const float inter = n >= 3 ? fabsf(signed_area(cur, n)) : 0.f;
  const float uni = area_a + area_b - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

static void weighted_merge(const float* g, const float* p, float* out) {
  const float sg = g[8], sp = p[8], s = sg + sp;
  for (int i = 0; i < 8; ++i) out[i] = (sg * g[i] + sp * p[i]) / s;
  out[8] = s;
}

/* boxes [k][9] in row-major scan order.  merged [k][9] (out), keep_idx [k] (out, indices into
 * merged, in keep order).  Returns n_keep; *n_merged receives the merged count. */
int lanms_oracle(const float* boxes, int k, float thr, float* merged, int* n_merged, int* keep_idx) {
  int m = 0, have = 0;
  float p[9], q[9];
  for (int i = 0; i < k; ++i) {
    const float* g = boxes + 9 * i;
    if (have && lanms_oracle_iou(g, p) > thr) {
      weighted_merge(g, p, q);
      memcpy(p, q, sizeof(p));
    } else {
      if (have) { memcpy(merged + 9 * m, p, sizeof(p)); ++m; }
      memcpy(p, g, sizeof(p));
      have = 1;
    }
  }
  if (have) { memcpy(merged + 9 * m, p, sizeof(p)); ++m; }
  *n_merged = m;
  /* standard NMS: stable order by descending score */
  int* order = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  for (int i = 0; i < m; ++i) {
    int r = 0;
    for (int j = 0; j < m; ++j)
      if (merged[9 * j + 8] > merged[9 * i + 8] || (merged[9 * j + 8] == merged[9 * i + 8] && j < i)) ++r;
    order[r] = i;
  }
  char* dead = (char*)calloc((size_t)(m > 0 ? m : 1), 1);
  int nk = 0;
  for (int a = 0; a < m; ++a) {
    if (dead[a]) continue;
    keep_idx[nk++] = order[a];
    for (int b = a + 1; b < m; ++b)
      if (!dead[b] && lanms_oracle_iou(merged + 9 * order[a], merged + 9 * order[b]) > thr) dead[b] = 1;
  }
  free(order);
  free(dead);
  return nk;
}
