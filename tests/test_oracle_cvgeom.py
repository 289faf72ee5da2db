"""CPU: the OpenCV-geometry restatement (oracle/cvgeom_oracle.c) against brute force and known
answers.  cv2 itself is absent here (PARITY UNPINNED, see the C file's header)."""
import numpy as np
import pytest

from oracle import cvgeom as C


def _hull_ref(P):
    """Andrew's monotone chain, strict vertices, counter-clockwise (y up) from the lexicographic min."""
    P = sorted(set(map(tuple, np.asarray(P).tolist())))
    if len(P) <= 2:
        return P

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lo, up = [], []
    for p in P:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], p) <= 0:
            lo.pop()
        lo.append(p)
    for p in reversed(P):
        while len(up) >= 2 and cross(up[-2], up[-1], p) <= 0:
            up.pop()
        up.append(p)
    return lo[:-1] + up[:-1]


def test_convex_hull_order_and_min_area_vs_brute_force():
    rng = np.random.default_rng(0)
    for t in range(1500):
        n = int(rng.integers(1, 60))
        P = rng.integers(0, int(rng.integers(2, 40)), size=(n, 2))
        if t % 5 == 0:
            P[:, 1] = P[:, 0] * 2 + 1          # collinear
        if t % 7 == 0:
            P[:, 1] = 3                        # one row
        r = _hull_ref(P)
        want = [r[0]] + r[1:][::-1] if len(r) > 2 else r     # clockwise, same start
        got = [tuple(int(v) for v in x) for x in C.convex_hull(P)]
        assert got == want
        rect, cal, hull = C.min_area_rect(P)
        if len(r) > 2:
            H = np.array(r, float)
            best = np.inf
            for i in range(len(H)):
                e = H[(i + 1) % len(H)] - H[i]
                e /= np.linalg.norm(e)
                nrm = np.array([-e[1], e[0]])
                best = min(best, np.ptp(H @ e) * np.ptp(H @ nrm))
            assert abs(float(rect[2]) * float(rect[3]) - best) <= 1e-3 * max(best, 1.0)
            # every point inside the returned box (half-plane test on the 4 corners)
            bp = C.box_points(rect).astype(float)
            ctr = bp.mean(0)
            for a, b in zip(bp, np.roll(bp, -1, axis=0)):
                nrm = np.array([-(b - a)[1], (b - a)[0]])
                if nrm @ (ctr - a) < 0:
                    nrm = -nrm
                assert ((H - a) @ nrm >= -1e-2 * max(np.linalg.norm(nrm), 1)).all()


def test_min_area_rect_known_answers():
    # axis-aligned 5x3 block of pixels: rectangle 4 x 2 around (2,1)
    ys, xs = np.mgrid[0:3, 0:5]
    pts = np.stack([xs.ravel(), ys.ravel()], 1)
    rect, cal, hull = C.min_area_rect(pts)
    assert len(hull) == 4
    assert np.allclose(rect[:2], [2, 1]) and sorted(np.round(rect[2:4]).tolist()) == [2, 4]
    assert abs(float(rect[4])) in (90.0, 0.0)
    box = C.box_points(rect)
    assert sorted(map(tuple, np.round(box).astype(int).tolist())) == [(0, 0), (0, 2), (4, 0), (4, 2)]
    # 45-degree diamond
    pts = np.array([[2, 0], [4, 2], [2, 4], [0, 2], [2, 2]])
    rect, _, hull = C.min_area_rect(pts)
    assert len(hull) == 4 and np.allclose(rect[:2], [2, 2], atol=1e-5)
    assert np.allclose(rect[2:4], [np.sqrt(8)] * 2, atol=1e-5) and abs(abs(float(rect[4])) - 45) < 1e-4
    # degenerate hulls: one point, two points (minAreaRect's n == 1 / n == 2 branches)
    rect, _, hull = C.min_area_rect(np.array([[7, 9]]))
    assert len(hull) == 1 and rect.tolist() == [7, 9, 0, 0, 0]
    rect, _, hull = C.min_area_rect(np.array([[0, 0], [3, 4], [6, 8]]))
    assert len(hull) == 2 and np.allclose(rect[:4], [3, 4, 10, 0]) and abs(float(rect[4]) - 53.130102) < 1e-4


def test_host_rect_formatting_equals_oracle_tail():
    """tool.pixellink_fn._rotated_rect / _box_points (host, O(1) per box) == the oracle's C tail."""
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    rng = np.random.default_rng(1)
    for t in range(300):
        n = int(rng.integers(1, 40))
        pts = rng.integers(0, 300, size=(n, 2))
        if t % 6 == 0:
            pts[:, 0] = 5
        rect, cal, hull = C.min_area_rect(pts)
        head = np.zeros(4, np.int32)
        head[:min(len(hull), 2) * 2] = hull[:2].ravel()
        mine = P._rotated_rect(len(hull), head, cal)
        assert mine.tobytes() == rect.tobytes()
        assert P._box_points(mine).tobytes() == C.box_points(rect).tobytes()


def _closed_form_cover(h, w, pts):
    """Independent statement of fillPoly's raster: per pixel, Bresenham membership of each edge
    (closed form, no clipping: vertices in range) or even-odd interior from 16.16 edge crossings."""
    pts = [tuple(int(v) for v in p) for p in pts]
    out = np.zeros((h, w), np.uint8)
    segs, edges = [], []
    for i in range(len(pts)):
        (x0, y0), (x1, y1) = pts[i - 1], pts[i]
        segs.append((x0, y0, x1, y1))
        if y0 != y1:
            top = (x0, y0) if y0 < y1 else (x1, y1)
            num, den = (x1 - x0) << 16, y1 - y0
            dx = abs(num) // abs(den) * (1 if (num >= 0) == (den > 0) else -1)       # C truncation
            edges.append((min(y0, y1), max(y0, y1), top[0] << 16, dx))
    for y in range(h):
        xs = sorted(e[2] + (y - e[0]) * e[3] for e in edges if e[0] <= y < e[1]) if len(edges) >= 2 else []
        for x in range(w):
            inside = any(((xs[k] + 65535) >> 16) <= x <= (xs[k + 1] >> 16) for k in range(0, len(xs) - 1, 2))
            for (x0, y0, x1, y1) in segs:
                dx, dy, sx, sy = x1 - x0, y1 - y0, x0, y0
                if dx < 0:
                    dx, dy, sx, sy = -dx, -dy, x1, y1
                ys = 1 if dy >= 0 else -1
                dy = abs(dy)
                if dy > dx:
                    j = (y - sy) * ys
                    inside |= 0 <= j <= dy and x == sx + (2 * dx * j + dy - 1) // (2 * dy)
                else:
                    j = x - sx
                    inside |= 0 <= j <= dx and y == sy + ys * ((2 * dy * j + dx - 1) // (2 * dx) if dx else 0)
            out[y, x] = inside
    return out


def test_fill_poly_known_answers_and_closed_form():
    img = C.fill_poly(np.zeros((12, 14), np.uint8), [[2, 3], [9, 3], [9, 7], [2, 7]], 5)
    want = np.zeros((12, 14), np.uint8)
    want[3:8, 2:10] = 5                                     # inclusive of the far edges
    assert np.array_equal(img, want)
    img = C.fill_poly(np.zeros((12, 14), np.uint8), [[0, 0], [10, 0], [0, 10]], 1)
    ys, xs = np.mgrid[0:12, 0:14]
    assert np.array_equal(img, (xs + ys <= 10).astype(np.uint8))
    assert C.fill_poly(np.zeros((8, 8), np.uint8), [[-3, -2], [12, 1], [9, 10], [-1, 6]], 1)[:7].all()
    rng = np.random.default_rng(5)
    for t in range(150):
        h, w = int(rng.integers(4, 30)), int(rng.integers(4, 30))
        k = int(rng.integers(3, 7))
        pts = np.stack([rng.integers(0, w, k), rng.integers(0, h, k)], 1)
        if t % 9 == 0:
            pts[:, 1] = pts[0, 1]
        assert np.array_equal(C.fill_poly(np.zeros((h, w), np.uint8), pts, 1), _closed_form_cover(h, w, pts))


def test_label_generators_small_known_answers():
    from oracle import labels as OL
    # one axis-aligned box in a 16x16 image: icdar.generate_rbox links are all 1 inside except
    # where the (transposed) neighbour falls outside the box
    poly = np.array([[[4, 4], [11, 4], [11, 11], [4, 11]]], np.float32)
    s, g, m = OL.icdar_generate_rbox((16, 16), poly, np.array([False]), min_text_size=3)
    assert s.sum() == 64 and m.all()
    assert g[8, 8].tolist() == [1] * 8
    assert g[4, 4].tolist() == [0, 0, 0, 1, 1, 0, 0, 1]     # channels look at (0,-1),(1,-1),(-1,-1),(0,1),(1,1),(-1,1),(-1,0),(1,0)
    assert g[0, 0].sum() == 0
    # a tagged polygon zeroes the training mask; a small one too
    s, g, m = OL.icdar_generate_rbox((16, 16), poly, np.array([True]))
    assert (m[4:12, 4:12] == 0).all() and m.sum() == 256 - 64
    s4, g4, m4 = OL.icdar_labels((16, 16), poly, np.array([False]), min_text_size=3)
    assert s4.shape == (4, 4, 1) and g4.shape == (4, 4, 8) and s4[..., 0].tolist() == [[0] * 4, [0, 1, 1, 0], [0, 1, 1, 0], [0] * 4]
    # pixellink_fn.generate_rbox: label equality with true directions, borders forced to 1
    xs = np.array([[0.0, 0.99, 0.99, 0.0]], np.float32)
    ys = np.array([[0.0, 0.0, 0.5, 0.5]], np.float32)
    sc, lk, sb = OL.pixellink_generate_rbox(32, 32, xs, ys, np.array([[1, 2, 3, 4]], np.float32), np.array([0]))
    assert sc.shape == (8, 8) and sc[:5].all() and not sc[5:].any()
    assert lk[0, 0].tolist() == [1] * 8 and lk[2, 2].tolist() == [1] * 8
    assert lk[4, 3].tolist() == [1, 0, 1, 1, 0, 1, 1, 0]    # bottom row of the box: the three 'down' links are 0
    assert sb[0].tolist() == [1, 2, 3, 4] and not sb[1:].any()


def test_resize_linear_matches_float_bilinear_and_special_cases():
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, size=(37, 53, 3)).astype(np.uint8)
    assert np.array_equal(C.resize_linear_u8(src, 37, 53), src)                 # identity
    half = C.resize_linear_u8(src[:36, :52], 18, 26)                            # exact /2 -> INTER_AREA
    a = src[:36, :52].astype(np.int32)
    assert np.array_equal(half, ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2))
    for dh, dw in [(64, 64), (20, 30), (74, 106)]:
        d = C.resize_linear_u8(src, dh, dw).astype(float)
        H, W, _ = src.shape
        fy = (np.arange(dh) + 0.5) * H / dh - 0.5
        fx = (np.arange(dw) + 0.5) * W / dw - 0.5
        y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
        wy, wx = fy - y0, fx - x0
        wx = np.where((x0 < 0) | (x0 >= W - 1), 0, wx)
        x0 = np.clip(x0, 0, W - 1)
        x1 = np.clip(x0 + 1, 0, W - 1)
        y1, y0 = np.clip(y0 + 1, 0, H - 1), np.clip(y0, 0, H - 1)
        s = src.astype(float)
        top = s[y0][:, x0] * (1 - wx)[None, :, None] + s[y0][:, x1] * wx[None, :, None]
        bot = s[y1][:, x0] * (1 - wx)[None, :, None] + s[y1][:, x1] * wx[None, :, None]
        ref = top * (1 - wy)[:, None, None] + bot * wy[:, None, None]
        assert np.abs(d - ref).max() < 1.0


def test_icdar_host_parsing_and_validation(tmp_path):
    from tensorflow_ocr_amd.datasets import icdar
    p = tmp_path / "gt_img_1.txt"
    p.write_text("\ufeff10,10,50,10,50,30,10,30,hello\n10,10,10,30,50,30,50,10,###\n5,5,5,5,5,5,5,5,dot\n"
                 "-20,4,700,4,700,40,-20,40,a,b\n", encoding="utf-8")
    polys, tags = icdar.load_annoataion(str(p))
    assert polys.shape == (4, 4, 2) and polys.dtype == np.float32 and tags.tolist() == [False, True, False, False]
    assert icdar.polygon_area(polys[0]) == -800.0 and icdar.polygon_area(polys[1]) == 800.0
    v, t = icdar.check_and_validate_polys(polys.copy(), tags, (100, 200))
    assert len(v) == 3 and t.tolist() == [False, True, False]
    assert np.array_equal(v[1], polys[1][[0, 3, 2, 1]])           # wrong direction: re-ordered
    assert v[2][:, 0].min() == 0 and v[2][:, 0].max() == 199       # clipped to the image
    assert icdar.txt_name("/data/x/img_1.jpg") == "/data/x/gt_img_1.txt"
    pk, cnt, ign = icdar.pack_polys([v, v[:0]], [t, t[:0]])
    assert pk.shape == (2, 3, 4, 2) and cnt.tolist() == [3, 0] and ign[0].tolist() == [0, 1, 0]
    with pytest.raises(ValueError):
        icdar.pack_polys([np.zeros((255, 4, 2), np.float32)], [np.zeros(255, bool)])


def test_golden_cv_geometry_fixture():
    """The restatements still produce the committed vectors (tests/golden/cv_geometry.npz, written by
    tests/golden/make_golden.py from these same functions: a pin against accidental change)."""
    import os
    from oracle import contours as OC
    from oracle import evalboxes as OE
    from oracle import labels as OL
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cv_geometry.npz"))
    rect, cal, hull = C.min_area_rect(g["mar_pts"])
    assert rect.tobytes() == g["mar_rect"].tobytes() and cal.tobytes() == g["mar_cal"].tobytes()
    assert np.array_equal(hull, g["mar_hull"]) and C.box_points(rect).tobytes() == g["mar_box"].tobytes()
    assert np.array_equal(C.fill_poly(np.zeros((48, 64), np.uint8), g["fill_quad"], 1), g["fill_img"])
    assert np.array_equal(C.resize_linear_u8(g["rs_src"], 64, 64), g["rs_64"])
    assert np.array_equal(C.resize_linear_u8(g["rs_src"][:36, :52], 18, 26), g["rs_half"])
    s4, g4, m4 = OL.icdar_labels((64, 64), g["lab_polys"], g["lab_tags"])
    assert np.array_equal(s4, g["lab_score"]) and np.array_equal(g4, g["lab_geo"]) and np.array_equal(m4, g["lab_mask"])
    polys = g["lab_polys"]
    ps, pl, _ = OL.pixellink_generate_rbox(64, 64, polys[:, :, 0] / 64, polys[:, :, 1] / 64,
                                           np.zeros((3, 4), np.float32), np.zeros(3, np.int32))
    assert np.array_equal(ps, g["pl_score"]) and np.array_equal(pl, g["pl_link"])
    gx, gy = polys[:, :, 0].astype(int), polys[:, :, 1].astype(int)
    n, tp, fp = OE.bboxes_matching(g["ev_det"], gx, gy, np.array([0, 0, 1]))
    assert n == int(g["ev_n"]) and np.array_equal(tp, g["ev_tp"]) and np.array_equal(fp, g["ev_fp"])
    assert np.array_equal(np.stack([OE.np_bboxes_jaccard(d, gx, gy) for d in g["ev_det"]]), g["ev_iou"])
    rects, boxes = OC.contour_boxes(g["ct_mask"])
    assert np.array_equal(np.stack(rects), g["ct_rects"]) and np.array_equal(np.stack(boxes), g["ct_boxes"])
    assert len(boxes) == 3 and g["ev_tp"].tolist() == [True, True, False] and g["ev_fp"].tolist() == [False, False, True]
    traced = OC.suzuki_contours(g["ct_mask"])       # the lower-right block was found last, so it comes first
    assert [k for k, _, _ in traced] == g["ct_kinds"].tolist() == [False, False, True]
    assert [p for _, _, p in traced] == g["ct_parents"].tolist() == [-1, -1, 1]
    assert np.array_equal(C.resize_cubic_f32(g["cub_src"] * np.float32(255), 45, 80), g["cub_up"])
    assert np.array_equal(C.resize_cubic_f32(g["cub_src"], 5, 9), g["cub_down"])


def test_contour_list_order_literal_tracing_equals_label_rule():
    """oracle/contours.py: `suzuki_contours` follows borders pixel by pixel the way OpenCV's
    contours.cpp does (transition scan, head insertion, pre-order walk); `contour_order` derives the
    list from region labels.  Same kinds, same parents, same order; hole contours visit exactly the
    surrounding component's pixels that touch the hole; an outer contour has its component's hull."""
    from scipy import ndimage
    from oracle import contours as OC
    # a hand-checked case: A (with hole H holding island I) above B; discovery A, H, I, B ->
    # top level newest first: B, then A followed by its subtree H, I
    m = np.zeros((12, 12), np.uint8)
    m[1:8, 1:9] = 1
    m[2:7, 2:8] = 0
    m[4, 4] = 1
    m[9:11, 3:6] = 1
    got = OC.suzuki_contours(m)
    assert [(k, p) for k, _, p in got] == [(False, -1), (False, -1), (True, 1), (False, 2)]
    assert sorted(map(tuple, got[0][1])) == sorted((x, y) for y in (9, 10) for x in (3, 4, 5))
    assert [tuple(p) for p in got[3][1]] == [(4, 4)]
    rng = np.random.default_rng(0)
    total = 0
    for t in range(120):
        h, w = int(rng.integers(3, 26)), int(rng.integers(3, 26))
        m = (rng.uniform(size=(h, w)) < rng.choice([0.3, 0.5, 0.7, 0.85, 0.95])).astype(np.uint8)
        if t % 4 == 1:
            m = ndimage.binary_dilation(m).astype(np.uint8)
        if t % 4 == 2:
            m[:] = 0
            for k in range(0, min(h, w) // 2, 2):
                m[k:h - k, k:w - k] = 1
                m[k + 1:h - k - 1, k + 1:w - k - 1] = 0
            m ^= (rng.uniform(size=(h, w)) < 0.03).astype(np.uint8)
        a, b = OC.suzuki_contours(m), OC.contour_order(m)
        assert len(a) == len(b)
        for (hole, pts, par), (kind, dpts, dpar) in zip(a, b):
            total += 1
            assert hole == (kind == "hole") and par == dpar
            sa, sb = set(map(tuple, pts)), set(map(tuple, dpts))
            if hole:
                assert sa == sb
            else:
                assert sa <= sb
                assert np.array_equal(C.min_area_rect(np.unique(pts, axis=0))[0], C.min_area_rect(dpts)[0])
    assert total > 800


def test_resize_cubic_known_properties():
    """cv2.resize INTER_CUBIC restatement: Keys' kernel with A = -0.75 (coefficients at x = 0.5 are
    -3/32, 19/32, 19/32, -3/32), identity at equal size, constants preserved, clamped borders."""
    src = np.zeros((1, 8), np.float32)
    src[0, 3] = 32.0
    up = C.resize_cubic_f32(src, 1, 16)              # 2x: sample positions k/2 - 0.25 -> fx in {0.75, 0.25}
    c = np.float32(-0.75)
    def coeffs(x):
        x = np.float32(x)
        c0 = ((c * (x + 1) - 5 * c) * (x + 1) + 8 * c) * (x + 1) - 4 * c
        c1 = ((c + 2) * x - (c + 3)) * x * x + 1
        c2 = ((c + 2) * (1 - x) - (c + 3)) * (1 - x) * (1 - x) + 1
        return np.array([c0, c1, c2, np.float32(1) - c0 - c1 - c2], np.float32)
    assert np.allclose(coeffs(0.5), [-3 / 32, 19 / 32, 19 / 32, -3 / 32])
    # dx = 6: fx = 2.75 -> sx = 2, taps 1..4 -> the impulse at 3 meets coefficient 2 of x = 0.75
    assert up[0, 6] == np.float32(32) * coeffs(0.75)[2]
    assert up[0, 7] == np.float32(32) * coeffs(0.25)[1]
    assert up[0, 9] == np.float32(32) * coeffs(0.25)[0]  # dx = 9: sx = 4, tap 0 is column 3
    rng = np.random.default_rng(1)
    s = rng.uniform(size=(7, 9)).astype(np.float32)
    assert np.array_equal(C.resize_cubic_f32(s, 7, 9), s)
    assert np.abs(C.resize_cubic_f32(np.full((4, 5), 0.3, np.float32), 13, 17) - 0.3).max() < 1e-6
    one = C.resize_cubic_f32(np.array([[2.0]], np.float32), 3, 3)          # every tap clamps onto the pixel
    assert np.abs(one - 2.0).max() < 1e-6
