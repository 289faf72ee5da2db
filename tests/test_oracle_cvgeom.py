"""CPU: the OpenCV-geometry restatement (oracle/cvgeom_oracle.c) against brute force and known
answers.  cv2 itself is absent here (PARITY UNPINNED, see the C file's header)."""
import numpy as np

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
