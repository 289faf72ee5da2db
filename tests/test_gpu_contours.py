"""GPU: the EAST script's decode (test.py:45-74,182-201): `pixel_detect` twin, findContours regions
(components + holes) and one oriented box per contour, against the definitional CPU restatement
(oracle/contours.py) — masks bit-exact, box SETS bit-exact (OpenCV's list order is not reproduced)."""
import numpy as np
import pytest
import torch

from oracle import contours as OC
from oracle import cvgeom

pytestmark = pytest.mark.gpu


def _mask(rng, h, w, blobs=10, holes=True):
    ys, xs = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), bool)
    for _ in range(blobs):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        a, b, th = rng.uniform(3, w / 5), rng.uniform(2, h / 6), rng.uniform(-1, 1)
        u = (xs - cx) * np.cos(th) + (ys - cy) * np.sin(th)
        v = -(xs - cx) * np.sin(th) + (ys - cy) * np.cos(th)
        blob = (np.abs(u) <= a) & (np.abs(v) <= b)
        if holes and rng.uniform() < 0.7:
            blob &= ~(((u / (a * 0.5)) ** 2 + (v / (b * 0.5)) ** 2) <= 1)      # ring: one hole (maybe open at the edge)
        m |= blob
    m ^= rng.uniform(size=(h, w)) < 0.01                                       # salt & pepper: tiny holes / dots
    return m.astype(np.uint8)


def _keyset(boxes):
    return sorted(tuple(np.asarray(b).ravel().tolist()) for b in boxes)


@pytest.mark.parametrize("h,w,seed", [(40, 56, 0), (96, 128, 1), (128, 128, 2), (64, 200, 3)])
def test_contour_boxes_match_definition(device, h, w, seed):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g = Graph(device)
    rng = np.random.default_rng(seed)
    m = _mask(rng, h, w)
    if seed == 0:
        m[:] = 0
        m[5:20, 5:30] = 1
        m[8:12, 8:12] = 0                 # one hole
        m[9:11, 9:11] = 1                 # an island inside the hole
        m[25:35, 0:10] = 1
        m[28:32, 0:4] = 0                 # a notch open to the image edge: NOT a hole
        m[38, 50] = 1                     # a single pixel
    rects, boxes = P.find_contour_boxes(m, graph=g)
    orects, oboxes = OC.contour_boxes(m)
    assert len(boxes) == len(oboxes)
    # same boxes in the same ORDER: OpenCV's contour list (newest sibling first, pre-order)
    assert [tuple(np.asarray(b).ravel().tolist()) for b in boxes] == [tuple(np.asarray(b).ravel().tolist()) for b in oboxes]
    assert [r.tobytes() for r in rects] == [r.tobytes() for r in orects]
    if seed == 0:
        kinds = [k for k, _ in OC.contour_point_sets(m)]
        assert kinds.count("outer") == 4 and kinds.count("hole") == 1


def test_contour_order_of_nested_rings(device):
    """Rings inside rings with islands and gaps: hole -> island -> hole chains several levels deep.  The
    order is checked against the LITERAL border-following restatement, not only the label-based one."""
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g = Graph(device)
    rng = np.random.default_rng(5)
    for h, w in ((33, 41), (48, 48), (25, 60)):
        m = np.zeros((h, w), np.uint8)
        for k in range(0, min(h, w) // 2, 2):
            m[k:h - k, k:w - k] = 1
            m[k + 1:h - k - 1, k + 1:w - k - 1] = 0
        m ^= (rng.uniform(size=(h, w)) < 0.03).astype(np.uint8)
        rects, boxes = P.find_contour_boxes(m, graph=g)
        traced = OC.suzuki_contours(m)
        assert len(traced) == len(boxes) and max(t[2] for t in traced) >= 3        # real nesting
        for (is_hole, pts, _), rect, box in zip(traced, rects, boxes):
            want, _, _ = cvgeom.min_area_rect(np.unique(pts, axis=0))
            assert rect.tobytes() == np.asarray(want, np.float32).tobytes()
            assert np.array_equal(box, cvgeom.box_points(want).astype(np.int64))


def test_empty_and_full_masks(device):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g = Graph(device)
    r, b = P.find_contour_boxes(np.zeros((16, 16), np.uint8), graph=g)
    assert r.shape == (0, 5) and b.shape == (0, 4, 2)
    r, b = P.find_contour_boxes(np.ones((16, 24), np.uint8), graph=g)
    assert len(b) == 1 and _keyset(b) == _keyset(OC.contour_boxes(np.ones((16, 24), np.uint8))[1])


def test_east_pixel_detect_twin(device):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g = Graph(device)
    rng = np.random.default_rng(4)
    h, w = 48, 48
    score = rng.uniform(size=(1, h, w, 1)).astype(np.float32)
    geo = rng.uniform(0.7, 1.0, size=(1, h, w, 16)).astype(np.float32)
    got = P.east_pixel_detect(score, geo, 0.8, 0.8, graph=g).cpu().numpy()
    want = OC.east_pixel_detect(score, geo, 0.8, 0.8)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    assert got.sum() > 100 and (got != (score[0, :, :, 0] > 0.8)).sum() <= 16
    # a channel with fewer than two pixels below the threshold: numpy raises IndexError, so do we
    geo[..., 5] = 0.95
    with pytest.raises(IndexError):
        OC.east_pixel_detect(score, geo, 0.8, 0.8)
    with pytest.raises(IndexError):
        P.east_pixel_detect(score, geo, 0.8, 0.8, graph=g)
