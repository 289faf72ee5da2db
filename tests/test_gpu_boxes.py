"""GPU: one oriented box per component (ocr_min_area_rects + host RotatedRect formatting) vs the
OpenCV restatement in oracle/cvgeom_oracle.c — hull sizes, calipers output and the integer corner
points bit-exact (test_pixellink_fast.py:193-202)."""
import numpy as np
import pytest
import torch

from oracle import cvgeom as C

pytestmark = pytest.mark.gpu


def _label_maps(rng, n, h, w, shapes):
    lab = np.zeros((n, h, w), np.int32)
    ncomp = np.zeros(n, np.int32)
    ys, xs = np.mgrid[0:h, 0:w]
    for b in range(n):
        k = 0
        for kind in shapes:
            cy, cx = rng.uniform(5, h - 5), rng.uniform(5, w - 5)
            if kind == "rot":
                a, bb, th = rng.uniform(3, w / 6), rng.uniform(1, 6), rng.uniform(-np.pi, np.pi)
                u = (xs - cx) * np.cos(th) + (ys - cy) * np.sin(th)
                v = -(xs - cx) * np.sin(th) + (ys - cy) * np.cos(th)
                m = (np.abs(u) <= a) & (np.abs(v) <= bb)
            elif kind == "ell":
                a, bb, th = rng.uniform(2, w / 8), rng.uniform(1, 8), rng.uniform(-np.pi, np.pi)
                u = (xs - cx) * np.cos(th) + (ys - cy) * np.sin(th)
                v = -(xs - cx) * np.sin(th) + (ys - cy) * np.cos(th)
                m = (u / a) ** 2 + (v / bb) ** 2 <= 1
            elif kind == "row":
                m = (ys == int(cy)) & (np.abs(xs - cx) < rng.uniform(1, 20))
            elif kind == "col":
                m = (xs == int(cx)) & (np.abs(ys - cy) < rng.uniform(1, 20))
            elif kind == "diag":
                m = (xs - int(cx) == ys - int(cy)) & (np.abs(xs - cx) < rng.uniform(2, 15))
            elif kind == "dot":
                m = (ys == int(cy)) & (xs == int(cx))
            elif kind == "box":
                m = (np.abs(xs - cx) <= rng.integers(1, 12)) & (np.abs(ys - cy) <= rng.integers(1, 12))
            else:   # noise blob
                m = ((xs - cx) ** 2 + (ys - cy) ** 2 <= rng.uniform(3, 10) ** 2) & (rng.uniform(size=(h, w)) < 0.5)
            m &= lab[b] == 0
            if not m.any():
                continue
            k += 1
            lab[b][m] = k
        ncomp[b] = k
    return lab, ncomp


@pytest.mark.parametrize("h,w,sx,sy,seed", [(64, 96, 4.0, 4.0, 0), (192, 320, 1280.0 / 320, 720.0 / 192, 1),
                                            (256, 256, 4.0, 4.0, 2), (48, 40, 1.0, 1.0, 3), (40, 64, 2.5, 3.3, 4)])
def test_min_area_rect_boxes_bit_exact(device, h, w, sx, sy, seed):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g = Graph(device)
    rng = np.random.default_rng(seed)
    shapes = ["rot"] * 6 + ["ell"] * 3 + ["row", "col", "diag", "dot", "box", "box", "blob", "blob"]
    lab, ncomp = _label_maps(rng, 3, h, w, shapes)
    lab[2] = 0
    ncomp[2] = 0                                   # an image without components
    out = P.min_area_rect_boxes(torch.from_numpy(lab), torch.from_numpy(ncomp), sx, sy, max_comps=64, graph=g)
    checked = 0
    for b in range(3):
        rects, boxes = out[b]
        assert len(rects) == ncomp[b]
        for i in range(1, ncomp[b] + 1):
            xy_in_poly = np.argwhere(lab[b] == i)
            show_xy = xy_in_poly.copy()
            show_xy[:, 0] = xy_in_poly[:, 1] * sx          # the reference's integer assignment
            show_xy[:, 1] = xy_in_poly[:, 0] * sy
            rect, cal, hull = C.min_area_rect(show_xy)
            assert rects[i - 1].tobytes() == rect.tobytes(), (b, i, rects[i - 1], rect, hull)
            want = C.box_points(rect).astype(np.int64)
            assert np.array_equal(boxes[i - 1], want)
            checked += 1
    assert checked > 20


def test_min_area_rects_raw_outputs_and_errors(device):
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd._lib import OcrHipError
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    lab = np.zeros((1, 16, 16), np.int32)
    lab[0, 2:5, 3:9] = 1
    lab[0, 10, 10] = 3                            # id 2 has no pixels
    t = torch.from_numpy(lab).to(device)
    nc = torch.tensor([3], dtype=torch.int32, device=device)
    hn = torch.full((1, 8), -7, dtype=torch.int32, device=device)
    hd = torch.zeros((1, 8, 4), dtype=torch.int32, device=device)
    cal = torch.zeros((1, 8, 6), dtype=torch.float32, device=device)
    ops.min_area_rects(t, nc, 8, 4.0, 4.0, hn, hd, cal, g.workspace())
    assert hn[0, :3].tolist() == [4, 0, 1] and hn[0, 3].item() == -7      # entries beyond ncomp untouched
    assert hd[0, 0].tolist() == [12, 8, 12, 16]       # (min X, min Y) first, then towards increasing Y
    assert hd[0, 2, :2].tolist() == [40, 40]
    with pytest.raises(OcrHipError):
        ops.min_area_rects(t, nc, 8, 0.5, 4.0, hn, hd, cal, g.workspace())   # shrinking scale unsupported
