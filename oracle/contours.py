"""ORACLE (test infrastructure only): the boxes test.py:182-190 derives from
`cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)` + `cv2.minAreaRect` + `cv2.boxPoints`,
restated by DEFINITION rather than by Suzuki's border following: a contour's rectangle depends only on
the convex hull of its points; the outer contour of an 8-connected component has the component's own
hull, and a hole's contour is the set of 1-pixels with a 4-neighbour in the hole (Suzuki & Abe 1985,
border points for 8-connected foreground; holes = 4-connected 0-regions not reaching the frame).
Regions are labelled with scipy.ndimage, rectangles come from cvgeom_oracle.c.  PARITY UNPINNED
against cv2 (absent); the order of OpenCV's contour list is not reproduced (set comparison)."""
import numpy as np
from scipy import ndimage

from . import cvgeom


def east_pixel_detect(score_map, geo_map, score_map_thresh=0.8, link_thresh=0.8):
    """test.py:45-74, literally."""
    if len(score_map.shape) == 4:
        score_map = score_map[0, :, :, 0]
        geo_map = geo_map[0, :, :, ]
    res_map = np.zeros((score_map.shape[0], score_map.shape[1]))
    xy_text = np.argwhere(score_map > score_map_thresh)
    for p in xy_text:
        res_map[p[0], p[1]] = 1
    res = res_map
    for i in range(8):
        geo_map_split = geo_map[:, :, i * 2 + 1]
        link_text = np.argwhere(geo_map_split < link_thresh)
        res[link_text[0], link_text[1]] = 0
    return np.array(res_map, dtype=np.uint8)


def contour_point_sets(mask):
    """[(kind, points [k,2] as (x,y))]: outer contours' defining sets, then hole borders."""
    m = np.asarray(mask) != 0
    h, w = m.shape
    out = []
    lab, k = ndimage.label(m, structure=np.ones((3, 3), int))
    for i in range(1, k + 1):
        ys, xs = np.nonzero(lab == i)
        out.append(("outer", np.stack([xs, ys], 1)))
    zl, kz = ndimage.label(~m, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    for i in range(1, kz + 1):
        reg = zl == i
        if reg[0].any() or reg[-1].any() or reg[:, 0].any() or reg[:, -1].any():
            continue                                       # reaches the frame: background
        near = np.zeros_like(reg)
        near[1:] |= reg[:-1]
        near[:-1] |= reg[1:]
        near[:, 1:] |= reg[:, :-1]
        near[:, :-1] |= reg[:, 1:]
        ys, xs = np.nonzero(near & m)
        out.append(("hole", np.stack([xs, ys], 1)))
    return out


def contour_boxes(mask):
    rects, boxes = [], []
    for _, pts in contour_point_sets(mask):
        rect, _, _ = cvgeom.min_area_rect(pts)
        rects.append(rect)
        boxes.append(cvgeom.box_points(rect).astype(np.int64))     # np.int0
    return rects, boxes


def order_points(pts):
    """test.py:24-35."""
    from scipy.spatial import distance as dist
    x_sorted = pts[np.argsort(pts[:, 0]), :]
    left_most = x_sorted[:2, :]
    right_most = x_sorted[2:, :]
    left_most = left_most[np.argsort(left_most[:, 1]), :]
    (tl, bl) = left_most
    D = dist.cdist(tl[np.newaxis], right_most, 'euclidean')[0]
    (br, tr) = right_most[np.argsort(D)[::-1], :]
    return np.array([tl, tr, br, bl], dtype='int32')
