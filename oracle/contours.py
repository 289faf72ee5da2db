"""ORACLE (test infrastructure only): the boxes test.py:182-190 derives from
`cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)` + `cv2.minAreaRect` + `cv2.boxPoints`,
restated by DEFINITION rather than by Suzuki's border following: a contour's rectangle depends only on
the convex hull of its points; the outer contour of an 8-connected component has the component's own
hull, and a hole's contour is the set of 1-pixels of the surrounding component with a 4-neighbour in
the hole (Suzuki & Abe 1985, border points for 8-connected foreground; holes = 4-connected 0-regions
not reaching the frame; pixels of islands inside the hole lie within that contour's hull, so
including them — as the GPU kernel does — leaves the rectangle unchanged).
Regions are labelled with scipy.ndimage, rectangles come from cvgeom_oracle.c.  PARITY UNPINNED
against cv2 (absent).

The ORDER of OpenCV's contour list (RETR_TREE) is restated twice: `suzuki_contours` follows borders
literally (Suzuki & Abe's Algorithm 1 in the form OpenCV's contours.cpp gives it: a transition scan
over a zero-padded copy, icvFetchContour's clockwise start / counter-clockwise walk and its
"right bound" marking, parents through the last border met on the row, cvInsertNodeIntoTree's
insertion at the HEAD of the parent's child list, cvTreeToNodeSeq's pre-order walk), and
`contour_order` derives the same order from region labels alone (discovery key = raster index of a
component's first pixel / a hole's first 0-pixel; parent = the region left of that pixel).  The
tests hold the two against each other."""
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
        y0, x0 = divmod(int(np.flatnonzero(reg)[0]), w)
        ys, xs = np.nonzero(near & (lab == lab[y0, x0 - 1]))   # the SURROUNDING component's pixels only: islands
        out.append(("hole", np.stack([xs, ys], 1)))            # inside the hole have outer borders of their own
    return out


_DX = (1, 1, 0, -1, -1, -1, 0, 1)          # icvCodeDeltas: E, NE, N, NW, W, SW, S, SE (y grows downwards)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def suzuki_contours(mask):
    """Literal border following.  Returns the contours in OpenCV's output order as
    [(is_hole, points [k,2] (x,y) of EVERY border pixel visited, parent position in the list or -1)]."""
    m = (np.asarray(mask) != 0)
    h, w = m.shape
    f = np.zeros((h + 2, w + 2), np.int64)
    f[1:-1, 1:-1] = m
    borders = [dict(hole=True, parent=None, pts=None, children=[])]        # number 1 = the frame (a hole border)
    for y in range(1, h + 1):
        lnbd = 1
        x = 1
        while x <= w + 1:
            prev, p = f[y, x - 1], f[y, x]
            is_hole = None
            if prev == 0 and p == 1:
                is_hole, ox = False, x
            elif p == 0 and prev >= 1:
                is_hole, ox = True, x - 1
                if prev > 1:
                    lnbd = int(prev)
            if is_hole is not None:
                nbd = len(borders) + 1
                b = borders[lnbd - 1]
                parent = (b["parent"] if b["hole"] == is_hole else lnbd)
                pts = []
                s_end = s = 0 if is_hole else 4
                found = False
                while True:
                    s = (s - 1) & 7
                    if f[y + _DY[s], ox + _DX[s]] != 0:
                        found = True
                        break
                    if s == s_end:
                        break
                if not found:
                    f[y, ox] = -nbd
                    pts.append((ox, y))
                else:
                    i1 = (ox + _DX[s], y + _DY[s])
                    i3 = (ox, y)
                    while True:
                        s_end = s
                        while True:
                            s += 1
                            i4 = (i3[0] + _DX[s & 7], i3[1] + _DY[s & 7])
                            if f[i4[1], i4[0]] != 0:
                                break
                        s &= 7
                        if ((s - 1) & 0xFFFFFFFF) < s_end:              # the pixel to the right was examined as 0
                            f[i3[1], i3[0]] = -nbd
                        elif f[i3[1], i3[0]] == 1:
                            f[i3[1], i3[0]] = nbd
                        pts.append(i3)
                        if i4 == (ox, y) and i3 == i1:
                            break
                        i3 = i4
                        s = (s + 4) & 7
                borders.append(dict(hole=is_hole, parent=parent, pts=np.array(pts, np.int64) - 1, children=[]))
                borders[parent - 1]["children"].insert(0, nbd)             # cvInsertNodeIntoTree: new head
            v = f[y, x]
            if v != 0 and v != 1:
                lnbd = int(abs(v))
            x += 1
    out, pos = [], {}

    def walk(k, parent_pos):                                               # cvTreeToNodeSeq: pre-order
        for c in borders[k - 1]["children"]:
            pos[c] = len(out)
            out.append((borders[c - 1]["hole"], borders[c - 1]["pts"], parent_pos))
            walk(c, pos[c])
    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    walk(1, -1)
    return out


def contour_order(mask):
    """The same list from region labels: [(kind, points, parent position)] in OpenCV's order."""
    m = np.asarray(mask) != 0
    h, w = m.shape
    lab, k = ndimage.label(m, structure=np.ones((3, 3), int))
    zl, kz = ndimage.label(~m, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    sets = contour_point_sets(mask)
    nodes = {}                       # id -> (key, parent id, kind, points); ids: ('c', i) / ('z', i)
    si = 0
    for i in range(1, k + 1):
        first = int(np.flatnonzero(lab == i)[0])
        y, x = divmod(first, w)
        par = None
        if x > 0:
            z = int(zl[y, x - 1])
            reg = zl == z
            if not (reg[0].any() or reg[-1].any() or reg[:, 0].any() or reg[:, -1].any()):
                par = ("z", z)
        nodes[("c", i)] = (first, par, "outer", sets[si][1])
        si += 1
    for i in range(1, kz + 1):
        reg = zl == i
        if reg[0].any() or reg[-1].any() or reg[:, 0].any() or reg[:, -1].any():
            continue
        first = int(np.flatnonzero(reg)[0])
        y, x = divmod(first, w)
        nodes[("z", i)] = (first, ("c", int(lab[y, x - 1])), "hole", sets[si][1])
        si += 1
    kids = {}
    for nid, (key, par, _, _) in nodes.items():
        kids.setdefault(par, []).append((key, nid))
    out = []

    def walk(par, ppos):
        for _, nid in sorted(kids.get(par, []), reverse=True):             # last discovered first
            me = len(out)
            out.append((nodes[nid][2], nodes[nid][3], ppos))
            walk(nid, me)
    walk(None, -1)
    return out


def contour_boxes(mask):
    """Rectangles and integer boxes in OpenCV's contour-list order."""
    rects, boxes = [], []
    for _, pts, _ in contour_order(mask):
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
