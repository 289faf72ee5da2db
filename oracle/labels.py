"""ORACLE (test infrastructure only): CPU restatement of the reference's two label generators on top
of the fillPoly restatement in cvgeom_oracle.c.

  icdar_generate_rbox      datasets/icdar.py:486-539 (`generate_rbox`), incl. `valid_link` :83-105 with
                           its transposed direction naming ((x, y) points, 'up' = x-1) and numpy's
                           wrap-around at index -1, plus the generator's `[::4, ::4]` subsample and
                           float casts (:632-634)
  pixellink_generate_rbox  tool/pixellink_fn.py:53-110 (`generate_rbox`) with `valid_link` :9-47 (true
                           directions, label equality) and the INTER_NEAREST 1/4 resize (:84-85)

PARITY UNPINNED against cv2 (absent); polygons are processed sequentially exactly as the reference
does (the GPU kernels use an order-free closed form and are pinned against this file bit for bit).
"""
import numpy as np

from . import cvgeom

# channel -> (dx, dy) of the neighbour looked at.  icdar.py:522-537 names the channels left,
# left_down, left_up, right, right_down, right_up, up, down but `valid_link` builds point_dir from
# point = [x, y] with 'up' = point[0]-1 ..., i.e. the x / y roles are swapped:
ICDAR_DIRS = [(0, -1), (1, -1), (-1, -1), (0, 1), (1, 1), (-1, 1), (-1, 0), (1, 0)]
# tool/pixellink_fn.py:94-108 with its valid_link (true directions)
PIXELLINK_DIRS = [(-1, 0), (-1, 1), (-1, -1), (1, 0), (1, 1), (1, -1), (0, -1), (0, 1)]


def icdar_generate_rbox(im_size, polys, tags, min_text_size=10):
    """datasets/icdar.py:486-539.  polys float32 [k,4,2], tags bool [k] -> (score_map uint8 [h,w],
    geo_map float32 [h,w,8], training_mask uint8 [h,w]) at full resolution."""
    h, w = im_size
    poly_mask = np.zeros((h, w), np.uint8)
    score_map = np.zeros((h, w), np.uint8)
    geo_map = np.zeros((h, w, 8), np.float32)
    training_mask = np.ones((h, w), np.uint8)
    for poly_idx, (poly, tag) in enumerate(zip(polys, tags)):
        poly = np.asarray(poly, np.float32)
        ipoly = poly.astype(np.int32)
        cvgeom.fill_poly(score_map, ipoly, 1)
        cvgeom.fill_poly(poly_mask, ipoly, poly_idx + 1)
        poly_h = min(np.linalg.norm(poly[0] - poly[3]), np.linalg.norm(poly[1] - poly[2]))
        poly_w = min(np.linalg.norm(poly[0] - poly[1]), np.linalg.norm(poly[2] - poly[3]))
        if min(poly_h, poly_w) < min_text_size:
            cvgeom.fill_poly(training_mask, ipoly, 0)
        if tag:
            cvgeom.fill_poly(training_mask, ipoly, 0)
        for y, x in np.argwhere(poly_mask == (poly_idx + 1)):
            for c, (dx, dy) in enumerate(ICDAR_DIRS):
                if x == h - 1 or y == w - 1:                 # valid_link's (transposed) border rule
                    v = 1
                else:
                    # numpy indexing: -1 wraps to the last row / column; beyond the end raises
                    v = 1 if (score_map[y, x] == 1 and score_map[y + dy, x + dx] == 1) else 0
                geo_map[y, x, c] = v
    return score_map, geo_map, training_mask


def icdar_labels(im_size, polys, tags, min_text_size=10):
    """generate_rbox + the generator's subsample (icdar.py:632-634): score [h/4,w/4,1], geo
    [h/4,w/4,8], training mask [h/4,w/4,1], all float32."""
    s, g, m = icdar_generate_rbox(im_size, polys, tags, min_text_size)
    return (s[::4, ::4, np.newaxis].astype(np.float32), g[::4, ::4, :].astype(np.float32),
            m[::4, ::4, np.newaxis].astype(np.float32))


def _resize_nearest(img, new_w, new_h):
    """cv2.resize(..., INTER_NEAREST): src index = min(floor(dst * (1 / (dsize/ssize))), ssize-1)."""
    h, w = img.shape
    ifx = 1.0 / (float(new_w) / w)
    ify = 1.0 / (float(new_h) / h)
    sx = np.minimum(np.floor(np.arange(new_w) * ifx).astype(np.int64), w - 1)
    sy = np.minimum(np.floor(np.arange(new_h) * ify).astype(np.int64), h - 1)
    return img[sy][:, sx]


def pixellink_generate_rbox(h, w, xs, ys, bboxes, ignored):
    """tool/pixellink_fn.py:53-110.  xs, ys normalised [k,4]; returns (score float32 [h/4,w/4], link
    float32 [h/4,w/4,8], show_bboxes float32 [200,4])."""
    assert len(xs) == len(ignored)
    new_h, new_w = h // 4, w // 4
    score_map = np.zeros((h, w), np.uint8)          # float 0/1 in the reference; same pixel set
    res_link_map = np.zeros((new_h, new_w, 8), np.float32)
    poly_mask = np.zeros((h, w), np.uint8)
    show_bboxes = np.zeros((200, 4), np.float32)
    xs = np.asarray(xs, np.float32)
    ys = np.asarray(ys, np.float32)
    num_rects = xs.shape[0]
    for idx in range(num_rects):
        points = list(zip(xs[idx, :] * w, ys[idx, :] * h))
        show_bboxes[idx, :] = bboxes[idx, :]
        draw_poly = np.array([points], np.int32)
        cvgeom.fill_poly(score_map, draw_poly[0], 1)
        cvgeom.fill_poly(poly_mask, draw_poly[0], idx + 1)      # uint8 image: colour saturates at 255
    res_score_map = _resize_nearest(score_map, new_w, new_h).astype(np.float32)
    poly_mask = _resize_nearest(poly_mask, new_w, new_h)
    for poly_idx in range(num_rects):
        val = poly_idx + 1
        for y, x in np.argwhere(poly_mask == val):
            for c, (dx, dy) in enumerate(PIXELLINK_DIRS):
                if x == new_w - 1 or y == new_h - 1 or x == 0 or y == 0:
                    v = 1.0
                else:
                    v = 1.0 if poly_mask[y + dy, x + dx] == val else 0.0
                res_link_map[y, x, c] = v
    return res_score_map, res_link_map, show_bboxes
