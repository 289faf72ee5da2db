"""Mirror of reference tool/pixellink_fn.py: `pixel_detect` / `tf_pixel_detect` (:120-158) and the
batched link-gated connected-component decode of test_pixellink_fast.py:110-178, on the GPU."""
import torch

from .. import ops
from ..graph import F32, get_default_graph


def _dev(g, t):
    if hasattr(t, "data") and not isinstance(t, torch.Tensor):
        t = t.data
    if not isinstance(t, torch.Tensor):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    return t.to(device=g.device, dtype=F32).contiguous()


def link_scores(link_cls, graph=None):
    """tf.stack([softmax(link_cls[..., 2i:2i+2]) for i in range(8)]) -> [8,N,h,w,2]
    (test_pixellink_fast.py:55-64)."""
    g = graph or get_default_graph()
    x = _dev(g, link_cls)
    n, h, w, _ = x.shape
    out = g.empty((8, n, h, w, 2), F32)
    ops.link_softmax_stack(x, out)
    return out


def pixel_scores(pixel_cls, graph=None):
    """slim.softmax(pixel_cls) -> [N,h,w,2] (nets/pixellink.py:71)."""
    g = graph or get_default_graph()
    x = _dev(g, pixel_cls)
    out = g.empty(x.shape, F32)
    ops.softmax_pairs(x, out)
    return out


def pixel_detect(score_map, geo_map, score_map_thresh=0.8, link_thresh=0.8, graph=None):
    """tool/pixellink_fn.py:120-154.  score_map [N,h,w,1] (P(text)), geo_map [8,N,h,w,2] (stacked
    link softmaxes) -> uint8 [h,w] for batch element 0."""
    g = graph or get_default_graph()
    s = _dev(g, score_map)
    lk = _dev(g, geo_map)
    if s.dim() != 4 or lk.dim() != 5 or lk.shape[0] != 8:
        raise ValueError("score_map must be [N,h,w,1] and geo_map [8,N,h,w,2]")
    n, h, w, _ = s.shape
    mask = torch.empty((h, w), dtype=torch.uint8, device=g.device)
    ops.pixel_detect(s, lk, n, h, w, float(score_map_thresh), float(link_thresh), mask)
    return mask


def tf_pixel_detect(score_map, geo_map, score_map_thresh, link_thresh, graph=None):
    """tool/pixellink_fn.py:156-158 (the tf.py_func wrapper): same call signature."""
    return pixel_detect(score_map, geo_map, score_map_thresh, link_thresh, graph=graph)


def link_cc_decode(pixel_score, link_score, pixel_conf_threshold=0.8, link_conf_threshold=0.9,
                   min_size=10, max_comps=4096, graph=None, mode="union", key_order="py27"):
    """test_pixellink_fast.py:110-178 for a whole batch.  pixel_score [N,h,w] = softmax(pixel_cls)
    [...,1]; link_score [8,N,h,w,2] (stacked softmaxes) or [8,N,h,w].  Returns (labels int32
    [N,h,w], ncomp int32 [N], comps int32 [N,max_comps,2] = (smallest pixel index, size)).

    mode="union" (default): weakly-connected components of the link graph (order independent; equals the
    script's result whenever the link predictions are symmetric).  mode="reference_dfs": the script's own
    rule, exactly — directed reachability from each unassigned key in the order its `for i in graph.keys()`
    meets them, groups of more than min_size pixels only (:153-178); comps = (seed pixel, size).
    key_order="py27": the iteration order of the script's Python-2 dict (keys inserted x-major; restated
    from CPython 2.7's dictobject.c, a HOST step: the segment mask crosses to the host, ~1 us per key, the order
    comes back); "ascending": ascending pixel index.  reference_dfs runs ONE workgroup per image in rounds of
    full-image sweeps — ~1 ms per 256 x 256 map, seconds per image at 720 x 1280: an exactness mode, opt-in."""
    if mode not in ("union", "reference_dfs"):
        raise ValueError("mode must be 'union' or 'reference_dfs'")
    if key_order not in ("py27", "ascending"):
        raise ValueError("key_order must be 'py27' or 'ascending'")
    g = graph or get_default_graph()
    ps = _dev(g, pixel_score)
    lk = _dev(g, link_score)
    n, h, w = ps.shape
    stride, off = (2, 1) if lk.dim() == 5 else (1, 0)
    labels = torch.empty((n, h, w), dtype=torch.int32, device=g.device)
    ncomp = torch.empty((n,), dtype=torch.int32, device=g.device)
    comps = torch.zeros((n, max_comps, 2), dtype=torch.int32, device=g.device)
    ops.link_cc(ps, lk, stride, off, n, h, w, float(pixel_conf_threshold), float(link_conf_threshold),
                int(min_size), labels, ncomp, comps, g.workspace())
    if mode == "union":
        return labels, ncomp, comps
    order = None
    if key_order == "py27":
        # the dict holds the interior pixels with pixel_score > threshold: the scores cross to the host, the order comes back
        order = ops.py27_dict_order(ps.cpu().contiguous(), float(pixel_conf_threshold)).to(g.device)
    labels_d = torch.empty_like(labels)
    ncomp_d = torch.empty_like(ncomp)
    comps_d = torch.zeros_like(comps)
    ops.link_cc_directed(ps, lk, stride, off, n, h, w, float(pixel_conf_threshold), float(link_conf_threshold),
                         int(min_size), labels, ncomp, labels_d, ncomp_d, comps_d, g.workspace(), seed_order=order)
    return labels_d, ncomp_d, comps_d


def resize_scores_cubic(pixel_score, link_score, out_h=720, out_w=1280, graph=None):
    """test_pixellink.py:95-109: the eight link maps are `b_score[0,:,:,1] * 255` up-sampled with
    cv2.resize(..., (1280, 720), INTER_CUBIC); the pixel map is up-sampled first and multiplied by 255
    after.  pixel_score [N,h,w]; link_score [8,N,h,w,2] or [8,N,h,w].  Returns f32 [9,N,out_h,out_w]
    (plane 0 = pixel, 1..8 = links), one launch per scaling order."""
    g = graph or get_default_graph()
    ps = _dev(g, pixel_score)
    lk = _dev(g, link_score)
    if lk.dim() == 5:
        lk = lk[..., 1]
    n, h, w = ps.shape
    src = torch.empty((9, n, h, w), dtype=F32, device=g.device)
    src[0].copy_(ps)
    src[1:].copy_(lk)
    up = torch.empty((9, n, out_h, out_w), dtype=F32, device=g.device)
    ops.resize_cubic_f32(src[0], up[0], 1.0, 255.0)
    ops.resize_cubic_f32(src[1:].reshape(8 * n, h, w), up[1:].reshape(8 * n, out_h, out_w), 255.0, 1.0)
    return up


def full_resolution_decode(pixel_score, link_score, out_h=720, out_w=1280, min_size=200, max_comps=4096,
                           graph=None):
    """test_pixellink.py:95-184: cubic up-sampling of the nine score maps to out_w x out_h, then the
    same link-gated grouping as the 1/4-resolution script with the constants of :111,:119
    (`pixel_score > int(255 * 0.8)`, `link > 255 * 0.9`) and `len(index_list) > 200`.
    Returns (labels int32 [N,out_h,out_w], ncomp int32 [N], comps int32 [N,max_comps,2])."""
    g = graph or get_default_graph()
    up = resize_scores_cubic(pixel_score, link_score, out_h, out_w, graph=g)
    return link_cc_decode(up[0], up[1:], float(int(255 * 0.8)), 255 * 0.9, min_size=min_size,
                          max_comps=max_comps, graph=g)


def _rotated_rect(nh, head, cal):
    """Tail of cv2.minAreaRect (OpenCV 3.x `minAreaRect`): RotatedRect (cx, cy, w, h, angle°) from the
    calipers' corner + edge vectors (hull of > 2 points) or from a 1/2-point hull.  float32
    arithmetic with the double-precision sqrt/atan2 steps of the original; O(1) per box."""
    import math
    import numpy as np
    f = np.float32
    cx = cy = w = h = ang = f(0)
    if nh > 2:
        c = [f(v) for v in cal]
        cx = c[0] + (c[2] + c[4]) * f(0.5)
        cy = c[1] + (c[3] + c[5]) * f(0.5)
        w = f(math.sqrt(float(c[2]) * float(c[2]) + float(c[3]) * float(c[3])))
        h = f(math.sqrt(float(c[4]) * float(c[4]) + float(c[5]) * float(c[5])))
        ang = f(math.atan2(float(c[3]), float(c[2])))
    elif nh == 2:
        x0, y0, x1, y1 = [f(v) for v in head]
        cx = (x0 + x1) * f(0.5)
        cy = (y0 + y1) * f(0.5)
        dx, dy = float(x1 - x0), float(y1 - y0)
        w = f(math.sqrt(dx * dx + dy * dy))
        ang = f(math.atan2(dy, dx))
    elif nh == 1:
        cx, cy = f(head[0]), f(head[1])
    ang = f(float(ang * f(180)) / math.pi)
    return np.array([cx, cy, w, h, ang], np.float32)


def _box_points(rect):
    """cv2.boxPoints = RotatedRect::points -> float32 [4,2]."""
    import math
    import numpy as np
    f = np.float32
    cx, cy, w, h, ang = [f(v) for v in rect]
    a_ = float(ang) * math.pi / 180.
    b = f(math.cos(a_)) * f(0.5)
    a = f(math.sin(a_)) * f(0.5)
    p0x = cx - a * h - b * w
    p0y = cy + b * h - a * w
    p1x = cx + a * h - b * w
    p1y = cy - b * h - a * w
    two = f(2)
    return np.array([[p0x, p0y], [p1x, p1y], [two * cx - p0x, two * cy - p0y],
                     [two * cx - p1x, two * cy - p1y]], np.float32)


def min_area_rect_boxes(labels, ncomp, scale_x=4.0, scale_y=4.0, max_comps=4096, graph=None):
    """`cv2.minAreaRect(show_xy)` + `np.int0(cv2.boxPoints(rectangle))` for every component of a batch
    of label maps (test_pixellink_fast.py:193-202): show_xy = (int(x*scale_x), int(y*scale_y)).
    labels int32 [N,h,w], ncomp int32 [N] (link_cc_decode's outputs).  Hull + rotating calipers run
    on the GPU (ocr_min_area_rects); the RotatedRect / corner formatting is O(1) per box here.
    Returns per image (rects float32 [k,5] = cx,cy,w,h,angle; boxes int64 [k,4,2])."""
    import numpy as np
    g = graph or get_default_graph()
    if labels.dim() != 3:
        raise ValueError("labels must be [N,h,w]")
    n = labels.shape[0]
    labels = labels.to(device=g.device, dtype=torch.int32).contiguous()
    ncomp = ncomp.to(device=g.device, dtype=torch.int32).contiguous()
    hull_n = torch.zeros((n, max_comps), dtype=torch.int32, device=g.device)
    head = torch.zeros((n, max_comps, 4), dtype=torch.int32, device=g.device)
    cal = torch.zeros((n, max_comps, 6), dtype=F32, device=g.device)
    ops.min_area_rects(labels, ncomp, max_comps, float(scale_x), float(scale_y), hull_n, head, cal, g.workspace())
    k = np.minimum(ncomp.cpu().numpy(), max_comps)
    kmax = int(k.max()) if n else 0
    hn = hull_n[:, :kmax].cpu().numpy()
    hd = head[:, :kmax].cpu().numpy()
    cl = cal[:, :kmax].cpu().numpy()
    out = []
    for b in range(n):
        rects = np.zeros((int(k[b]), 5), np.float32)
        boxes = np.zeros((int(k[b]), 4, 2), np.int64)
        for i in range(int(k[b])):
            rects[i] = _rotated_rect(int(hn[b, i]), hd[b, i], cl[b, i])
            boxes[i] = _box_points(rects[i]).astype(np.int64)        # np.int0: truncation
        out.append((rects, boxes))
    return out


def generate_rbox(h, w, xs, ys, bboxes, ignored, graph=None):
    """tool/pixellink_fn.py:53-110 for one image.  xs, ys normalised corner coordinates [k,4], bboxes
    [k,4], ignored [k] -> (res_score_map float32 [h/4,w/4], res_link_map float32 [h/4,w/4,8],
    show_bboxes float32 [200,4]) as NumPy arrays, like the reference's py_func."""
    s, l, sb = generate_rbox_batch(h, w, [xs], [ys], [bboxes], [ignored], graph=graph)
    return s[0].cpu().numpy(), l[0].cpu().numpy(), sb[0]


def tf_pixellink_get_rbox(img_size, xs, ys, bboxes, ignored, graph=None):
    """tool/pixellink_fn.py:112-118 (the tf.py_func wrapper of generate_rbox): img_size = (h, w); returns DEVICE tensors
    pixel_map [h/4, w/4], link_map [h/4, w/4, 8] and show_bboxes [200, 4] (the shapes the reference sets)."""
    h, w = img_size
    s, l, sb = generate_rbox_batch(h, w, [xs], [ys], [bboxes], [ignored], graph=graph)
    return s[0], l[0], torch.from_numpy(sb[0]).to(s.device)


def generate_rbox_batch(h, w, xs_list, ys_list, bboxes_list, ignored_list, graph=None):
    """generate_rbox for a batch: device tensors score [n,h/4,w/4], link [n,h/4,w/4,8]; show_bboxes
    [n,200,4] stays a NumPy copy of the inputs (:66,77)."""
    import numpy as np
    g = graph or get_default_graph()
    h, w = int(h), int(w)
    n = len(xs_list)
    P = max([len(x) for x in xs_list] + [1])
    if P > 200:
        raise IndexError("show_bboxes holds 200 rows (tool/pixellink_fn.py:66)")
    polys = np.zeros((n, P, 4, 2), np.int32)
    counts = np.zeros(n, np.int32)
    show_bboxes = np.zeros((n, 200, 4), np.float32)
    for b in range(n):
        xs = np.asarray(xs_list[b], np.float32).reshape(-1, 4)
        ys = np.asarray(ys_list[b], np.float32).reshape(-1, 4)
        if len(xs) != len(ignored_list[b]):
            raise AssertionError('the length of xs and ignored must be the same, but got %s and %s'
                                 % (len(xs), len(ignored_list[b])))
        counts[b] = len(xs)
        if len(xs):
            # points = zip(xs*w, ys*h); np.array([points], np.int32): float32 products, truncated
            polys[b, :len(xs), :, 0] = (xs * w).astype(np.int32)
            polys[b, :len(xs), :, 1] = (ys * h).astype(np.int32)
            show_bboxes[b, :len(xs)] = np.asarray(bboxes_list[b], np.float32).reshape(-1, 4)
    dev = g.device
    cover = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    ops.poly_cover(torch.from_numpy(polys).to(dev), torch.from_numpy(counts).to(dev),
                   torch.zeros((n, P), dtype=torch.uint8, device=dev), h, w, cover)
    nh, nw = h // 4, w // 4
    score = torch.empty((n, nh, nw), dtype=F32, device=dev)
    link = torch.empty((n, nh, nw, 8), dtype=F32, device=dev)
    ops.pixellink_labels(cover, nh, nw, score, link)
    return score, link, show_bboxes


def east_pixel_detect(score_map, geo_map, score_map_thresh=0.8, link_thresh=0.8, graph=None):
    """test.py:45-74, the EAST script's own `pixel_detect` — NOT the same function as
    tool/pixellink_fn.pixel_detect above: after `res = score > t`, each of the 8 link channels only
    contributes `res[link_text[0], link_text[1]] = 0` with link_text = np.argwhere(channel < t_l), i.e.
    the first two below-threshold pixels (y0,x0), (y1,x1) in raster order zero res[y0,y1] and
    res[x0,x1].  Reproduced as written, including numpy's IndexError when a channel has fewer than
    two such pixels or an index falls outside the map.  score_map [1,h,w,1] or [h,w]; geo_map
    [1,h,w,16] or [h,w,16] (softmaxed link pairs).  Returns uint8 [h,w] on the device."""
    import numpy as np
    g = graph or get_default_graph()
    s = _dev(g, score_map)
    lk = _dev(g, geo_map)
    if s.dim() == 4:
        s = s[0, :, :, 0].contiguous()
        lk = lk[0].contiguous()
    h, w = s.shape
    mask = torch.empty((h, w), dtype=torch.uint8, device=g.device)
    fs = torch.full((2, 8), 2 ** 31 - 1, dtype=torch.int32, device=g.device)
    ops.east_pixel_detect(s, lk, h, w, float(score_map_thresh), float(link_thresh), mask, fs)
    first, second = fs.cpu().numpy()
    idx = []
    for i in range(8):
        if second[i] == 2 ** 31 - 1:
            n_found = 0 if first[i] == 2 ** 31 - 1 else 1
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (n_found, n_found))
        (y0, x0), (y1, x1) = divmod(int(first[i]), w), divmod(int(second[i]), w)
        for row, col in ((y0, y1), (x0, x1)):            # res[[y0, x0], [y1, x1]] = 0
            if row >= h or col >= w:
                raise IndexError("index %d is out of bounds" % (row if row >= h else col))
            idx.append(row * w + col)
    ops.zero_pixels(mask, torch.tensor(idx, dtype=torch.int32, device=g.device))
    return mask


def find_contour_boxes(mask, scale_x=1.0, scale_y=1.0, max_regions=4096, graph=None):
    """The boxes test.py:182-190 draws from `cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)`:
    one `cv2.minAreaRect` per contour.  A contour is either the outer border of an 8-connected
    component of 1-pixels or the border of a hole (a 4-connected 0-region that does not reach the
    image edge; its contour = the surrounding 1-pixels with a 4-neighbour in it).  A contour's
    rectangle depends only on the convex hull of its points, so regions are labelled (ocr_mask_cc)
    and hulled (ocr_min_area_rects / ocr_hole_border_rects) on the GPU without tracing borders.
    mask uint8 [h,w].  Returns (rects float32 [k,5], boxes int64 [k,4,2]) in OpenCV's list order:
    border following discovers an outer border at its component's first pixel in raster order and a
    hole border at the hole's first 0-pixel, every new contour becomes the FIRST child of its parent
    (cvInsertNodeIntoTree), and the list is the pre-order walk of that tree (cvTreeToNodeSeq) — so
    siblings come newest first, each followed by its own subtree (ocr_contour_parents gives the
    parent links)."""
    import numpy as np
    g = graph or get_default_graph()
    m = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8))
    m = (m.to(g.device) != 0).to(torch.uint8).contiguous()[None]
    _, h, w = m.shape
    keep = {}
    nodes = []                                   # (discovery key, kind, index, rect)
    for value, conn in ((1, 8), (0, 4)):
        labels = torch.empty((1, h, w), dtype=torch.int32, device=g.device)
        ncomp = torch.empty((1,), dtype=torch.int32, device=g.device)
        comps = torch.zeros((1, max_regions, 2), dtype=torch.int32, device=g.device)
        ops.mask_cc(m, value, conn, labels, ncomp, comps, g.workspace())
        k = int(ncomp[0].item())
        if k > max_regions:
            raise ValueError("more than %d regions" % max_regions)
        keep[value] = (labels, comps, k)
        if k == 0:
            continue
        hull_n = torch.zeros((1, max_regions), dtype=torch.int32, device=g.device)
        head = torch.zeros((1, max_regions, 4), dtype=torch.int32, device=g.device)
        cal = torch.zeros((1, max_regions, 6), dtype=F32, device=g.device)
        if value == 1:
            ops.min_area_rects(labels, ncomp, max_regions, float(scale_x), float(scale_y), hull_n, head, cal, g.workspace())
        else:
            ops.hole_border_rects(m, labels, ncomp, max_regions, float(scale_x), float(scale_y), hull_n, head, cal,
                                  g.workspace())
        hn, hd, cl = hull_n[0, :k].cpu().numpy(), head[0, :k].cpu().numpy(), cal[0, :k].cpu().numpy()
        first = comps[0, :k, 0].cpu().numpy()
        for i in range(k):
            if hn[i] == 0:                       # a 0-region that reaches the edge: background, no contour
                continue
            nodes.append((int(first[i]), value, i, _rotated_rect(int(hn[i]), hd[i], cl[i])))
    if not nodes:
        return np.zeros((0, 5), np.float32), np.zeros((0, 4, 2), np.int64)
    (lab1, comps1, k1), (lab0, comps0, k0) = keep[1], keep[0]
    par_c = torch.zeros((max(k1, 1),), dtype=torch.int32, device=g.device)
    par_z = torch.zeros((max(k0, 1),), dtype=torch.int32, device=g.device)
    ops.contour_parents(lab1, lab0, comps1, k1, comps0, k0, par_c, par_z)
    par_c, par_z = par_c.cpu().numpy(), par_z.cpu().numpy()
    holes = {n[2] for n in nodes if n[1] == 0}
    kids = {}
    for key, value, i, r in nodes:
        if value == 1:                           # outer border: child of the hole it lies in, else top level
            z = int(par_c[i]) - 1
            parent = (0, z) if z in holes else None
        else:                                    # hole border: child of the surrounding component
            parent = (1, int(par_z[i]) - 1)
        kids.setdefault(parent, []).append((key, (value, i), r))
    out_r, out_b = [], []
    stack = [iter(sorted(kids.get(None, []), key=lambda t: -t[0]))]
    while stack:                                 # pre-order, newest sibling first
        nxt = next(stack[-1], None)
        if nxt is None:
            stack.pop()
            continue
        _, nid, r = nxt
        out_r.append(r)
        out_b.append(_box_points(r).astype(np.int64))
        stack.append(iter(sorted(kids.get(nid, []), key=lambda t: -t[0])))
    if not out_r:
        return np.zeros((0, 5), np.float32), np.zeros((0, 4, 2), np.int64)
    return np.stack(out_r), np.stack(out_b)
