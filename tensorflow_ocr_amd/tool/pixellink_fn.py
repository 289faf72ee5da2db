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
                   min_size=10, max_comps=4096, graph=None):
    """test_pixellink_fast.py:110-178 for a whole batch.  pixel_score [N,h,w] = softmax(pixel_cls)
    [...,1]; link_score [8,N,h,w,2] (stacked softmaxes) or [8,N,h,w].  Returns (labels int32
    [N,h,w], ncomp int32 [N], comps int32 [N,max_comps,2] = (smallest pixel index, size))."""
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
    return labels, ncomp, comps
