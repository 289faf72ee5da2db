"""Mirror of the evaluation half of the reference's tool/bboxes.py: detection / ground-truth matching
with RASTERISED IoU (`bboxes_matching` :171-240, `bboxes_jaccard` :242-245, `np_bboxes_jaccard`
:246-283).  The reference draws each pair of quadrangles into 0/1 masks with cv2 and counts pixels on
the host, once per detection inside a tf.while_loop; here all pairs of an image are counted in one
launch (ocr_quad_iou) and the sequential greedy matching — O(detections) scalar work — stays on the
host exactly as written."""
import numpy as np
import torch

from .. import ops
from ..graph import get_default_graph


def _pair_counts(bboxes, gxs, gys, graph=None):
    """(inter, union) int32 [n_det, n_gt] pixel counts of the filled rasters."""
    g = graph or get_default_graph()
    dets = np.asarray(bboxes).reshape(len(bboxes), 4, 2).astype(np.int32)         # points_to_contours: int32
    gts = np.stack([np.asarray(gxs), np.asarray(gys)], axis=-1).astype(np.int32)  # [G,4,2]
    # the reference sizes its masks per detection as max coordinate + 10 (:252-256); any size that
    # keeps every vertex inside gives the same raster, so one size serves all pairs
    xmax = int(max(dets[:, :, 0].max(), gts[:, :, 0].max())) + 10
    ymax = int(max(dets[:, :, 1].max(), gts[:, :, 1].max())) + 10
    if xmax <= 0 or ymax <= 0:
        raise ValueError("negative dimensions are not allowed")                    # np.zeros in util.img.black
    d_d = torch.from_numpy(dets).to(g.device)
    d_g = torch.from_numpy(gts).to(g.device)
    inter = torch.empty((len(dets), len(gts)), dtype=torch.int32, device=g.device)
    uni = torch.empty_like(inter)
    ops.quad_iou(d_d, d_g, ymax, xmax, inter, uni)
    return inter.cpu().numpy(), uni.cpu().numpy()


def np_bboxes_jaccard(bbox, gxs, gys, graph=None):
    """tool/bboxes.py:246-283: IoU of one detection (8 numbers) with every ground truth -> float32 [G]."""
    inter, uni = _pair_counts(np.reshape(bbox, (1, 8)), gxs, gys, graph)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter[0] * 1.0 / uni[0]).astype(np.float32)


bboxes_jaccard = np_bboxes_jaccard          # :242-245 (the tf.py_func wrapper)


def bboxes_matching(bboxes, gxs, gys, gignored, matching_threshold=0.5, scope=None, graph=None):
    """tool/bboxes.py:171-240.  bboxes [N,8] detections sorted by score, gxs/gys [G,4], gignored [G].
    Returns (n_gbboxes, tp_match bool [N], fp_match bool [N]): a detection is a TP when its best
    ground truth (first argmax) has IoU > threshold, is not ignored and was not matched before; an FP
    when that ground truth is not ignored and (already matched or IoU too low); neither when ignored."""
    bboxes = np.asarray(bboxes).reshape(-1, 8)
    gignored = np.asarray(gignored).astype(bool)
    n_gbboxes = int(np.count_nonzero(~gignored))
    n = len(bboxes)
    tp = np.zeros(n, bool)
    fp = np.zeros(n, bool)
    if n == 0:
        return n_gbboxes, tp, fp
    inter, uni = _pair_counts(bboxes, gxs, gys, graph)
    with np.errstate(divide="ignore", invalid="ignore"):
        jac = (inter * 1.0 / uni).astype(np.float32)
    gmatch = np.zeros(gignored.shape, bool)
    for i in range(n):
        idxmax = int(np.argmax(jac[i]))
        match = jac[i, idxmax] > matching_threshold
        existing_match = gmatch[idxmax]
        not_ignored = not gignored[idxmax]
        tp[i] = not_ignored and match and not existing_match
        fp[i] = not_ignored and (existing_match or not match)
        gmatch[idxmax] = gmatch[idxmax] or (not_ignored and match)
    return n_gbboxes, tp, fp
