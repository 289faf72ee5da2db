"""ORACLE (test infrastructure only): the reference's evaluation loop restated literally on the CPU —
tool/bboxes.py np_bboxes_jaccard :246-283 (two filled masks per pair via the fillPoly restatement in
cvgeom_oracle.c, `intersect = sum(a*b)`, `union = sum(a+b >= 1)`) and bboxes_matching :171-240 (the
tf.while_loop body, one detection at a time).  PARITY UNPINNED against cv2 / TF (absent)."""
import numpy as np

from . import cvgeom


def np_bboxes_jaccard(bbox, gxs, gys):
    bbox_points = np.reshape(bbox, (4, 2))
    cnt = np.asarray(bbox_points, np.int32)
    xmax = max(np.max(bbox_points[:, 0]), np.max(gxs)) + 10
    ymax = max(np.max(bbox_points[:, 1]), np.max(gys)) + 10
    bbox_mask = np.zeros((int(ymax), int(xmax)), np.uint8)
    cvgeom.fill_poly(bbox_mask, cnt, 1)
    jaccard = np.zeros((len(gxs),), np.float32)
    for gt_idx, gt_bbox in enumerate(zip(gxs, gys)):
        gt_mask = np.zeros_like(bbox_mask)
        cvgeom.fill_poly(gt_mask, np.asarray(np.transpose(gt_bbox), np.int32), 1)
        intersect = np.sum(bbox_mask * gt_mask)
        union = np.sum(bbox_mask + gt_mask >= 1)
        with np.errstate(divide="ignore", invalid="ignore"):
            jaccard[gt_idx] = intersect * 1.0 / union
    return jaccard


def bboxes_matching(bboxes, gxs, gys, gignored, matching_threshold=0.5):
    gignored = np.asarray(gignored).astype(bool)
    n_gbboxes = int(np.count_nonzero(~gignored))
    gmatch = np.zeros(gignored.shape, bool)
    grange = np.arange(gignored.size)
    tp, fp = [], []
    for i in range(len(bboxes)):
        jaccard = np_bboxes_jaccard(bboxes[i], gxs, gys)
        idxmax = int(np.argmax(jaccard))
        jcdmax = jaccard[idxmax]
        match = jcdmax > matching_threshold
        existing_match = gmatch[idxmax]
        not_ignored = not gignored[idxmax]
        tp.append(not_ignored and match and not existing_match)
        fp.append(not_ignored and (existing_match or not match))
        mask = (grange == idxmax) & (not_ignored and match)
        gmatch = gmatch | mask
    return n_gbboxes, np.array(tp, bool), np.array(fp, bool)
