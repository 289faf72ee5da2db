"""Locality-aware NMS on the GPU (EAST, Zhou et al. 2017, Algorithm 1).  The reference tree has no
NMS; the entry point keeps the name EAST-style drivers use
(`lanms.merge_quadrangle_n9(boxes, nms_thres)`)."""
import numpy as np
import torch

from .. import ops
from ..graph import F32, get_default_graph


def lanms_batch(boxes, counts, iou_thresh=0.2, graph=None):
    """boxes: f32 [n_images, max_k, 9] (device or host), counts: int32 [n_images].
    Returns (merged [n,max_k,9], n_merged [n], keep_idx [n,max_k], n_keep [n]) device tensors."""
    g = graph or get_default_graph()
    b = boxes if isinstance(boxes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(boxes, np.float32))
    c = counts if isinstance(counts, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(counts, np.int32))
    b = b.to(device=g.device, dtype=F32).contiguous()
    c = c.to(device=g.device, dtype=torch.int32).contiguous()
    n, k, _ = b.shape
    merged = torch.empty_like(b)
    n_merged = torch.empty((n,), dtype=torch.int32, device=g.device)
    keep = torch.empty((n, k), dtype=torch.int32, device=g.device)
    n_keep = torch.empty((n,), dtype=torch.int32, device=g.device)
    ops.lanms(b, c, float(iou_thresh), merged, n_merged, keep, n_keep, g.workspace())
    return merged, n_merged, keep, n_keep


def merge_quadrangle_n9(polys, thres=0.3, graph=None):
    """One image: polys [k,9] -> kept quads [m,9] (merged coordinates, summed scores)."""
    polys = np.ascontiguousarray(polys, np.float32)
    if polys.shape[0] == 0:
        return polys.reshape(0, 9)
    merged, n_merged, keep, n_keep = lanms_batch(polys[None], np.array([polys.shape[0]], np.int32), thres, graph)
    nk = int(n_keep[0].item())
    idx = keep[0, :nk].long()
    return merged[0][idx].cpu().numpy()
