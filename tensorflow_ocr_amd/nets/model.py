"""Mirror of reference nets/model.py: `unpool` (:14-15), `mean_image_subtraction` (:18-31),
`dice_coefficient` (:145-159), the OHNM softmax `loss` that multigpu_train.py:32 calls (:204-261)
with its helpers, and `model` (:84-143, ResNet-v1-50 + PixelLink fuse heads)."""
from .. import losses
from ..graph import get_default_graph
from . import model_vgg_16 as _mv

unpool = _mv.unpool
mean_image_subtraction = _mv.mean_image_subtraction
dice_coefficient = _mv.dice_coefficient


def model(images, weight_decay=1e-5, is_training=True, graph=None):
    """nets/model.py:84-143: ResNet-v1-50 trunk + BN'd 1x1 fuse heads -> (pixel_4 [N,H/4,W/4,2],
    link_4 [N,H/4,W/4,16]) logits."""
    from . import resnet_model
    return resnet_model.model_resnet50_pixellink(images, weight_decay, is_training, graph)


def loss(y_true_pixel, y_pred_pixel, y_true_link, y_pred_link, training_mask, graph=None):
    """nets/model.py:204-261: online hard negative mining (k = min(3 n_pos, n_neg) per image, batch
    size from the tensor instead of the hard-coded 14), pixel CE over the selected mask / n_pos,
    8 link CEs weighted by the selected mask (no zero guard, like the reference), total =
    sum(link) + 2 * pixel.  `training_mask` is unused by the reference."""
    g = graph or get_default_graph()
    return losses.softmax_loss(g, y_pred_pixel, y_pred_link, y_true_pixel, y_true_link,
                               pixel_rule=0, label_rule=0, link_gate=True)


# --- the mining helpers of nets/model.py:161-201, callable on their own (the training loss above fuses them) -----------
def _as_device(g, t, dtype):
    import numpy as np
    import torch
    if hasattr(t, "data") and not isinstance(t, (torch.Tensor, np.ndarray)):
        t = t.data
    if not isinstance(t, torch.Tensor):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.to(device=g.device, dtype=dtype).contiguous()


def get_pos_and_neg_masks(labels, graph=None):
    """nets/model.py:199-201: (labels == 1, labels == 0) as bool device tensors of the labels' shape."""
    import torch
    from .. import ops
    from ..graph import F32
    g = graph or get_default_graph()
    lab = _as_device(g, labels, F32)
    pos = torch.empty(lab.shape, dtype=torch.uint8, device=g.device)
    neg = torch.empty(lab.shape, dtype=torch.uint8, device=g.device)
    ops.label_masks(lab, 0, pos, neg)
    return pos.view(torch.bool), neg.view(torch.bool)


def OHNM_single_image(scores, n_pos, neg_mask, graph=None):
    """nets/model.py:161-184: scores = P(negative class) per pixel of ONE image, n_pos = its number of positives,
    neg_mask = its negatives.  Returns the float 0/1 mask of the selected negatives: the min(3 n_pos, #neg) lowest-scoring
    ones, ties at the threshold included (`scores <= -threshold`); all zeros when n_pos == 0."""
    import torch
    from .. import ops
    from ..graph import F32
    g = graph or get_default_graph()
    sc = _as_device(g, scores, F32)
    neg = _as_device(g, neg_mask, torch.uint8)
    if sc.numel() != neg.numel():
        raise ValueError("scores and neg_mask must have the same number of elements")
    npos = torch.tensor([int(n_pos)], dtype=torch.int32).to(g.device)
    out = torch.empty(sc.shape, dtype=F32, device=g.device)
    ops.ohnm_select(sc, None, neg, npos, 1, sc.numel(), 3.0, out, None)
    return out


def OHNM_batch(batch_size, neg_conf, pos_mask, neg_mask, graph=None):
    """nets/model.py:186-197: neg_conf / pos_mask / neg_mask [batch, pixels]; per image OHNM_single_image with n_pos =
    sum(pos_mask[i]); returns float(pos_mask) + selected negatives, [batch_size, pixels] (the reference passes a literal 14)."""
    import torch
    from .. import ops
    from ..graph import F32
    g = graph or get_default_graph()
    sc = _as_device(g, neg_conf, F32)
    pos = _as_device(g, pos_mask, torch.uint8)
    neg = _as_device(g, neg_mask, torch.uint8)
    if sc.shape[0] < batch_size or pos.shape != sc.shape or neg.shape != sc.shape:
        raise ValueError("neg_conf, pos_mask, neg_mask must be [batch >= batch_size, pixels]")
    hw = sc[0].numel()
    out = torch.empty((batch_size,) + tuple(sc.shape[1:]), dtype=F32, device=g.device)
    ops.ohnm_select(sc, pos, neg, None, int(batch_size), hw, 3.0, None, out)
    return out
