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
