"""Mirror of reference nets/model_vgg_16.py: `model_vgg` (VGG-16 trunk + BN'd PixelLink fuse
heads, :138-177), `dice_coefficient` (:179-193), `loss` (:196-225) with the same call
signatures and output names (pixel_cls [N,H/4,W/4,2], link_cls [N,H/4,W/4,16]).
"""
import numpy as np
import torch

from .. import layers, ops
from ..graph import F32, get_default_graph
from . import vgg


def unpool(inputs, graph=None):
    """tf.image.resize_bilinear x2, legacy sampling (nets/model_vgg_16.py:15-16)."""
    g = graph or get_default_graph()
    n, h, w, c = inputs.data.shape
    return layers.fuse(g, (n, 2 * h, 2 * w, c), prev=inputs)


def mean_image_subtraction(images, means=(123.68, 116.78, 103.94), graph=None):
    """nets/model_vgg_16.py:19-32.  Returns the prepared f16 [n,h,w,4] image Act."""
    g = graph or get_default_graph()
    if len(means) != images.shape[-1]:
        raise ValueError('len(means) must match the number of channels')
    return layers.prep_images(g, _to_device(g, images))


def _to_device(g, arr, dtype=F32):
    if isinstance(arr, torch.Tensor):
        return arr.to(device=g.device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(g.device)


def model(images, weight_decay=1e-5, is_training=True, graph=None, blocks=None):
    """nets/model_vgg_16.py:85-136: ResNet-v1-50 + the EAST feature-merging branch
    h_i = conv3x3(conv1x1(concat(unpool(h_{i-1}), f_i))), widths 128/64/32, one more 3x3, then
    F_score = sigmoid(conv1x1 -> 1), geo_map = sigmoid(conv1x1 -> 8)."""
    from .. import resnet_layers as R
    from . import resnet_v1
    g = graph or get_default_graph()
    g.weight_decay = weight_decay
    x4 = mean_image_subtraction(images, graph=g)
    if blocks is None:
        _, end_points = resnet_v1.resnet_v1_50(x4, is_training=is_training, scope='resnet_v1_50', graph=g)
    else:
        _, end_points = resnet_v1.resnet_v1(x4, blocks, is_training=is_training, scope='resnet_v1_50', graph=g)
    g.end_points = end_points
    f = [end_points['pool5'], end_points['pool4'], end_points['pool3'], end_points['pool2']]
    num_outputs = [None, 128, 64, 32]
    names = iter(['Conv'] + ['Conv_%d' % i for i in range(1, 9)])
    with g.variable_scope('feature_fusion'):
        h = f[0]
        for i in range(1, 4):
            # conv1x1(concat(unpool(h), f_i)): the upsampled branch's share of the convolution runs before the resize
            c1_1 = R.unpool_concat_conv_bn_relu(g, h, f[i], num_outputs[i], next(names), is_training)
            h = R.conv_bn_act(g, c1_1, num_outputs[i], 3, next(names), is_training=is_training)
        gi = R.conv_bn_act(g, h, num_outputs[3], 3, next(names), is_training=is_training)
        F_score, geo_map = R.sigmoid_heads(g, gi, (1, 8), (next(names), next(names)))
    return F_score, geo_map


def model_vgg(images, weight_decay=1e-5, is_training=True, graph=None):
    """nets/model_vgg_16.py:138-177.  images: [N,H,W,3] float (0..255).  Returns (pixel_cls, link_cls)."""
    g = graph or get_default_graph()
    g.weight_decay = weight_decay
    x4 = mean_image_subtraction(images, graph=g)
    _, end_points = vgg.basenet(x4, scope='vgg_16', graph=g, normalizer="bn", is_training=is_training)
    g.end_points = end_points
    with g.variable_scope('feature_fusion'):
        # slim auto-names: pixel convs Conv..Conv_4, link convs Conv_5..Conv_9
        srcs = [('fc7', ('Conv', 'Conv_5')), ('conv5_3', ('Conv_1', 'Conv_6')),
                ('conv4_3', ('Conv_2', 'Conv_7')), ('conv3_3', ('Conv_3', 'Conv_8'))]
        trip = layers.head_group(g, [end_points[key] for key, _ in srcs], [names for _, names in srcs], (2, 16),
                                 mode="bn", is_training=is_training)
        heads = {key: t for (key, _), t in zip(srcs, trip)}
        n, h, w, _ = end_points['fc7'].shape
        s1 = layers.fuse(g, (n, h, w, 18), a=heads['fc7'], b=heads['conv5_3'])
        s2 = layers.fuse(g, (n, 2 * h, 2 * w, 18), a=heads['conv4_3'], prev=s1)
        s3 = layers.fuse(g, (n, 4 * h, 4 * w, 18), a=heads['conv3_3'], prev=s2)
        pixel_cls, link_cls = layers.pointwise_pair(g, s3, ('Conv_4', 'Conv_9'), mode="bn", is_training=is_training)
    return pixel_cls, link_cls


class Scalar:
    """Device scalar (the loss) with its components; `.item()` syncs."""

    def __init__(self, buf):
        self.data = buf

    def item(self):
        return float(self.data[0].item())

    def terms(self):
        return self.data[1:10].detach().cpu().numpy()


def dice_coefficient(y_true_cls, y_pred_cls, training_mask, graph=None):
    """nets/model_vgg_16.py:179-193 for one map (host convenience: builds the 9-map call with
    zero link maps and returns the pixel term)."""
    g = graph or get_default_graph()
    yt = _to_device(g, y_true_cls)
    yp = y_pred_cls.data if hasattr(y_pred_cls, "data") and not isinstance(y_pred_cls, torch.Tensor) else _to_device(g, y_pred_cls)
    m = _to_device(g, training_mask)
    P = m.numel()
    ztl = g.zeros((P, 8))
    sums, out = g.zeros((27,)), g.zeros((10,))
    ops.dice_loss_fwd(yt, yp, ztl, ztl, m, sums, out, g.workspace())
    return Scalar(out[1:2].clone())


def loss(y_true_pixel, y_pred_pixel, y_true_link, y_pred_link, training_mask, graph=None):
    """nets/model_vgg_16.py:196-225: 2*dice(pixel) + sum_8 dice(link_i), TF broadcasting of the
    1-channel labels against 2-/16-channel predictions included.  Records the backward seed."""
    g = graph or get_default_graph()
    ytp, ytl, m = _to_device(g, y_true_pixel), _to_device(g, y_true_link), _to_device(g, training_mask)
    sums, out = g.empty((27,), F32), g.empty((10,), F32)      # fully written by the finalize kernel
    ops.dice_loss_fwd(ytp, y_pred_pixel.data, ytl, y_pred_link.data, m, sums, out, g.workspace())

    def backward():
        y_pred_pixel.grad = g.empty(y_pred_pixel.data.shape, F32)
        y_pred_link.grad = g.empty(y_pred_link.data.shape, F32)
        ops.dice_loss_bwd(ytp, ytl, m, sums, g.seed_scale(), y_pred_pixel.grad, y_pred_link.grad)
    g.record(backward)
    res = Scalar(out)
    g.collections["losses"].append(res)
    return res


def ohem_loss(y_true_pixel, y_pred_pixel, y_true_link, y_pred_link, training_mask, graph=None):
    """nets/model_vgg_16.py:243-282 (with cal_link_loss :227-241): despite the name no mining —
    pixel CE over positives / n_pos, link CEs weighted by the positive-pixel mask, L_pixel*2 + L_link."""
    from .. import losses
    g = graph or get_default_graph()
    return losses.softmax_loss(g, y_pred_pixel, y_pred_link, y_true_pixel, y_true_link,
                               pixel_rule=1, label_rule=0, link_gate=True)


def cal_link_loss(link_gt, link_pred, W_pixel, graph=None):
    """nets/model_vgg_16.py:227-241 for one direction: link_gt [...,1] (or [...]) 0/1 labels, link_pred [...,2] logits —
    tensors, tf.split-style channel slices of the 8- / 16-channel maps (strided views are read in place), or a head handle
    (`.data`, `.grad`) — W_pixel [pixels] weights.  sum(CE Wpos)/sum(Wpos) + sum(CE Wneg)/sum(Wneg), unguarded like the
    reference.  Returns a device Scalar; with a head handle the backward seed is recorded (its `.grad` = d loss / d logits)."""
    g = graph or get_default_graph()
    handle = link_pred if (hasattr(link_pred, "grad") and hasattr(link_pred, "data") and not isinstance(link_pred, torch.Tensor)) else None
    pred = handle.data if handle is not None else link_pred
    gt = link_gt.data if (hasattr(link_gt, "data") and not isinstance(link_gt, (torch.Tensor, np.ndarray))) else link_gt

    def dev(t):
        if not isinstance(t, torch.Tensor):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        return t.to(device=g.device, dtype=F32)          # a view stays a view

    def rows(t, width):
        """(tensor, row stride in elements) of a [..., width] view whose rows are uniformly strided."""
        if t.dim() == 0 or t.shape[-1] != width:
            if width == 1:
                t = t.unsqueeze(-1)
            else:
                raise ValueError("link_pred must end in 2 logits")
        if t.stride(-1) != 1 and width > 1:
            raise ValueError("the logit pair must be contiguous")
        lead = t.dim() - 1
        step = t.stride(lead - 1) if lead >= 1 else width
        for i in range(lead - 1):
            if t.shape[i + 1] != 1 and t.stride(i) != t.stride(i + 1) * t.shape[i + 1]:
                raise ValueError("rows are not uniformly strided: pass a contiguous tensor or a channel slice of one")
        return t, int(step)
    gt_t, gs = rows(dev(gt), 1)
    pr_t, ps = rows(dev(pred), 2)
    W = dev(W_pixel).contiguous().reshape(-1)
    P = pr_t.numel() // 2
    if gt_t.numel() != P or W.numel() != P:
        raise ValueError("link_gt, link_pred and W_pixel must cover the same pixels")
    sums, out = g.empty((4,), F32), g.empty((1,), F32)
    ops.link_ce_fwd(gt_t, gs, pr_t, ps, W, P, sums, out, g.workspace())
    if handle is not None:
        def backward():
            handle.grad = g.empty(pr_t.shape, F32)
            ops.link_ce_bwd(gt_t, gs, pr_t, ps, W, P, sums, g.seed_scale(), handle.grad, 2)
        g.record(backward)
    return Scalar(out)
