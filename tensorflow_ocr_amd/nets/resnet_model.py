"""`model.model` of reference nets/model.py:84-143: ResNet-v1-50 + BN'd 1x1 fuse heads on
pool5..pool2 (top-down unpool + add), final 1x1 convs with biases and no activation ->
pixel_4 [N,H/4,W/4,2], link_4 [N,H/4,W/4,16] logits."""
from .. import layers
from ..graph import get_default_graph
from . import model_vgg_16 as _mv
from . import resnet_v1


def model_resnet50_pixellink(images, weight_decay=1e-5, is_training=True, graph=None, blocks=None):
    g = graph or get_default_graph()
    g.weight_decay = weight_decay
    x4 = _mv.mean_image_subtraction(images, graph=g)
    if blocks is None:
        _, end_points = resnet_v1.resnet_v1_50(x4, is_training=is_training, scope='resnet_v1_50', graph=g)
    else:       # reduced block lists (tests)
        _, end_points = resnet_v1.resnet_v1(x4, blocks, is_training=is_training, scope='resnet_v1_50', graph=g)
    g.end_points = end_points
    f = [end_points['pool5'], end_points['pool4'], end_points['pool3'], end_points['pool2']]
    with g.variable_scope('feature_fusion'):
        names = [('Conv', 'Conv_4'), ('Conv_1', 'Conv_5'), ('Conv_2', 'Conv_6'), ('Conv_3', 'Conv_7')]
        heads = layers.head_group(g, f, names, (2, 16), mode="bn", is_training=is_training)
        n, h, w, _ = f[0].shape
        s0 = layers.fuse(g, (n, h, w, 18), a=heads[0])                          # relu(bn(conv(pool5)))
        s1 = layers.fuse(g, (n, 2 * h, 2 * w, 18), a=heads[1], prev=s0)       # unpool(.) + conv(pool4)
        s2 = layers.fuse(g, (n, 4 * h, 4 * w, 18), a=heads[2], prev=s1)
        s3 = layers.fuse(g, (n, 8 * h, 8 * w, 18), a=heads[3], prev=s2)
        pixel_4, link_4 = layers.pointwise_pair(g, s3, ('Conv_8', 'Conv_9'), mode="bias")
    return pixel_4, link_4
