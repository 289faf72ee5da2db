"""VGG-16 trunk: mirror of reference nets/vgg.py:6-42 (`basenet`).

conv1_1..conv5_3 (3x3, SAME), pool1-4 2x2/2, pool5 3x3/1, fc6 = 3x3 dilation 6 -> 1024,
fc7 = 1x1 -> 1024.  Variable names follow slim.repeat: `conv1/conv1_1/weights`, ...
The reference ignores `scope` (it never opens a variable_scope); so do we.
"""
from .. import layers
from ..graph import Act, get_default_graph


class LazyEndpoint:
    """An end point the forward pass did not materialise at full resolution (conv1_2/conv2_2
    are only consumed by their pooling layer); keeps the shape for `PixelLinkNet.get_shape`."""

    def __init__(self, shape):
        self.shape = tuple(shape)
        self.data = None


_CFG = [("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 3, 256), ("conv4", 3, 512), ("conv5", 3, 512)]


def basenet(inputs, scope='vgg16', *, graph=None, normalizer="bn", is_training=True,
            bn_training=True, initializer=None, keep_all_endpoints=False):
    """inputs: Act from layers.prep_images ([n,h,w,4] f16).  Returns (net, end_points) with keys
    conv1_2, conv2_2, conv3_3, conv4_3, conv5_3, fc6, fc7 (nets/vgg.py:11-40).

    normalizer="bn": every conv gets slim.batch_norm in TRAINING mode regardless of
    is_training, as under resnet_arg_scope in model_vgg;
    normalizer=None: bias + ReLU (PixelLinkNet)."""
    g = graph or get_default_graph()
    end_points = {}
    net = inputs
    kw = dict(normalizer=normalizer, is_training=is_training, bn_training=bn_training,
              initializer=initializer)
    for bi, (block, reps, cout) in enumerate(_CFG):
        with g.variable_scope(block):
            for r in range(1, reps + 1):
                last = r == reps
                first = bi == 0 and r == 1
                name = "%s_%d" % (block, r)
                if last and bi < 4:
                    keep = keep_all_endpoints or bi >= 2
                    full, pooled = layers.conv2d(g, net, cout, 3, name, pool=2, keep_full=keep,
                                                 first=first, **kw)
                    n, h, w, _ = net.shape
                    end_points[name] = full if full is not None else LazyEndpoint((n, h, w, cout))
                    net = pooled
                else:
                    full, _ = layers.conv2d(g, net, cout, 3, name, first=first, **kw)
                    net = full
                    if last:
                        end_points[name] = full
                    else:
                        # conv1_1, conv2_1, conv3_1/2, conv4_1/2, conv5_1/2: not an end point, read only by the next conv
                        full.sole_consumer = True
    net = layers.max_pool2d(g, net, 3, 1, scope="pool5")
    net, _ = layers.conv2d(g, net, 1024, 3, "fc6", rate=6, **kw)
    net.sole_consumer = True        # fc7 below; anything else that reads end_points['fc6'] is built later, contributes its
    end_points['fc6'] = net         # gradient earlier and switches the shortcut off (layers._conv_backward)
    net, _ = layers.conv2d(g, net, 1024, 1, "fc7", **kw)
    end_points['fc7'] = net
    return net, end_points
