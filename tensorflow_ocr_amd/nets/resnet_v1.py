"""Mirror of reference nets/resnet_v1.py + nets/resnet_utils.py: `resnet_v1_50` (:237-259) — the
slim variant with the stride in the LAST unit of each block (:248-254), classifier removed, end
points pool2 (/4, 64), pool3 = block1 (/8, 256), pool4 = block2 (/16, 512), pool5 = net (/32, 2048)
(:196,210-216).  Only depth 50 is real in the reference."""
from .. import layers, resnet_layers
from ..graph import get_default_graph

BLOCKS_50 = [
    ('block1', [(256, 64, 1)] * 2 + [(256, 64, 2)]),
    ('block2', [(512, 128, 1)] * 3 + [(512, 128, 2)]),
    ('block3', [(1024, 256, 1)] * 5 + [(1024, 256, 2)]),
    ('block4', [(2048, 512, 1)] * 3),
]


def resnet_v1(inputs, blocks, num_classes=None, is_training=True, global_pool=True, output_stride=None,
              include_root_block=True, spatial_squeeze=True, reuse=None, scope=None, graph=None):
    """nets/resnet_v1.py:114-231.  inputs: prepared image Act ([n,h,w,4] f16)."""
    g = graph or get_default_graph()
    if output_stride is not None:
        raise NotImplementedError("atrous output_stride is not used by the reference's callers")
    end_points = {}
    with g.variable_scope(scope):
        net = inputs
        if include_root_block:
            net = resnet_layers.root_block(g, net, "conv1", 64, is_training)
            net = layers.max_pool2d(g, net, 3, 2, scope="pool1")
            end_points['pool2'] = net
        for bname, units in blocks:
            with g.variable_scope(bname):
                for i, (depth, depth_bottleneck, stride) in enumerate(units):
                    net = resnet_layers.bottleneck(g, net, depth, depth_bottleneck, stride,
                                                   'unit_%d' % (i + 1), is_training)
            end_points['%s/%s' % (scope, bname)] = net
    end_points['pool3'] = end_points['%s/block1' % scope]
    end_points['pool4'] = end_points['%s/block2' % scope]
    end_points['pool5'] = net
    return net, end_points


def resnet_v1_50(inputs, num_classes=None, is_training=True, global_pool=True, output_stride=None,
                 spatial_squeeze=True, reuse=None, scope='resnet_v1_50', graph=None):
    """nets/resnet_v1.py:237-259."""
    return resnet_v1(inputs, BLOCKS_50, num_classes, is_training, global_pool, output_stride, True,
                     spatial_squeeze, reuse, scope, graph)


def _broken(depth):
    def f(*a, **k):
        raise KeyError("resnet_v1_%d: the reference hard-codes end_points['resnet_v1_50/block1'] "
                       "(nets/resnet_v1.py:210-215), so only depth 50 builds (SURVEY D7)" % depth)
    return f


resnet_v1_101, resnet_v1_152, resnet_v1_200 = _broken(101), _broken(152), _broken(200)
