"""Mirror of reference nets/pixellink.py: `PixelLinkNet` (VGG-16 with biases, no BN; fuse heads
`stage_{6,5,4,3}_{pixel,link}_fuse`, `text_predication`, `link_predication`; :8-86) and
`build_loss` (:88-263), same constructor / attribute / method names.

Variable scopes follow the reference: `vgg/conv1/conv1_1/{weights,biases}` ...,
`pixellink_layers/stage_6_pixel_fuse/...` (the pixel and link fuse convs that read the same feature
map are one merged parameter internally; `checkpoint.internal_to_tf` splits it)."""
import torch

from .. import layers, losses, ops
from ..graph import F32, constant, get_default_graph, xavier_uniform
from . import vgg


class config:
    """Stand-in for the reference's missing global `config` module: the few fields the
    net reads."""
    batch_size_per_gpu = None        # taken from the tensor
    max_neg_pos_ratio = 3
    train_with_ignored = True
    data_format = 'NHWC'


class _LossTerm:
    """One entry of tf.GraphKeys.LOSSES (train_pixellink.py:260-263 asserts there are exactly 2)."""

    def __init__(self, scalar, fn):
        self._s, self._fn = scalar, fn

    def item(self):
        return float(self._fn(self._s.data.detach().cpu().numpy()))


class PixelLinkNet(object):
    def __init__(self, inputs, weight_decay=None, basenet_type='vgg', data_format='NHWC',
                 weights_initializer=None, biases_initializer=None, graph=None, input_norm=None):
        """input_norm=(mean, std): `inputs` are RAW images and the input pipeline's normalisation
        `(x - mean) / std` (what the reference's queue applies before the net, train_pixellink.py:150-154)
        runs inside the image-preparation kernel; None: `inputs` are already preprocessed."""
        if data_format != 'NHWC':
            raise ValueError("only NHWC is implemented (the reference default)")
        self.g = graph or get_default_graph()
        self.inputs = inputs
        self.input_norm = input_norm
        self.weight_decay = weight_decay
        self.basenet_type = basenet_type
        self.data_format = data_format
        self.weights_initializer = weights_initializer or xavier_uniform(self.g.rng)
        self.biases_initializer = biases_initializer or constant(0.0)
        self._build_network()
        self.shapes = self.get_shapes()

    def get_shapes(self):
        return {k: tuple(v.shape[1:-1]) for k, v in self.end_points.items()}

    def get_shape(self, name):
        return self.shapes[name]

    def unpool(self, inputs):
        n, h, w, c = inputs.data.shape
        return layers.fuse(self.g, (n, 2 * h, 2 * w, c), prev=inputs)

    def _build_network(self):
        g = self.g
        x = self.inputs
        if isinstance(x, torch.Tensor) or not hasattr(x, "requires_grad"):
            x = losses.to_device(g, x)
            if self.input_norm is None:
                x = layers.prep_images(g, x, means=(0.0, 0.0, 0.0))   # PixelLinkNet takes preprocessed input
            else:
                m, sd = self.input_norm
                x = layers.prep_images(g, x, means=(m, m, m), div=sd)
        with g.variable_scope(self.basenet_type):
            basenet, end_points = vgg.basenet(x, graph=g, normalizer=None,
                                              initializer=self.weights_initializer)
            self.basenet = basenet
            self.end_points = end_points
        with g.variable_scope('pixellink_layers'):
            self._add_pixellink_layers(basenet, end_points)

    def _add_pixellink_layers(self, basenet, end_points):
        g = self.g
        srcs = [('fc7', 'stage_6'), ('conv5_3', 'stage_5'), ('conv4_3', 'stage_4'), ('conv3_3', 'stage_3')]
        trip = layers.head_group(g, [end_points[key] for key, _ in srcs],
                                 [(st + '_pixel_fuse', st + '_link_fuse') for _, st in srcs], (2, 16), mode="bias",
                                 initializer=xavier_uniform)
        heads = {key: t for (key, _), t in zip(srcs, trip)}
        n, h, w, _ = end_points['fc7'].shape
        s1 = layers.fuse(g, (n, h, w, 18), a=heads['fc7'], b=heads['conv5_3'])
        s2 = layers.fuse(g, (n, 2 * h, 2 * w, 18), a=heads['conv4_3'], prev=s1)
        s3 = layers.fuse(g, (n, 4 * h, 4 * w, 18), a=heads['conv3_3'], prev=s2)
        self.pixel_cls, self.link_cls = layers.pointwise_pair(g, s3, ('text_predication', 'link_predication'), mode="bias")
        return self.pixel_cls, self.link_cls

    @property
    def pixel_scores(self):
        """slim.softmax(pixel_cls) (nets/pixellink.py:71)."""
        out = self.g.empty(self.pixel_cls.data.shape, F32)
        ops.softmax_pairs(self.pixel_cls.data, out)
        return out

    def build_loss(self, pixel_labels, link_labels, do_summary=True, focal=None):
        """nets/pixellink.py:88-263: adds (2 * mean pixel CE) and (sum of the 8 link losses) to the
        LOSSES collection; `focal=(alpha, gamma)` swaps the link CE for the focal loss."""
        g = self.g
        s = losses.softmax_loss(g, self.pixel_cls, self.link_cls, pixel_labels, link_labels,
                                pixel_rule=2, label_rule=1, link_gate=False, focal=focal)
        g.collections["losses"].pop()        # replaced by the two reference entries
        g.collections["losses"].append(_LossTerm(s, lambda v: 2.0 * v[1]))
        g.collections["losses"].append(_LossTerm(s, lambda v: v[2:10].sum()))
        self.loss = s
        return s


def tensor_shape(t):
    return list(t.shape)
