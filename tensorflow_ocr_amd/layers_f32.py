"""Forward-only f32 INFERENCE precision of the slim-style layers (Graph(precision="f32"); test.py / test_pixellink*.py
--precision f32).

Same variables, scopes, padding rules and batch-norm formulas as layers.py, but activations are f32, the convolutions
run on the matrix cores in f32 (ocr_conv2d_f32_mfma: v_mfma_f32_32x32x2_f32, csrc/f32_infer.hip) and the element-wise
kernels are the f32 ones of the same file — all in the product library.  Purpose: outputs within the north star's 1e-3 of
the f32 reference (tests/test_gpu_f32_verify.py, test_gpu_f32_mfma.py); the 16-bit MFMA kernels are checked layer by layer
against that reference's 16-bit-storage mode.  OCR_F32_CONV=direct swaps in the plain direct convolution of
libocr_verify.so (the independent checker).  No backward."""
from . import ops
from .graph import Act, F32, constant, variance_scaling
from ._lib import CONV_BIAS, CONV_RELU


def _bn_vars(g, C):
    from .layers import _bn_vars as f
    return f(g, C)


def conv2d(g, x, cout, k, scope, *, stride=1, rate=1, normalizer="bn", relu=True, pool=0, keep_full=True,
           is_training=True, bn_training=None, first=False, weight_decay=True, initializer=None):
    from .layers import BN_DECAY, BN_EPS
    n, h, w, cin = x.shape
    bn_training = is_training if bn_training is None else bn_training
    with g.variable_scope(scope):
        init = initializer or variance_scaling(g.rng)
        wv = g.get_variable("weights", (k, k, cin, cout), init, regularized=weight_decay)
        if normalizer == "bn":
            gamma, beta, mm, mv = _bn_vars(g, cout)
            bias = None
        else:
            bias = g.get_variable("biases", (cout,), constant(0.0))
    d = ops.conv_desc((n, h, w, cin), cout, k, k, stride, rate)
    oh, ow = d.oh, d.ow
    y = g.empty((n, oh, ow, cout), F32)
    if normalizer != "bn":
        d.flags = CONV_BIAS | (CONV_RELU if relu else 0)
        ops.conv2d_f32(d, x.data, wv.data, y, bias.data)
        a_full = Act(y, requires_grad=False, name=scope)
        a_pool = max_pool2d(g, a_full, 2, 2, scope=scope + "/pool") if pool else None
        return a_full, a_pool
    d.flags = 0
    ops.conv2d_f32(d, x.data, wv.data, y)
    scale, shift = g.empty((cout,), F32), g.empty((cout,), F32)
    if bn_training:
        mean, invstd = g.empty((cout,), F32), g.empty((cout,), F32)
        T = ops.channel_stats_f32_num_partials(n * oh * ow, cout)
        g.workspace()
        part, stage = g.ws_small.two(T * 2 * cout * 4, ops.bn_reduce_workspace(T, cout))
        ops.channel_stats_f32(y, cout, part)
        ops.bn_finalize(part, T, cout, float(n) * oh * ow, gamma.data, beta.data, BN_EPS, BN_DECAY, mm.data,
                        mv.data, scale, shift, mean, invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, scale, shift)
    full = pooled = None
    if pool:
        pooled = g.empty((n, (oh + 1) // 2, (ow + 1) // 2, cout), F32)
        if keep_full:
            full = g.empty((n, oh, ow, cout), F32)
        ops.bn_relu_f32(y, scale, shift, relu, 2, full, pooled)
    else:
        full = g.empty((n, oh, ow, cout), F32)
        ops.bn_relu_f32(y, scale, shift, relu, 0, full, None)
    a_full = Act(full, requires_grad=False, name=scope) if full is not None else None
    a_pool = Act(pooled, requires_grad=False, name=scope + "/pool") if pooled is not None else None
    return a_full, a_pool


def max_pool2d(g, x, k, stride, scope="pool"):
    n, h, w, c = x.shape
    oh, pt = ops.same_pad(h, k, stride)
    ow, pl = ops.same_pad(w, k, stride)
    y = g.empty((n, oh, ow, c), F32)
    ops.maxpool_f32(x.data, k, stride, (pt, pl), y)
    return Act(y, requires_grad=False, name=scope)


def prep_images(g, images, means):
    n, h, w, _ = images.shape
    out = g.empty((n, h, w, 3), F32)
    ops.prep_images_f32(images, out, means)
    return Act(out, requires_grad=False, name="images")


def head_conv(g, feat, wv, C, z, bias=None):
    """1x1 head conv on an f32 feature map; wv is the merged [cin, C] parameter (= HWIO [1,1,cin,C])."""
    n, h, w, cin = feat.shape
    d = ops.conv_desc((n, h, w, cin), C, 1, 1, 1, 1)
    d.flags = CONV_BIAS if bias is not None else 0
    ops.conv2d_f32(d, feat.data, wv.data, z, bias.data if bias is not None else None)


# ------------------------------------------------------------------ ResNet-v1 / EAST merge branch
class ConvBN:
    """Raw conv output + the batch-norm affine (what resnet_layers.conv_bn_raw returns)."""
    __slots__ = ("y", "scale", "shift")


def conv_bn_raw(g, x, cout, k, scope, *, stride=1, rate=1, is_training=True, weight_decay=True):
    """conv2d_same (stride-1 SAME conv, then subsample: resnet_utils.py:74-123) + BN statistics."""
    from .layers import BN_DECAY, BN_EPS
    n, h, w, cin = x.shape
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (k, k, cin, cout), variance_scaling(g.rng), regularized=weight_decay)
        gamma, beta, mm, mv = _bn_vars(g, cout)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, 1, rate)
    d.flags = 0
    y = g.empty((n, d.oh, d.ow, cout), F32)
    ops.conv2d_f32(d, x.data, wv.data, y)
    if stride > 1:
        oh, ow = (d.oh + stride - 1) // stride, (d.ow + stride - 1) // stride
        ys = g.empty((n, oh, ow, cout), F32)
        ops.maxpool_f32(y, 1, stride, (0, 0), ys)
        y = ys
    c = ConvBN()
    c.y = y
    c.scale, c.shift = _bn_affine(g, y, gamma, beta, mm, mv, is_training, BN_EPS, BN_DECAY)
    return c


def _bn_affine(g, y, gamma, beta, mm, mv, is_training, eps, decay):
    n, oh, ow, cout = y.shape
    scale, shift = g.empty((cout,), F32), g.empty((cout,), F32)
    if is_training:
        mean, invstd = g.empty((cout,), F32), g.empty((cout,), F32)
        T = ops.channel_stats_f32_num_partials(n * oh * ow, cout)
        g.workspace()
        part, stage = g.ws_small.two(T * 2 * cout * 4, ops.bn_reduce_workspace(T, cout))
        ops.channel_stats_f32(y, cout, part)
        ops.bn_finalize(part, T, cout, float(n) * oh * ow, gamma.data, beta.data, eps, decay, mm.data, mv.data,
                        scale, shift, mean, invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, eps, scale, shift)
    return scale, shift


def conv_bn_act(g, x, cout, k, scope, *, stride=1, rate=1, relu=True, is_training=True):
    c = conv_bn_raw(g, x, cout, k, scope, stride=stride, rate=rate, is_training=is_training)
    a = Act(g.empty(c.y.shape, F32), requires_grad=False, name=scope)
    ops.bn_relu_f32(c.y, c.scale, c.shift, relu, 0, a.data, None)
    return a


def bn_add_relu(g, c3, shortcut, scope):
    out = Act(g.empty(c3.y.shape, F32), requires_grad=False, name=scope)
    ops.bn_add_relu_f32(c3.y, c3.scale, c3.shift, shortcut.data, out.data)
    return out


def unpool(g, x):
    n, h, w, c = x.shape
    out = Act(g.empty((n, 2 * h, 2 * w, c), F32), requires_grad=False, name="unpool")
    ops.unpool_f32(x.data, out.data)
    return out


def concat_conv_bn_relu(g, xa, xb, cout, scope, is_training=True):
    from .layers import BN_DECAY, BN_EPS
    from ._lib import CONV_ACCUM_F16
    n, h, w, ca = xa.shape
    cb = xb.shape[-1]
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (1, 1, ca + cb, cout), variance_scaling(g.rng), regularized=True)
        gamma, beta, mm, mv = _bn_vars(g, cout)
    y = g.empty((n, h, w, cout), F32)
    da = ops.conv_desc((n, h, w, ca), cout, 1, 1)
    db = ops.conv_desc((n, h, w, cb), cout, 1, 1)
    da.flags = 0
    ops.conv2d_f32(da, xa.data, wv.data[0, 0, :ca], y)          # rows [0, ca) of the [ca+cb, cout] matrix
    db.flags = CONV_ACCUM_F16
    ops.conv2d_f32(db, xb.data, wv.data[0, 0, ca:], y)
    scale, shift = _bn_affine(g, y, gamma, beta, mm, mv, is_training, BN_EPS, BN_DECAY)
    a = Act(g.empty(y.shape, F32), requires_grad=False, name=scope)
    ops.bn_relu_f32(y, scale, shift, True, 0, a.data, None)
    return a
