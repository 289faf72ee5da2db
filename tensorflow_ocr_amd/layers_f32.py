"""Forward-only f32 VERIFICATION precision of the slim-style layers (Graph(precision="f32")).

Same variables, scopes, padding rules and batch-norm formulas as layers.py, but activations are f32
and the kernels are the plain ones of csrc/verify_f32.hip.  Purpose: whole-graph outputs that can be
compared with the f32 CPU reference at the north star's 1e-3 (tests/test_gpu_f32_verify.py); the f16 MFMA
kernels themselves are checked layer by layer against that reference's f16-storage mode.  No backward."""
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
