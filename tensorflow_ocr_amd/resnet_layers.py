"""ResNet-v1 building blocks on the HIP kernels: conv + slim.batch_norm (+ReLU), the 7x7/2 root
convolution, and the bottleneck tail relu(shortcut + bn(conv3)) — reference nets/resnet_v1.py:68-111,
181-196 and nets/resnet_utils.py:59-122.

Strided 3x3 convolutions (three in ResNet-50) reuse the stride-1 MFMA kernels.  For stride 2 on even
sizes the input is shuffled space-to-depth and the layer becomes ONE stride-1 2x2 convolution over
4*cin channels (csrc/s2d.hip: 16/9 of the minimal work, output-resolution tiles, fused BN partial
sums).  Other strides fall back on the identity the reference's own docstring states
(resnet_utils.py:85-93): conv2d_same(x, n, 3, stride=s) == subsample(conv2d(x, n, 3, stride=1, SAME), s)
— 4x the work at stride 2 plus a full-resolution intermediate, which is why it is only the fallback.
"""
import torch

from . import ops
from .graph import Act, F32, constant, variance_scaling
from .layers import BN_DECAY, BN_EPS, FUSE_BN_REDUCE, _bn_vars, _packs
from ._lib import CONV_ACCUM_F16, CONV_STATS


USE_S2D = __import__("os").environ.get("OCR_RESNET_S2D", "1") == "1"     # measurement switch (A/B against the subsample form)
FUSE_TAIL = __import__("os").environ.get("OCR_RESNET_FUSE_TAIL", "1") == "1"   # measurement switch (bottleneck tail fusion)
FUSE_SUB = __import__("os").environ.get("OCR_RESNET_FUSE_SUB", "1") == "1"     # ... subsampled shortcut gradient in that conv's epilogue
# round 3: the element-wise passes around the 1x1 convolutions applied while those convolutions load their operand
FUSE_FWD = __import__("os").environ.get("OCR_RESNET_FUSE_FWD", "1") == "1"     # relu(bn(conv3) + shortcut) inside the NEXT 1x1 conv
FUSE_BWD = __import__("os").environ.get("OCR_RESNET_FUSE_BWD", "1") == "1"     # conv3's BN-backward apply inside its input-gradient conv
# round 4: the same on-load apply for the BN (+ReLU) above conv1 and above a projection shortcut, in front of the tail epilogue
# relu(bn(conv2)) inside conv3 (ocr_conv2d_pw_bnrelu_f16): measured NEUTRAL at 64 x 640^2 (stage 2 -7 us per unit, stages 3 / 4
# +6 / +12: the register-staged loader gives back in the 256 -> 1024 GEMM what the removed pass saved), off; path kept under test
FUSE_FWD_ACT = __import__("os").environ.get("OCR_RESNET_FUSE_FWD_ACT", "0") == "1"
FUSE_BWD_WIDE = __import__("os").environ.get("OCR_RESNET_FUSE_BWD_WIDE", "1") == "1"
FUSE_ROOT_POOL = __import__("os").environ.get("OCR_RESNET_FUSE_ROOT_POOL", "1") == "1"   # root conv: BN + ReLU inside the max-pool that follows
FUSE_ROOT_WGRAD = __import__("os").environ.get("OCR_RESNET_FUSE_ROOT_WGRAD", "1") == "1"   # root conv: BN-backward apply inside its weight gradient
FUSE_ROOT_GATHER = __import__("os").environ.get("OCR_RESNET_FUSE_ROOT_GATHER", "1") == "1"   # ... and the pool's backward inside that BN's reduction pass
MASK_BITS = __import__("os").environ.get("OCR_RESNET_MASK_BITS", "0") == "1"   # tail mask as bits instead of the output tensor: measured SLOWER (byte stores +8 % on the writers, byte loads no faster in the latency-bound tail epilogue), off


class ConvBN:
    """Raw conv output + the batch-norm coefficients of one conv layer."""
    __slots__ = ("y", "scale", "shift", "mean", "invstd", "gamma", "beta", "wv", "backward_from", "can_fuse_bwd")


def _pw_fusable(d):
    """The pointwise GEMM kernel (and so conv_pwx_kernel) takes this 1x1 convolution."""
    return d.kh == 1 and d.cin >= 128 and ops.conv2d_variant(d).startswith("conv_pw_kernel")


def conv_bn_raw(g, x, cout, k, scope, *, stride=1, rate=1, is_training=True, weight_decay=True, owner=None):
    """slim.conv2d(normalizer_fn=slim.batch_norm) up to (not including) the normalise step.
    x: Act f16 [n,h,w,cin].  Returns ConvBN; `backward_from(dy)` propagates the gradient of the
    conv output into the weights and into x."""
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.conv_bn_raw(g, x, cout, k, scope, stride=stride, rate=rate, is_training=is_training,
                                      weight_decay=weight_decay)
    n, h, w, cin = x.shape
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (k, k, cin, cout), variance_scaling(g.rng), regularized=weight_decay)
        gamma, beta, mm, mv = _bn_vars(g, cout)
    ws = g.workspace()
    strided = stride != 1
    if strided and k == 1:
        raise NotImplementedError("strided 1x1 conv (the slim variant in the reference has none)")
    if owner is None or owner is not x.pending_owner:
        x.consumed = True
    s2d = strided and stride == 2 and k == 3 and rate == 1 and h % 2 == 0 and w % 2 == 0 and cin % 16 == 0 and USE_S2D
    if s2d:
        def mk22(old):
            t = old if old is not None else g.empty((2, 2, 4 * cin, cout), F32)
            ops.weights_s2d(wv.data, t)
            return t
        w22 = g.packed(wv, "s2d_f32", mk22)

        def mkp(old):
            if old is None:
                old = (g.empty((4, cout, 4 * cin)), g.empty((4, 4 * cin, cout)))
            ops.pack_weights(w22, old[0], old[1])
            return old
        w_fwd, w_dg = g.packed(wv, "s2d_kc_ck", mkp)
        oh, ow = h // 2, w // 2
        xs = g.empty((n, oh, ow, 4 * cin))
        ops.space_to_depth(x.data, xs)
        d = ops.conv_desc((n, oh, ow, 4 * cin), cout, 2, 2, 1, 1, pad=(1, 1), out_hw=(oh, ow))
        y_full = None
        y = g.empty((n, oh, ow, cout))
        T = ops.conv2d_num_mtiles(d)
        part, stage = g.ws_small.two(T * 2 * cout * 4, ops.bn_reduce_workspace(T, cout))
        d.flags = CONV_STATS if is_training else 0
        ops.conv2d(d, xs, w_fwd, y, None, part if is_training else None)
        x_in = xs
    else:
        w_fwd, w_dg = _packs(g, wv, False)
        d = ops.conv_desc((n, h, w, cin), cout, k, k, 1, rate)          # stride-1 SAME
        x_in = x._data            # (the buffer: a deferred x is filled by the convolution below or by x.data)
        y_full = g.empty((n, d.oh, d.ow, cout))
        mt = ops.conv2d_num_mtiles(d)
    if s2d:
        pass
    elif strided:
        d.flags = 0
        ops.conv2d(d, x.data, w_fwd, y_full, None, None)
        oh, ow = -(-h // stride), -(-w // stride)
        y = g.empty((n, oh, ow, cout))
        ops.maxpool(y_full, 1, stride, (0, 0), y)                   # subsample (resnet_utils.py:74)
        T = ops.channel_stats_num_partials(n * oh * ow, cout)
        part, stage = g.ws_small.two(T * 2 * cout * 4, ops.bn_reduce_workspace(T, cout))
        if is_training:
            ops.channel_stats(y, part)
    else:
        oh, ow = d.oh, d.ow
        y = y_full
        T = mt
        part, stage = g.ws_small.two(T * 2 * cout * 4, ops.bn_reduce_workspace(T, cout))
        d.flags = CONV_STATS if is_training else 0
        if (x.deferred is not None and x.tail_fwd is not None and is_training and FUSE_FWD and cout <= 512
                and _pw_fusable(d)):
            # x is the previous unit's output and nobody has computed it yet: this convolution does, while loading
            # its pixel operand, and writes it (and its ReLU-mask bits) for the consumers that follow
            x.deferred = None
            py, psc, psh, short, ssc, ssh, bits = x.tail_fwd
            ops.conv2d_pw_bnaddrelu(d, py, psc, psh, short, ssc, ssh, x._data, bits, w_fwd, y, part)
        elif (x.deferred is not None and x.bn_fwd is not None and x.bn_fwd[3] and is_training and FUSE_FWD_ACT
                and _pw_fusable(d)):
            # x = relu(bn(conv2)) that nobody has computed yet (conv_bn_act(defer=True)): applied to the operand rows
            x.deferred = None
            by, bsc, bsh, _ = x.bn_fwd
            ops.conv2d_pw_bnrelu(d, by, bsc, bsh, x._data, w_fwd, y, part)
        else:
            ops.conv2d(d, x.data, w_fwd, y, None, part if is_training else None)
    c = ConvBN()
    c.y, c.gamma, c.beta, c.wv = y, gamma, beta, wv
    c.scale, c.shift = g.empty((cout,), F32), g.empty((cout,), F32)
    c.mean, c.invstd = g.empty((cout,), F32), g.empty((cout,), F32)
    if is_training:
        ops.bn_finalize(part, T, cout, float(n) * oh * ow, gamma.data, beta.data, BN_EPS, BN_DECAY, mm.data,
                        mv.data, c.scale, c.shift, c.mean, c.invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, c.scale, c.shift)

    def backward_from(dy, fused=None):
        """dy: gradient of the conv output.  fused = (dz, y_bn, (A, B, C)): dy is still EMPTY — it is the batch-norm
        backward apply A*dz + B*y_bn + C, which the input-gradient convolution computes while loading its operand and
        writes into `dy` for the weight gradient (see `can_fuse_bwd`)."""
        if fused is not None:
            dz_f, ybn_f, coef_f, relu_shift = fused
            flags = 0
            if x.grad is None:
                x.grad = g.empty(x.shape)
            else:
                flags |= CONV_ACCUM_F16
                x.bn_partial = None
            last = False
            if x.pending is not None and owner is not None and owner is x.pending_owner:
                x.pending -= 1
                assert x.pending >= 0, "more gradient contributions than the owning unit has consumers"
                last = x.pending == 0
            dg = ops.ConvDesc(d.n, d.oh, d.ow, d.cout, d.h, d.w, d.cin, 1, 1, 1, 1, 0, 0, 1, flags)
            Tm = ops.conv2d_num_mtiles(dg)
            if last and x.tail_ctx is not None and FUSE_TAIL:
                # (as the unfused path below: the last contribution to a bottleneck output's gradient)
                partial = g.empty((Tm, 2, d.cin), F32)
                ops.conv2d_pw_bnbwd_tail(dg, dz_f, ybn_f, coef_f, relu_shift, dy, w_dg, x.grad, partial, x.tail_ctx, x.sub_grad)
                x.sub_grad = None
                x.tail_partial = (partial, Tm)
            elif x.bn_ctx is not None and not flags and FUSE_BN_REDUCE and relu_shift is None and x.pending is None:
                partial = g.empty((Tm, 2, d.cin), F32)
                ops.conv2d_pw_bnbwd_bnred(dg, dz_f, ybn_f, coef_f, dy, w_dg, x.grad, partial, x.bn_ctx)
                x.bn_partial = (partial, Tm)
            else:
                ops.conv2d_pw_bnbwd_tail(dg, dz_f, ybn_f, coef_f, relu_shift, dy, w_dg, x.grad)
            dd = ops.ConvDesc(d.n, d.h, d.w, d.cin, d.oh, d.ow, d.cout, 1, 1, 1, 1, 0, 0, 0, 0)
            ops.conv2d_wgrad(dd, x_in, dy, wv.grad, g.ws_wgrad)
            return
        if strided and not s2d:
            dy_full = g.empty(y_full.shape)
            ops.maxpool_bwd(y_full, dy, 1, stride, (0, 0), dy_full, False)   # zero insertion
            dy = dy_full
        dd = ops.ConvDesc(d.n, d.h, d.w, d.cin, d.oh, d.ow, d.cout, d.kh, d.kw, 1, d.dilation, d.pad_top,
                          d.pad_left, 0, 0)
        def weight_gradient():
            if s2d:
                dw22 = g.empty((2, 2, 4 * cin, cout), F32)
                ops.conv2d_wgrad(dd, x_in, dy, dw22, g.ws_wgrad)
                ops.weights_s2d_grad(dw22, wv.grad)
            else:
                ops.conv2d_wgrad(dd, x_in, dy, wv.grad, g.ws_wgrad)
        def input_gradient():
            pt = d.dilation * (d.kh - 1) - d.pad_top
            pl = d.dilation * (d.kw - 1) - d.pad_left
            if s2d:
                # input gradient in the shuffled layout, then back to full resolution (added to what is there)
                dxs = g.empty(x_in.shape)
                dg = ops.ConvDesc(d.n, d.oh, d.ow, d.cout, d.h, d.w, d.cin, d.kh, d.kw, 1, d.dilation, pt, pl, 1, 0)
                ops.conv2d(dg, dy, w_dg, dxs, None, None)
                had = x.grad is not None
                if not had:
                    x.grad = g.empty(x.shape)
                ops.depth_to_space(dxs, x.grad, had)
                x.bn_partial = None
                return
            flags = 0
            if x.grad is None:
                x.grad = g.empty(x.shape)
            else:
                flags |= CONV_ACCUM_F16
                x.bn_partial = None     # an earlier consumer's fused BN-backward sums no longer cover the full gradient
            dg = ops.ConvDesc(d.n, d.oh, d.ow, d.cout, d.h, d.w, d.cin, d.kh, d.kw, 1, d.dilation, pt, pl, 1, flags)
            last = False
            if x.pending is not None and owner is not None and owner is x.pending_owner:
                # only the owning unit's two contributions are counted (any other consumer of x was built
                # later, runs earlier in the backward pass and has already added its share)
                x.pending -= 1
                assert x.pending >= 0, "more gradient contributions than the owning unit has consumers"
                last = x.pending == 0
            if last and x.tail_ctx is not None and k == 1 and FUSE_TAIL:
                # x is the previous bottleneck's output and this is the last contribution to its gradient: store
                # the gradient past its ReLU and emit the BN-backward sums of its last conv (ops.conv2d_bnred_tail)
                Tm = ops.conv2d_num_mtiles(dg)
                partial = g.empty((Tm, 2, d.cin), F32)
                ops.conv2d_bnred_tail(dg, dy, w_dg, x.grad, partial, x.tail_ctx, x.sub_grad)
                x.sub_grad = None
                x.tail_partial = (partial, Tm)
            elif x.bn_ctx is not None and not flags and FUSE_BN_REDUCE:
                # sole consumer of a conv+BN(+ReLU) output: this input-gradient kernel also emits that
                # layer's BN-backward sums, so its backward skips the reduction pass (layers.py does the same)
                Tm = ops.conv2d_num_mtiles(dg)
                partial = g.empty((Tm, 2, d.cin), F32)
                ops.conv2d_bnred(dg, dy, w_dg, x.grad, partial, x.bn_ctx)
                x.bn_partial = (partial, Tm)
            else:
                ops.conv2d(dg, dy, w_dg, x.grad, None, None)
        weight_gradient()
        if x.requires_grad:
            input_gradient()

    def can_fuse_bwd(wide=False):
        """The fused form of `backward_from`: this is a stride-1 1x1 convolution the pointwise kernel takes.  Default: its
        input is a conv+BN(+ReLU) output nobody else has contributed a gradient to (the epilogue's fused reduction
        applies: conv3).  wide: any input — the epilogue is whatever the unfused input-gradient launch would use (plain,
        accumulate, bottleneck tail): conv1 and the projection shortcut."""
        if not (FUSE_BWD and k == 1 and not strided and x.requires_grad and cout >= 128):
            return False
        if wide:
            if not FUSE_BWD_WIDE:
                return False
        elif not (FUSE_BN_REDUCE and x.grad is None and x.bn_ctx is not None and x.pending is None):
            return False
        dg = ops.ConvDesc(d.n, d.oh, d.ow, d.cout, d.h, d.w, d.cin, 1, 1, 1, 1, 0, 0, 1, 0)
        return _pw_fusable(dg)
    c.backward_from = backward_from
    c.can_fuse_bwd = can_fuse_bwd
    return c


# (Guests beside held-back weight gradients — layers.conv2d / train.schedule_guests — were measured for THIS net too and
# are not used: 39.13-39.18 ms/step at 64 x 640^2 against 38.77 without; its apply passes are 47 us each, a fork + join
# costs the main queue ~25 us, and its hosts are the HBM-hungry pointwise weight gradients.)


def conv_bn_act(g, x, cout, k, scope, *, stride=1, rate=1, relu=True, is_training=True, owner=None, defer=False):
    """conv + batch_norm + (ReLU | identity) -> Act.  defer: the activation is left to its first reader (a 1x1 convolution
    that applies the batch norm + ReLU while loading — conv_bn_raw — or anyone's `.data`, which runs the plain pass)."""
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.conv_bn_act(g, x, cout, k, scope, stride=stride, rate=rate, relu=relu,
                                      is_training=is_training)
    c = conv_bn_raw(g, x, cout, k, scope, stride=stride, rate=rate, is_training=is_training, owner=owner)
    a = Act(g.empty(c.y.shape), name=scope)

    def fill():
        ops.bn_relu(c.y, c.scale, c.shift, relu, 0, a._data, None)
    if defer and is_training and FUSE_FWD_ACT:
        a.deferred = fill
        a.bn_fwd = (c.y, c.scale, c.shift, relu)
    else:
        fill()
    ws = g.workspace()
    if is_training:
        a.bn_ctx = (c.y, c.scale, c.shift, c.mean, c.invstd, relu)

    def backward():
        if a.grad is None:
            return
        dy = g.empty(c.y.shape)
        if a.bn_partial is not None and c.can_fuse_bwd(wide=True):
            # the sums are reduced already (the consumer's input-gradient epilogue): finalize them into the apply step's
            # coefficients and let this layer's own input-gradient convolution apply them (and the ReLU mask) on load
            part_f, T_f = a.bn_partial
            n_, h_, w_, c_ = c.y.shape
            coef = (g.empty((c_,), F32), g.empty((c_,), F32), g.empty((c_,), F32))
            ops.bn_bwd_coefficients(part_f, T_f, c_, float(n_) * h_ * w_, c.scale, c.mean, c.invstd, c.gamma.grad,
                                    c.beta.grad, coef, ws)
            a.bn_partial = None
            c.backward_from(dy, fused=(a.grad, c.y, coef, c.shift if relu else None))
            a.grad = None
            return
        if a.bn_partial is not None:
            part_f, T_f = a.bn_partial
            ops.bn_relu_bwd_apply(c.y, c.scale, c.shift, c.mean, c.invstd, a.grad, relu, part_f, T_f,
                                  c.gamma.grad, c.beta.grad, dy, ws)
            a.bn_partial = None
        else:
            ops.bn_relu_bwd(c.y, c.scale, c.shift, c.mean, c.invstd, a.grad, None, relu, 0, c.gamma.grad,
                            c.beta.grad, dy, ws)
        c.backward_from(dy)
        a.grad = None
    g.record(backward, (c.wv, c.gamma, c.beta))
    return a


def root_block(g, x4, scope="conv1", cout=64, is_training=True):
    """conv2d_same(inputs, 64, 7, stride=2) + BN + ReLU (nets/resnet_v1.py:193).  x4: prepared image."""
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.conv_bn_act(g, x4, cout, 7, scope, stride=2, is_training=is_training)
    n, h, w, _ = x4.shape
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (7, 7, 3, cout), variance_scaling(g.rng), regularized=True)
        gamma, beta, mm, mv = _bn_vars(g, cout)
    ws = g.workspace()

    def mk(old):
        t = old if old is not None else g.empty((7, 2, cout, 16))
        ops.pack_weights_stem(wv.data, t)
        return t
    w_stem = g.packed(wv, "stem", mk)
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = g.empty((n, oh, ow, cout))
    mt = ops.conv2d_stem_num_mtiles(n, h, w)
    part, stage = g.ws_small.two(mt * 2 * cout * 4, ops.bn_reduce_workspace(mt, cout))
    ops.conv2d_stem(x4.data, w_stem, y, CONV_STATS if is_training else 0, None, part if is_training else None)
    scale, shift = g.empty((cout,), F32), g.empty((cout,), F32)
    mean, invstd = g.empty((cout,), F32), g.empty((cout,), F32)
    if is_training:
        ops.bn_finalize(part, mt, cout, float(n) * oh * ow, gamma.data, beta.data, BN_EPS, BN_DECAY, mm.data,
                        mv.data, scale, shift, mean, invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, scale, shift)
    a = Act(g.empty(y.shape), name=scope)

    def fill():
        ops.bn_relu(y, scale, shift, True, 0, a._data, None)
    if is_training and FUSE_ROOT_POOL:
        # the 3x3/2 max-pool that follows (nets/resnet_v1.py:194) is the activation's only reader: it evaluates
        # relu(bn(y)) per window element (layers.max_pool2d) and the 64-channel half-resolution activation is never written
        a.deferred = fill
        a.bn_fwd = (y, scale, shift, True)
        a.takes_pool_grad = FUSE_ROOT_WGRAD and FUSE_ROOT_GATHER
    else:
        fill()

    def backward():
        if a.pool_grad is not None and a.grad is None:
            # the pool was the only contributor: its backward (the gather of the routed gradient) and this layer's BN
            # reduction are one pass — the gradient is summed while it is written for the weight gradient
            coef = (g.empty((cout,), F32), g.empty((cout,), F32), g.empty((cout,), F32))
            da = g.empty(y.shape)
            ops.bn_relu_bwd_reduce_pooled(y, scale, shift, mean, invstd, a.pool_grad, True, da, gamma.grad, beta.grad, coef, ws)
            ops.conv2d_stem_wgrad_bn(x4.data, da, y, shift, coef, True, wv.grad, ws)
            a.pool_grad = None
            return
        if a.pool_grad is not None:              # somebody else contributed as well: materialise the pool's share
            dap, argmax, pk, ps, pads = a.pool_grad
            ops.maxpool_bwd(a._data, dap, pk, ps, pads, a.grad, True, argmax=argmax, in_shape=a.shape)
            a.pool_grad = None
        if a.grad is None:
            return
        if FUSE_ROOT_WGRAD:
            # no input gradient: the weight gradient is the only reader of dy and applies the BN backward on load
            coef = (g.empty((cout,), F32), g.empty((cout,), F32), g.empty((cout,), F32))
            ops.bn_relu_bwd_reduce(y, scale, shift, mean, invstd, a.grad, True, gamma.grad, beta.grad, coef, ws)
            ops.conv2d_stem_wgrad_bn(x4.data, a.grad, y, shift, coef, True, wv.grad, ws)
        else:
            dy = g.empty(y.shape)
            ops.bn_relu_bwd(y, scale, shift, mean, invstd, a.grad, None, True, 0, gamma.grad, beta.grad, dy, ws)
            ops.conv2d_stem_wgrad(x4.data, dy, wv.grad, ws)
        a.grad = None
    g.record(backward, (wv, gamma, beta))
    return a


def bottleneck(g, x, depth, depth_bottleneck, stride, scope, is_training=True):
    """nets/resnet_v1.py:68-111: shortcut = subsample(x, stride) if depth == depth_in else
    1x1 conv(stride) + BN;  residual = 1x1 -> 3x3(stride) -> 1x1 with BN (ReLU, ReLU, none);
    output = relu(shortcut + residual).

    Training mode keeps the element-wise passes of the unit's widest tensors out of HBM where a 1x1 convolution can
    carry them: the output is DEFERRED to the next unit's first 1x1 convolution (FUSE_FWD), a projection
    shortcut's batch norm is applied inside that same expression and never stored, the output's ReLU mask travels to
    the backward pass as bits, and conv3's batch-norm backward apply runs inside conv3's input-gradient convolution
    (FUSE_BWD)."""
    from .layers import max_pool2d
    import torch
    depth_in = x.shape[-1]
    ws = g.workspace()
    owner = object()                       # identifies this unit's own consumers of x (conv1, shortcut)
    fuse_in = g.precision != "f32" and x.tail_ctx is not None and not x.consumed
    if fuse_in:
        # x is the previous unit's output.  This unit reads it twice (conv1 and the shortcut); whatever else
        # consumes it was built later and so contributes to its gradient EARLIER in the backward pass.  The
        # 1x1 convolution that makes the last of this unit's two contributions finishes x's gradient
        # (conv_bn_raw.backward_from): conv1 for an identity / subsampling shortcut (recorded first, runs
        # last), the projection shortcut otherwise (built before conv1 below).
        x.pending = 2
        x.pending_owner = owner
    projection = depth != depth_in
    sc_hold = {"dz": None}
    sc = None

    def make_shortcut():
        nonlocal sc
        if not projection:
            if stride == 1:
                return x
            return max_pool2d(g, x, 1, stride, scope="shortcut")
        if stride != 1:
            raise NotImplementedError("strided projection shortcut")
        if g.precision == "f32":
            return conv_bn_act(g, x, depth, 1, "shortcut", relu=False, is_training=is_training)
        # the projection's batch norm is applied where the shortcut is added (never stored on its own): keep the raw
        # conv output and its coefficients; its backward runs where the reference's shortcut closure ran (last)
        sc = conv_bn_raw(g, x, depth, 1, "shortcut", is_training=is_training, owner=owner)

        def sc_backward():
            dz = sc_hold["dz"]
            if dz is None:
                return
            dy = g.empty(sc.y.shape)
            if sc.can_fuse_bwd(wide=True):
                # reduction pass only; the apply step runs inside the projection's input-gradient convolution
                c_ = sc.y.shape[-1]
                coef = (g.empty((c_,), F32), g.empty((c_,), F32), g.empty((c_,), F32))
                ops.bn_relu_bwd_reduce(sc.y, sc.scale, sc.shift, sc.mean, sc.invstd, dz, False, sc.gamma.grad,
                                       sc.beta.grad, coef, ws)
                sc.backward_from(dy, fused=(dz, sc.y, coef, None))
            else:
                ops.bn_relu_bwd(sc.y, sc.scale, sc.shift, sc.mean, sc.invstd, dz, None, False, 0, sc.gamma.grad,
                                sc.beta.grad, dy, ws)
                sc.backward_from(dy)
            sc_hold["dz"] = None
        g.record(sc_backward, (sc.wv, sc.gamma, sc.beta))
        return Act(sc.y, name="shortcut_raw")
    with g.variable_scope(scope):
        with g.variable_scope("bottleneck_v1"):
            # variables are created in the reference's order (shortcut first) when there is a projection; a
            # subsampling shortcut has none and is recorded AFTER the residual branch, so that its backward
            # (zero insertion into x's gradient) runs before conv1's
            late = depth == depth_in and stride != 1
            shortcut = None if late else make_shortcut()
            r = conv_bn_act(g, x, depth_bottleneck, 1, "conv1", is_training=is_training, owner=owner)
            r = conv_bn_act(g, r, depth_bottleneck, 3, "conv2", stride=stride, is_training=is_training, defer=True)
            c3 = conv_bn_raw(g, r, depth, 1, "conv3", is_training=is_training)
            if late and fuse_in and FUSE_TAIL and FUSE_SUB and is_training:
                # the subsample's gradient is not zero-inserted into a full-size tensor: it waits in
                # x.sub_grad for conv1's input-gradient conv (the last contribution), whose epilogue adds it at
                # the even positions (ocr_conv2d_bnred_tail_f16)
                n_, h_, w_, c_ = x.shape
                ys = g.empty((n_, -(-h_ // stride), -(-w_ // stride), c_))
                ops.maxpool(x.data, 1, stride, (0, 0), ys, None)
                shortcut = Act(ys, name="shortcut")

                def sub_back():
                    if shortcut.grad is None:
                        return
                    if x.grad is None and x.pending == 2 and stride == 2:
                        x.sub_grad = shortcut.grad
                    else:                            # someone else contributed already: materialise
                        acc = x.grad is not None
                        if not acc:
                            x.grad = g.empty(x.shape)
                        ops.maxpool_bwd(x.data, shortcut.grad, 1, stride, (0, 0), x.grad, acc, in_shape=x.shape)
                    x.pending -= 1
                    shortcut.grad = None
                g.record(sub_back)
            elif late:
                shortcut = make_shortcut()
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.bn_add_relu(g, c3, shortcut, scope)
    out = Act(g.empty(c3.y.shape), name=scope)
    bits = None
    if is_training and FUSE_TAIL and MASK_BITS:
        bits = g.empty((out._data.numel() // 8,), torch.uint8)
    sc_scale, sc_shift = (sc.scale, sc.shift) if sc is not None else (None, None)
    short_t = shortcut.data                       # (an identity shortcut is x itself: computed by conv1 above)

    def fill():
        ops.bn_add_relu(c3.y, c3.scale, c3.shift, short_t, out._data, sc_scale, sc_shift, bits)
    if is_training and FUSE_FWD:
        out.deferred = fill
        out.tail_fwd = (c3.y, c3.scale, c3.shift, short_t, sc_scale, sc_shift, bits)
    else:
        fill()
    if is_training and FUSE_TAIL:
        out.tail_ctx = (c3.y, c3.mean, c3.invstd, out._data, bits)

    def backward():
        if out.grad is None:
            return
        dy = g.empty(c3.y.shape)
        if out.tail_partial is not None:
            # out.grad is already the gradient past the ReLU and the BN-backward sums are reduced
            dz = out.grad
            part_f, T_f = out.tail_partial
            if c3.can_fuse_bwd():
                n_, h_, w_, c_ = c3.y.shape
                coef = (g.empty((c_,), F32), g.empty((c_,), F32), g.empty((c_,), F32))
                ops.bn_bwd_coefficients(part_f, T_f, c_, float(n_) * h_ * w_, c3.scale, c3.mean, c3.invstd,
                                        c3.gamma.grad, c3.beta.grad, coef, ws)
                c3.backward_from(dy, fused=(dz, c3.y, coef, None))
            else:
                ops.bn_relu_bwd_apply(c3.y, c3.scale, c3.shift, c3.mean, c3.invstd, dz, False, part_f, T_f,
                                      c3.gamma.grad, c3.beta.grad, dy, ws)
                c3.backward_from(dy)
            out.tail_partial = None
        else:
            dz = g.empty(out.shape)
            ops.relu_bwd(out.data, out.grad, dz)
            ops.bn_relu_bwd(c3.y, c3.scale, c3.shift, c3.mean, c3.invstd, dz, None, False, 0, c3.gamma.grad,
                            c3.beta.grad, dy, ws)
            c3.backward_from(dy)
        if sc is not None:
            sc_hold["dz"] = dz                       # the projection's backward closure runs later (recorded first)
        elif shortcut.requires_grad:
            if shortcut.grad is None:
                shortcut.grad = dz
            else:
                ops.add_inplace(shortcut.grad, dz)
            if shortcut is x and x.pending is not None:
                x.pending -= 1
        out.grad = None
        out.pending = None
    g.record(backward, (c3.wv, c3.gamma, c3.beta))
    return out


# ----------------------------------------------------------------- EAST feature-merging branch
def unpool(g, x):
    """tf.image.resize_bilinear x2 (legacy sampling) on an f16 feature map (nets/model_vgg_16.py:15-16)."""
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.unpool(g, x)
    n, h, w, c = x.shape
    out = Act(g.empty((n, 2 * h, 2 * w, c)), name="unpool")
    ops.unpool_f16(x.data, out.data)

    def backward():
        if out.grad is None or not x.requires_grad:
            return
        acc = x.grad is not None
        if not acc:
            x.grad = g.empty(x.shape)
        ops.unpool_bwd_f16(out.grad, x.grad, acc)
        out.grad = None
    g.record(backward)
    return out


def concat_conv_bn_relu(g, xa, xb, cout, scope, is_training=True):
    """slim.conv2d(tf.concat([xa, xb], axis=-1), cout, 1) + BN + ReLU (nets/model_vgg_16.py:118) without
    materialising the concatenation: conv(xa, W[:ca]) + conv(xb, W[ca:]) through the accumulating
    epilogue; ONE variable `<scope>/weights` [1,1,ca+cb,cout] as in the reference."""
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.concat_conv_bn_relu(g, xa, xb, cout, scope, is_training=is_training)
    n, h, w, ca = xa.shape
    cb = xb.shape[-1]
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (1, 1, ca + cb, cout), variance_scaling(g.rng), regularized=True)
        gamma, beta, mm, mv = _bn_vars(g, cout)
    ws = g.workspace()

    def mk(old):
        if old is None:
            old = [g.empty((1, cout, ca)), g.empty((1, ca, cout)), g.empty((1, cout, cb)), g.empty((1, cb, cout))]
        ops.pack_weights(wv.data[:, :, :ca, :], old[0], old[1])
        ops.pack_weights(wv.data[:, :, ca:, :], old[2], old[3])
        return old
    wa_kc, wa_ck, wb_kc, wb_ck = g.packed(wv, "cat", mk)
    da = ops.conv_desc((n, h, w, ca), cout, 1, 1)
    db = ops.conv_desc((n, h, w, cb), cout, 1, 1)
    y = g.empty((n, h, w, cout))
    mt = ops.conv2d_num_mtiles(da)
    part, stage = g.ws_small.two(mt * 2 * cout * 4, ops.bn_reduce_workspace(mt, cout))
    da.flags = 0
    ops.conv2d(da, xa.data, wa_kc, y, None, None)
    db.flags = CONV_ACCUM_F16 | (CONV_STATS if is_training else 0)
    ops.conv2d(db, xb.data, wb_kc, y, None, part if is_training else None)
    scale, shift = g.empty((cout,), F32), g.empty((cout,), F32)
    mean, invstd = g.empty((cout,), F32), g.empty((cout,), F32)
    if is_training:
        ops.bn_finalize(part, mt, cout, float(n) * h * w, gamma.data, beta.data, BN_EPS, BN_DECAY, mm.data,
                        mv.data, scale, shift, mean, invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, scale, shift)
    a = Act(g.empty(y.shape), name=scope)
    ops.bn_relu(y, scale, shift, True, 0, a.data, None)

    def backward():
        if a.grad is None:
            return
        dy = g.empty(y.shape)
        ops.bn_relu_bwd(y, scale, shift, mean, invstd, a.grad, None, True, 0, gamma.grad, beta.grad, dy, ws)
        for x, c, w_ck, sl in ((xa, ca, wa_ck, slice(0, ca)), (xb, cb, wb_ck, slice(ca, ca + cb))):
            dd = ops.conv_desc((n, h, w, c), cout, 1, 1)
            ops.conv2d_wgrad(dd, x.data, dy, wv.grad[:, :, sl, :], g.ws_wgrad)
            if x.requires_grad:
                flags = 0
                if x.grad is None:
                    x.grad = g.empty(x.shape)
                else:
                    flags = CONV_ACCUM_F16
                dg = ops.ConvDesc(n, h, w, cout, h, w, c, 1, 1, 1, 1, 0, 0, 1, flags)
                ops.conv2d(dg, dy, w_ck, x.grad, None, None)
        a.grad = None
    g.record(backward, (wv, gamma, beta))
    return a


MERGE_HEADS = __import__("os").environ.get("OCR_RESNET_MERGE_HEADS", "1") == "1"   # F_score + geo_map: one pass over the feature


def sigmoid_heads(g, feat, couts, scopes):
    """The two sigmoid heads on the merge branch's output (nets/model_vgg_16.py:129-131: F_score = conv1x1 -> 1,
    geo_map = conv1x1 -> 8, both sigmoid, biases, no normaliser) as ONE merged 1x1 convolution over the feature
    (layers.head_conv_bias: variable `<scope0>+<scope1>/weights` [cin, 1+8], split back into the reference's two by
    checkpoint.internal_to_tf) — one pass over the 105 MB feature for the forward, the weight gradient and the input
    gradient instead of two each.  Returns the two activation handles."""
    from .layers import SmallAct, head_conv_bias
    if g.precision == "f32" or not MERGE_HEADS:
        return tuple(sigmoid_head(g, feat, c, s) for c, s in zip(couts, scopes))
    z, _, _ = head_conv_bias(g, feat, tuple(scopes), tuple(couts))
    n, h, w, _ = z.data.shape
    o0 = SmallAct(g.empty((n, h, w, couts[0]), F32))
    o1 = SmallAct(g.empty((n, h, w, couts[1]), F32))
    ops.sc_sigmoid_split(z.data, couts[0], o0.data, o1.data)

    def backward():
        if o0.grad is None and o1.grad is None:
            return
        z.grad = g.empty(z.data.shape, F32)
        ops.sc_sigmoid_split_bwd(o0.data, o0.grad, o1.data, o1.grad, z.grad)
        o0.grad = o1.grad = None
    g.record(backward)
    return o0, o1


MERGE_REORDER = __import__("os").environ.get("OCR_RESNET_MERGE_REORDER", "1") == "1"   # 1x1 merge conv before the upsample


def unpool_concat_conv_bn_relu(g, lo, xb, cout, scope, is_training=True):
    """slim.conv2d(tf.concat([unpool(lo), xb], axis=-1), cout, 1) + BN + ReLU (nets/model_vgg_16.py:118-121) with the
    upsampled branch's share of the convolution taken BEFORE the resize: conv(unpool(lo), Wa) = unpool(conv(lo, Wa)) —
    a 1x1 convolution mixes channels, the bilinear resize mixes positions — so that share runs on a quarter of the pixels
    and the upsampled tensor (2048 channels at 40^2, 419 MB at 64 x 640^2) and its gradient are never written.
    y = conv(xb, Wb) + unpool(conv(lo, Wa)); ONE variable `<scope>/weights` [1,1,ca+cb,cout], channel order of the
    reference's concat (upsampled branch first)."""
    if g.precision == "f32" or not MERGE_REORDER:
        return concat_conv_bn_relu(g, unpool(g, lo), xb, cout, scope, is_training=is_training)
    n, lh, lw, ca = lo.shape
    _, h, w, cb = xb.shape
    assert (h, w) == (2 * lh, 2 * lw)
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (1, 1, ca + cb, cout), variance_scaling(g.rng), regularized=True)
        gamma, beta, mm, mv = _bn_vars(g, cout)
    ws = g.workspace()

    def mk(old):
        if old is None:
            old = [g.empty((1, cout, ca)), g.empty((1, ca, cout)), g.empty((1, cout, cb)), g.empty((1, cb, cout))]
        ops.pack_weights(wv.data[:, :, :ca, :], old[0], old[1])
        ops.pack_weights(wv.data[:, :, ca:, :], old[2], old[3])
        return old
    wa_kc, wa_ck, wb_kc, wb_ck = g.packed(wv, "cat", mk)
    da = ops.conv_desc((n, lh, lw, ca), cout, 1, 1)
    db = ops.conv_desc((n, h, w, cb), cout, 1, 1)
    t = g.empty((n, lh, lw, cout))
    y = g.empty((n, h, w, cout))
    da.flags = 0
    ops.conv2d(da, lo.data, wa_kc, t, None, None)
    db.flags = 0
    ops.conv2d(db, xb.data, wb_kc, y, None, None)
    T = ops.channel_stats_num_partials(n * h * w, cout)
    part, stage = g.ws_small.two(T * 2 * cout * 4, ops.bn_reduce_workspace(T, cout))
    ops.unpool_add_stats(t, y, part if is_training else None)
    scale, shift = g.empty((cout,), F32), g.empty((cout,), F32)
    mean, invstd = g.empty((cout,), F32), g.empty((cout,), F32)
    if is_training:
        ops.bn_finalize(part, T, cout, float(n) * h * w, gamma.data, beta.data, BN_EPS, BN_DECAY, mm.data,
                        mv.data, scale, shift, mean, invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, scale, shift)
    a = Act(g.empty(y.shape), name=scope)
    ops.bn_relu(y, scale, shift, True, 0, a.data, None)

    def backward():
        if a.grad is None:
            return
        dy = g.empty(y.shape)
        ops.bn_relu_bwd(y, scale, shift, mean, invstd, a.grad, None, True, 0, gamma.grad, beta.grad, dy, ws)
        dt = g.empty(t.shape)
        ops.unpool_bwd_f16(dy, dt, False)                # gradient of conv(lo, Wa): the resize's transpose on cout channels
        for x, c, w_ck, sl, d_out, (hh, ww) in ((lo, ca, wa_ck, slice(0, ca), dt, (lh, lw)),
                                                (xb, cb, wb_ck, slice(ca, ca + cb), dy, (h, w))):
            dd = ops.conv_desc((n, hh, ww, c), cout, 1, 1)
            ops.conv2d_wgrad(dd, x.data, d_out, wv.grad[:, :, sl, :], g.ws_wgrad)
            if x.requires_grad:
                flags = 0
                if x.grad is None:
                    x.grad = g.empty(x.shape)
                else:
                    flags = CONV_ACCUM_F16
                dg = ops.ConvDesc(n, hh, ww, cout, hh, ww, c, 1, 1, 1, 1, 0, 0, 1, flags)
                ops.conv2d(dg, d_out, w_ck, x.grad, None, None)
        a.grad = None
    g.record(backward, (wv, gamma, beta))
    return a


def sigmoid_head(g, feat, cout, scope):
    """slim.conv2d(feat, cout, 1, activation_fn=tf.nn.sigmoid, normalizer_fn=None) (model_vgg_16.py:129-131)."""
    from .layers import SmallAct, head_conv_bias
    z, _, _ = head_conv_bias(g, feat, (scope,), (cout,))
    out = SmallAct(g.empty(z.data.shape, F32))
    ops.sc_sigmoid(z.data, out.data)

    def backward():
        if out.grad is None:
            return
        z.grad = g.empty(z.data.shape, F32)
        ops.sc_sigmoid_bwd(out.data, out.grad, z.grad)
        out.grad = None
    g.record(backward)
    return out
