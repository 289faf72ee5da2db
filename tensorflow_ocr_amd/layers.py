"""slim-style layers (conv2d [+batch_norm] [+relu] [+max_pool2d], heads) driving the HIP kernels.

Each function runs the forward kernels immediately and records ONE backward closure on the
graph's tape.  Variable names follow tf.contrib.slim so checkpoints/state-dicts line up with
the reference: `<scope>/weights`, `<scope>/biases`,
`<scope>/BatchNorm/{gamma,beta,moving_mean,moving_variance}`.
"""
import torch

from . import ops
from .graph import Act, F16, F32, constant, variance_scaling, xavier_uniform
from ._lib import CONV_ACCUM_F16, CONV_BIAS, CONV_RELU, CONV_STATS

BN_DECAY = 0.997
BN_EPS = 1e-5
FUSE_BN_REDUCE = __import__("os").environ.get("OCR_FUSE_BN", "1") == "1"   # BN-backward reduction inside the consumer's input-gradient kernel


def _bn_vars(g, C):
    with g.variable_scope("BatchNorm"):
        gamma = g.get_variable("gamma", (C,), constant(1.0))
        beta = g.get_variable("beta", (C,), constant(0.0))
        mm = g.get_variable("moving_mean", (C,), constant(0.0), trainable=False)
        mv = g.get_variable("moving_variance", (C,), constant(1.0), trainable=False)
    return gamma, beta, mm, mv


def _packs(g, w, first):
    """f16 operand layouts of a conv weight: forward [tap][cout][cin] and dgrad [tap][cin][cout]."""
    kh, kw, cin, cout = w.shape
    if first:
        def mk(old):
            t = old if old is not None else g.empty((3, cout, 16))
            ops.pack_weights_first(w.data, t)
            return t
        return g.packed(w, "first", mk), None

    def mk(old):
        if old is None:
            old = (g.empty((kh * kw, cout, cin)), g.empty((kh * kw, cin, cout)))
        ops.pack_weights(w.data, old[0], old[1])
        return old
    return g.packed(w, "kc_ck", mk)


FUSE_BIAS_POOL = __import__("os").environ.get("OCR_FUSE_BIAS_POOL", "1") == "1"   # bias nets: conv1_2's bias + ReLU + pool in the conv epilogue
FUSE_BIAS_RELU = __import__("os").environ.get("OCR_FUSE_BIAS_RELU", "1") == "1"   # bias nets: ReLU mask + bias gradient in the consumer's dgrad epilogue


def _const_vec(g, n, value):
    """A cached per-graph constant f32 vector (filled once, outside any recorded step)."""
    cache = g.__dict__.setdefault("_const_vecs", {})
    t = cache.get((n, value))
    if t is None:
        t = cache[(n, value)] = torch.full((n,), float(value), dtype=F32, device=g.device)
    return t


FUSE_BN_W4_MAXHW = int(__import__("os").environ.get("OCR_FUSE_BN_W4_MAXHW", "1000000"))
FUSE_BN_POOL_REDUCE = __import__("os").environ.get("OCR_FUSE_BN_POOL_REDUCE", "1") == "1"    # measurement switch
FIRST_RECOMPUTE = __import__("os").environ.get("OCR_FIRST_RECOMPUTE", "1") == "1"            # measurement switch (forward: 439 -> 318 us)
FIRST_DROP_Y = __import__("os").environ.get("OCR_FIRST_DROP_Y", "1") == "1"                  # conv1_1's y is never stored (ops.LazyFirstY)
FIRST_MOMENTS = __import__("os").environ.get("OCR_FIRST_MOMENTS", "1") == "1"                # ... and its statistics come from the image's moments
# conv1_1's weight gradient recomputing y even when y IS stored: bit-identical, but no faster by itself (413 vs 402 us at
# 32 x 512^2: with one stream instead of two the kernel is bound by its per-tile LDS work, not by HBM).  With
# FIRST_DROP_Y (y never stored) the recomputing form is what runs regardless of this switch.
FIRST_WGRAD_RECOMPUTE = __import__("os").environ.get("OCR_FIRST_WGRAD_RECOMPUTE", "0") == "1"
GUEST_REDUCE = __import__("os").environ.get("OCR_GUEST_REDUCE", "1") == "1"            # end-point layers' BN-backward reduction as a guest pass
FUSE_FIRST_WGRAD = __import__("os").environ.get("OCR_FUSE_FIRST_WGRAD", "1") == "1"    # measurement switch
# conv1_1's weight gradient from SUMS (csrc/conv_first.hip: first_wgrad_sums_kernel): conv1_2's input-gradient launch
# leaves S1 = V^T dz instead of the 1 GiB gradient, dW = A .* S1 + B .* (M W) + C .* m
FIRST_WGRAD_SUMS = __import__("os").environ.get("OCR_FIRST_WGRAD_SUMS", "1") == "1"


def conv2d(g, x, cout, k, scope, *, stride=1, rate=1, normalizer="bn", relu=True, pool=0,
           keep_full=True, is_training=True, bn_training=None, first=False, weight_decay=True,
           initializer=None):
    """slim.conv2d(+batch_norm)(+ReLU), optionally fused with the following 2x2/2 max-pool.

    x: Act f16 [n,h,w,cin] (for `first`: the [n,h,w,4] prepared image).
    normalizer: "bn" (no bias, slim drops it) or None (bias).
    Returns (full, pooled) Acts; `full` is None when keep_full=False and pool>0.
    Reference: nets/vgg.py:14-39 under resnet_arg_scope (nets/model_vgg_16.py:144) or the
    PixelLink bias scope (nets/pixellink.py:41-48).
    """
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.conv2d(g, x, cout, k, scope, stride=stride, rate=rate, normalizer=normalizer,
                                 relu=relu, pool=pool, keep_full=keep_full, is_training=is_training,
                                 bn_training=bn_training, first=first, weight_decay=weight_decay,
                                 initializer=initializer)
    n, h, w, cin_x = x.shape
    cin = 3 if first else cin_x
    bn_training = is_training if bn_training is None else bn_training
    with g.variable_scope(scope):
        init = initializer or variance_scaling(g.rng)
        wv = g.get_variable("weights", (k, k, cin, cout), init, regularized=weight_decay)
        if normalizer == "bn":
            gamma, beta, mm, mv = _bn_vars(g, cout)
            bias = None
        else:
            bias = g.get_variable("biases", (cout,), constant(0.0))
    ws = g.workspace()
    w_fwd, w_dg = _packs(g, wv, first)
    if first:
        d = None
        oh, ow = h, w
        mt = ops.conv2d_first_num_mtiles(n, h, w)
    else:
        d = ops.conv_desc((n, h, w, cin), cout, k, k, stride, rate)
        oh, ow = d.oh, d.ow
        mt = ops.conv2d_num_mtiles(d)
    # conv1_1 under training-mode batch norm: y is never stored — the statistics pass writes nothing, the activation and
    # the two readers of y in the backward pass (conv1_2's fused BN-backward sums, conv1_1's own weight gradient)
    # evaluate the 3-channel convolution again; a consumer that cannot asks the LazyFirstY for the tensor
    drop_y = (first and normalizer == "bn" and not pool and cout == 64 and FIRST_RECOMPUTE and FIRST_DROP_Y
              and (is_training if bn_training is None else bn_training) and FUSE_FIRST_WGRAD)
    y = ops.LazyFirstY(g.empty, x.data, w_fwd, (n, oh, ow, cout)) if drop_y else g.empty((n, oh, ow, cout))

    if normalizer == "bn":
        train_stats = bn_training
        flags = CONV_STATS if train_stats else 0
        part, stage = g.ws_small.two(mt * 2 * cout * 4, ops.bn_reduce_workspace(mt, cout))
        mt_fin = mt
        if first and drop_y and train_stats and FIRST_MOMENTS:
            # the statistics from the image's second moments (csrc/conv_first.hip: first_moments_kernel): no pass over the
            # 64-channel output at all; one partial row
            if FIRST_WGRAD_SUMS:
                y.moments = g.empty((32 * 32,), torch.float64)      # kept for the backward (ops.conv2d_first_wgrad_sums)
            ops.conv2d_first_moments(x.data, w_fwd, part, cout, ws, moments=y.moments)
            mt_fin = 1
        elif first:
            ops.conv2d_first(x.data, w_fwd, None if drop_y else y, flags, None, part if train_stats else None, cout=cout)
        else:
            d.flags = flags
            ops.conv2d(d, x.data, w_fwd, y, None, part if train_stats else None)
        scale, shift = g.empty((cout,), F32), g.empty((cout,), F32)
        mean, invstd = g.empty((cout,), F32), g.empty((cout,), F32)
        if train_stats:
            ops.bn_finalize(part, mt_fin, cout, float(n) * oh * ow, gamma.data, beta.data, BN_EPS, BN_DECAY,
                            mm.data, mv.data, scale, shift, mean, invstd, stage)
        else:
            ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, scale, shift)
        full = pooled = None
        argmax = y_pool = argmax_full = None
        if pool:
            pooled = g.empty((n, (oh + 1) // 2, (ow + 1) // 2, cout))
            if keep_full:
                full = g.empty((n, oh, ow, cout))
                if train_stats and relu and ops.guest_apply_ok((n, oh, ow, cout)) and oh % 2 == 0 and ow % 2 == 0:
                    # an end point that is also pooled (conv3_3 / conv4_3): keep the first-max positions, so that the
                    # backward's apply pass can run as a guest (ops.bn_relu_poolfull_bwd_apply_affine) without
                    # re-deriving them
                    argmax_full = g.empty(pooled.shape, torch.uint8)
                    ops.bn_relu_pool_idx(y, scale, shift, relu, full, pooled, argmax_full, None)
                else:
                    ops.bn_relu(y, scale, shift, relu, 2, full, pooled)
            else:
                # the pool is the only consumer: keep the first-max position so that the backward routes
                # the pooled gradient without re-deriving the four candidates' activations
                argmax = g.empty(pooled.shape, torch.uint8)
                # ... and, while training, the conv output AT that position: dz is zero everywhere else, so the conv
                # that consumes the pooled activation can sum this layer's BN-backward terms over the pooled
                # positions in its input-gradient epilogue (a_pool.bn_ctx below) and the reduction pass over y goes
                y_pool = g.empty(pooled.shape) if (train_stats and FUSE_BN_POOL_REDUCE) else None
                ops.bn_relu_pool_idx(y, scale, shift, relu, None, pooled, argmax, y_pool)
        else:
            full = g.empty((n, oh, ow, cout))
            if first and FIRST_RECOMPUTE:
                # conv1_1: evaluating the 3-channel convolution again costs 67 MB of reads, the element-wise pass 1 GiB
                ops.conv2d_first_bn_relu(x.data, w_fwd, scale, shift, relu, full)
            else:
                ops.bn_relu(y.tensor() if isinstance(y, ops.LazyFirstY) else y, scale, shift, relu, 0, full, None)
        a_full = Act(full, name=scope) if full is not None else None
        a_pool = Act(pooled, name=scope + "/pool") if pooled is not None else None
        if not pool and train_stats:
            # lets the consumer conv's input-gradient kernel do this layer's BN-backward reduction
            a_full.bn_ctx = (y, scale, shift, mean, invstd, relu)
        if pool and argmax is not None and y_pool is not None:
            a_pool.bn_ctx = (y_pool, scale, shift, mean, invstd, relu)

        def backward():
            if not train_stats:
                raise NotImplementedError("backward through inference-mode batch norm")
            da_full = a_full.grad if a_full is not None else None
            da_pool = a_pool.grad if a_pool is not None else None
            if first and a_full is not None and a_full.first_s1 is not None:
                # conv1_1 whose sole consumer left the sums instead of the gradient (_conv_dgrad): dgamma / dbeta and the
                # apply coefficients from the partial rows, then dW = A .* S1 + B .* (M W) + C .* m — no pass over a
                # 64-channel tensor at all
                part_f, T_f = a_full.bn_partial
                coef = (g.empty((cout,), F32), g.empty((cout,), F32), g.empty((cout,), F32))
                ops.bn_bwd_coefficients(part_f, T_f, cout, float(n) * oh * ow, scale, mean, invstd, gamma.grad,
                                        beta.grad, coef, ws)
                ops.conv2d_first_wgrad_sums(a_full.first_s1, y.moments, w_fwd, coef, wv.grad)
                a_full.bn_partial = None
                a_full.first_s1 = None
                return
            if da_full is None and da_pool is None:
                return
            if pool and da_pool is None:
                da_pool = g.empty(a_pool.data.shape)
                ops.fill_(da_pool, 0.0)
            if not pool and da_full is None:
                return
            if first and not pool and a_full.bn_partial is not None and FUSE_FIRST_WGRAD:
                # conv1_1: no input gradient, so the weight gradient is the only reader of dy — it applies the BN
                # backward while staging its tiles and dy is never written (ocr_conv2d_first_wgrad_bn_f16)
                part_f, T_f = a_full.bn_partial
                coef = (g.empty((cout,), F32), g.empty((cout,), F32), g.empty((cout,), F32))
                ops.bn_bwd_coefficients(part_f, T_f, cout, float(n) * oh * ow, scale, mean, invstd, gamma.grad,
                                        beta.grad, coef, ws)
                lazy = isinstance(y, ops.LazyFirstY)
                if lazy and y.t is not None:
                    lazy, y_t = False, y.t                # somebody had it evaluated: read it
                else:
                    y_t = None if lazy else y
                ops.conv2d_first_wgrad_bn(x.data, da_full, y_t, shift, coef, relu, wv.grad, ws,
                                          w_first=w_fwd if (lazy or FIRST_WGRAD_RECOMPUTE) else None)
                a_full.bn_partial = None
                a_full.grad = None
                return
            if isinstance(y, ops.LazyFirstY):
                y_b = y.tensor()                  # (the unfused routes below read y)
            else:
                y_b = y
            dy = g.empty(y_b.shape)
            if not pool and a_full.bn_partial is not None:
                part_f, T_f = a_full.bn_partial
                if ops.guest_apply_ok(y_b.shape):
                    # the apply pass as a GUEST (csrc/guest_bn.hip): dgamma / dbeta / coefficients first, then one slim
                    # launch that the recorded step runs beside the weight gradient of the layer above
                    coef = (g.empty((cout,), F32), g.empty((cout,), F32), g.empty((cout,), F32))
                    ops.bn_bwd_coefficients_pre(part_f, T_f, cout, float(n) * oh * ow, scale, mean, invstd, gamma.grad,
                                                beta.grad, coef, ws)
                    ops.bn_relu_bwd_apply_affine(y_b, da_full, scale, shift, coef[1], coef[2], relu, dy)
                else:
                    ops.bn_relu_bwd_apply(y_b, scale, shift, mean, invstd, da_full, relu, part_f, T_f,
                                          gamma.grad, beta.grad, dy, ws)
                a_full.bn_partial = None
            elif pool and argmax is not None and da_full is None and a_pool.bn_partial is not None:
                part_p, T_p = a_pool.bn_partial
                if ops.guest_apply_ok(y_b.shape) and oh % 2 == 0 and ow % 2 == 0:
                    coef = (g.empty((cout,), F32), g.empty((cout,), F32), g.empty((cout,), F32))
                    ops.bn_bwd_coefficients_pre(part_p, T_p, cout, float(n) * oh * ow, scale, mean, invstd, gamma.grad,
                                                beta.grad, coef, ws)
                    ops.bn_relu_pool_bwd_idx_apply_affine(y_b, argmax, da_pool, coef, relu, dy)
                else:
                    ops.bn_relu_pool_bwd_idx_apply(y_b, scale, mean, invstd, argmax, da_pool, relu, part_p, T_p,
                                                   gamma.grad, beta.grad, dy, ws)
                a_pool.bn_partial = None
            elif pool and argmax is not None and da_full is None:
                ops.bn_relu_pool_bwd_idx(y_b, scale, mean, invstd, a_pool.data, argmax, da_pool, relu,
                                         gamma.grad, beta.grad, dy, ws)
            elif (not pool or argmax_full is not None) and da_full is not None and ops.guest_apply_ok(y_b.shape):
                # no consumer summed this layer's terms (several contributed to its gradient: an end point of the fuse
                # heads): the reduction pass stays, its finalize emits the coefficients, the apply pass is a guest
                coef = (g.empty((cout,), F32), g.empty((cout,), F32), g.empty((cout,), F32))
                if GUEST_REDUCE and (relu or not pool):
                    # ... as a guest as well (beside weight gradients still held back), its rows folded on the main stream
                    part_e, T_e = ops.bn_relu_bwd_reduce_rows(y_b, da_full, scale, shift, mean, invstd, relu, g.empty,
                                                              da_pool=da_pool if pool else None,
                                                              argmax=argmax_full if pool else None)
                    ops.bn_bwd_coefficients_pre(part_e, T_e, cout, float(n) * oh * ow, scale, mean, invstd, gamma.grad,
                                                beta.grad, coef, ws)
                else:
                    ops.bn_relu_bwd_reduce(y_b, scale, shift, mean, invstd, da_full, relu, gamma.grad, beta.grad, coef, ws,
                                           da_pool=da_pool if pool else None)
                if pool:
                    ops.bn_relu_poolfull_bwd_apply_affine(y_b, da_full, da_pool, argmax_full, scale, shift, coef, relu, dy)
                else:
                    ops.bn_relu_bwd_apply_affine(y_b, da_full, scale, shift, coef[1], coef[2], relu, dy)
            else:
                ops.bn_relu_bwd(y_b, scale, shift, mean, invstd, da_full, da_pool, relu, 2 if pool else 0,
                                gamma.grad, beta.grad, dy, ws)
            _conv_backward(g, x, wv, w_dg, d, dy, first)
            if a_full is not None:
                a_full.grad = None
            if a_pool is not None:
                a_pool.grad = None
        g.record(backward, (wv, gamma, beta))
        return a_full, a_pool

    # bias (+ReLU) path: PixelLinkNet's VGG (nets/pixellink.py:41-48)
    flags = CONV_BIAS | (CONV_RELU if relu else 0)
    if (pool == 2 and relu and not keep_full and not first and FUSE_BIAS_POOL
            and ops.conv2d_variant(d) == "conv_c64_persist_kernel<64>"):
        # conv1_2 (64 -> 64 at full resolution) followed by its pool, the full-resolution activation wanted by nobody:
        # bias + ReLU + 2x2 max-pool inside the convolution's epilogue, only the pooled activation (+ first-max
        # positions while training) is written — 1/4 of the bytes and no pooling pass
        ph, pw_ = (oh + 1) // 2, (ow + 1) // 2
        pooled = g.empty((n, ph, pw_, cout))
        argmax = g.empty((n, ph, pw_, cout), torch.uint8) if is_training else None
        d.flags = flags
        ops.conv2d_relu_pool(d, x.data, w_fwd, bias.data, pooled, argmax)
        a_pool = Act(pooled, name=scope + "/pool")

        def backward_pool():
            if a_pool.grad is None:
                return
            if argmax is None:
                # an is_training=False build is forward-only here as it is under batch norm: the first-max positions
                # the routing needs were not kept, and the full-resolution activation was never written
                raise NotImplementedError("backward through a conv + ReLU + pool built with is_training=False "
                                          "(the first-max positions are kept only while training)")
            # ReLU mask and bias gradient on the pooled tensors (a window maximum is positive iff the element its
            # gradient is routed to is), then the routed gradient IS dz of the convolution
            dzp = g.empty(pooled.shape)
            ops.bias_relu_bwd(pooled, a_pool.grad, True, dzp, bias.grad, ws)
            dz = g.empty((n, oh, ow, cout))
            ops.maxpool_bwd(None, dzp, 2, 2, (0, 0), dz, False, argmax=argmax, in_shape=(n, oh, ow, cout))
            _conv_backward(g, x, wv, w_dg, d, dz, first)
            a_pool.grad = None
        g.record(backward_pool, (wv, bias))
        return None, a_pool
    if first:
        ops.conv2d_first(x.data, w_fwd, y, flags, bias.data, None)
    else:
        d.flags = flags
        ops.conv2d(d, x.data, w_fwd, y, bias.data, None)
    a_full = Act(y, name=scope)
    if relu and is_training:
        # (activation, scale 1, shift 0, mean 0, 1/std 1, relu): lets the ONE convolution that consumes this activation
        # store the gradient past the ReLU and sum the bias gradient in its input-gradient epilogue (_conv_backward);
        # used only when the net marks the activation `sole_consumer` (nets/vgg.py)
        a_full.bias_ctx = (y, _const_vec(g, cout, 1.0), _const_vec(g, cout, 0.0), _const_vec(g, cout, 0.0),
                           _const_vec(g, cout, 1.0), True)

    def backward_bias():
        if a_full.grad is None:
            return
        if a_full.bias_done:
            dz = a_full.grad                       # masked and summed on the POOLED tensors (max_pool2d below)
            a_full.bias_done = False
        elif a_full.bias_partial is not None:
            # a_full.grad IS dz already (masked by the consumer's input-gradient kernel); its partial rows sum to dbias
            part_f, T_f = a_full.bias_partial
            ops.bn_bwd_sums(part_f, T_f, cout, bias.grad, g.ws_small.get(cout * 4)[:cout * 4].view(F32), ws)
            dz = a_full.grad
            a_full.bias_partial = None
        else:
            dz = g.empty(y.shape)
            ops.bias_relu_bwd(y, a_full.grad, relu, dz, bias.grad, ws)
        _conv_backward(g, x, wv, w_dg, d, dz, first)
        a_full.grad = None
    g.record(backward_bias, (wv, bias))
    a_pool = max_pool2d(g, a_full, 2, 2, scope=scope + "/pool", bias_relu=bias if (relu and is_training) else None) if pool else None
    return a_full, a_pool


def _conv_backward(g, x, wv, w_dg, d, dy, first):
    """Weight gradient (always) and input gradient (when the input needs one)."""
    ws = g.workspace()
    if first:
        ops.conv2d_first_wgrad(x.data, dy, wv.grad, ws)
        return
    dd = ops.ConvDesc(d.n, d.h, d.w, d.cin, d.oh, d.ow, d.cout, d.kh, d.kw, d.stride, d.dilation,
                      d.pad_top, d.pad_left, 0, 0)
    if ops.GUEST_BN and x.requires_grad:
        # input gradient FIRST: the layer below can then start its backward while this layer's weight gradient runs —
        # its batch-norm apply pass is a guest beside it (csrc/guest_bn.hip; the recorded step pairs the two)
        _conv_dgrad(g, x, wv, w_dg, d, dy)
        ops.conv2d_wgrad(dd, x.data, dy, wv.grad, g.ws_wgrad, alloc=lambda nb: g.empty((nb,), torch.uint8))
        return
    ops.conv2d_wgrad(dd, x.data, dy, wv.grad, g.ws_wgrad)
    if not x.requires_grad:
        return
    _conv_dgrad(g, x, wv, w_dg, d, dy)


def _conv_dgrad(g, x, wv, w_dg, d, dy):
    if d.stride != 1:
        raise NotImplementedError("strided dgrad")
    # dx = conv(dy, W^T flipped): input = dy [n,oh,ow,cout], output = [n,h,w,cin]
    pt = d.dilation * (d.kh - 1) - d.pad_top
    pl = d.dilation * (d.kw - 1) - d.pad_left
    flags = 0
    by0 = x.bn_ctx[0] if x.bn_ctx is not None else None
    if (x.grad is None and x.sole_consumer and FIRST_WGRAD_SUMS and FUSE_BN_REDUCE and isinstance(by0, ops.LazyFirstY)
            and by0.t is None and by0.moments is not None and d.cin == 64):
        # conv1_1's activation, read by this convolution only: its gradient has one more reader — conv1_1's weight
        # gradient — which needs sums of it, not the tensor; the launch leaves those (epilogue mode 7) and nothing else
        dg = ops.ConvDesc(d.n, d.oh, d.ow, d.cout, d.h, d.w, d.cin, d.kh, d.kw, 1, d.dilation, pt, pl, 1, 0)
        blocks = ops.conv2d_bnred_first_wgrad_blocks(dg)
        if blocks > 0:
            T = ops.conv2d_num_mtiles(dg)
            partial = g.empty((T, 2, d.cin), F32)
            s1 = g.empty((blocks, 32, 64), F32)
            ops.conv2d_bnred_first_wgrad(dg, dy, w_dg, partial, (by0.x4, by0.w_first) + tuple(x.bn_ctx[1:]), s1)
            x.bn_partial = (partial, T)
            x.first_s1 = s1
            return
    if x.grad is None:
        x.grad = g.empty(x.shape)
    else:
        flags |= CONV_ACCUM_F16
        x.bn_partial = None     # an earlier consumer's fused BN-backward sums no longer cover the full gradient
        x.bias_partial = None   # ... nor its masked store (backward_bias then masks the whole sum again: idempotent)
    dg = ops.ConvDesc(d.n, d.oh, d.ow, d.cout, d.h, d.w, d.cin, d.kh, d.kw, 1, d.dilation, pt, pl, 1,
                      flags)
    if x.bias_ctx is not None and x.sole_consumer and not flags and FUSE_BIAS_RELU:
        # bias + ReLU producer whose only reader is this convolution: dz and the bias-gradient partial sums come out of
        # this kernel's epilogue, the producer's bias_relu_bwd pass (read a, read da, write dz) disappears
        T = ops.conv2d_num_mtiles(dg)
        partial = g.empty((T, 2, d.cin), F32)
        ops.conv2d_bnred(dg, dy, w_dg, x.grad, partial, x.bias_ctx, store_masked=True)
        x.bias_partial = (partial, T)
        return
    fuse = x.bn_ctx is not None and not flags and FUSE_BN_REDUCE
    if fuse and d.h >= FUSE_BN_W4_MAXHW and ops.conv2d_variant(dg) == "conv3x3_w4_kernel":
        fuse = False        # (measurement switch: the one-wave-per-SIMD kernel cannot overlap its epilogue with anything)
    if fuse:
        T = ops.conv2d_num_mtiles(dg)
        partial = g.empty((T, 2, d.cin), F32)
        by = x.bn_ctx[0]
        if (isinstance(by, ops.LazyFirstY) and by.t is None and d.cin == 64
                and ops.conv2d_variant(dg) == "conv_c64_persist_kernel<64>"):
            # the consumer of conv1_1's activation (conv1_2): conv1_1's y is recomputed in the epilogue, not read
            ops.conv2d_bnred_first(dg, dy, w_dg, x.grad, partial, (by.x4, by.w_first) + tuple(x.bn_ctx[1:]))
        else:
            ops.conv2d_bnred(dg, dy, w_dg, x.grad, partial, x.bn_ctx)
        x.bn_partial = (partial, T)
    else:
        ops.conv2d(dg, dy, w_dg, x.grad, None, None)


def max_pool2d(g, x, k, stride, scope="pool", bias_relu=None):
    """slim.max_pool2d(padding='SAME') as a standalone op (pool5 3x3/1, ResNet pool1, subsample).
    bias_relu: x = relu(conv + bias) and this is that layer's own pool (`bias_relu` = its bias variable): when the pool
    is the first to contribute to x's gradient, the ReLU mask and the bias gradient are taken on the POOLED tensors
    (the window maximum is positive iff the element the gradient is routed to is) and the routed gradient IS dz — the
    layer's full-resolution bias_relu_bwd pass disappears (PixelLink conv1_2 / conv2_2: 3 GiB and 1.5 GiB per step)."""
    if g.precision == "f32":
        from . import layers_f32
        return layers_f32.max_pool2d(g, x, k, stride, scope)
    n, h, w, c = x.shape
    oh, pt = ops.same_pad(h, k, stride)
    ow, pl = ops.same_pad(w, k, stride)
    y = g.empty((n, oh, ow, c))
    # index of the first maximum per window (TF gradient routing), kept for the backward pass
    argmax = g.empty((n, oh, ow, c), torch.uint8) if x.requires_grad else None
    fwd = getattr(x, "bn_fwd", None)
    if x.deferred is not None and fwd is not None:
        # x = relu(bn(y)) that nobody has computed yet (resnet_layers.root_block): the pool evaluates it per window
        # element from the raw conv output and x stays unwritten unless somebody else asks for x.data
        by, bsc, bsh, brelu = fwd
        ops.bn_relu_maxpool(by, bsc, bsh, brelu, k, stride, (pt, pl), y, argmax)
    else:
        ops.maxpool(x.data, k, stride, (pt, pl), y, argmax)
    out = Act(y, name=scope)

    def backward():
        if out.grad is None or not x.requires_grad:
            return
        if bias_relu is not None and x.grad is None and argmax is not None and FUSE_BIAS_RELU:
            dzp = g.empty(out.shape)
            ops.bias_relu_bwd(out.data, out.grad, True, dzp, bias_relu.grad, g.workspace())
            x.grad = g.empty(x.shape)
            ops.maxpool_bwd(x._data, dzp, k, stride, (pt, pl), x.grad, False, argmax=argmax, in_shape=x.shape)
            x.bias_done = True
            out.grad = None
            return
        if x.takes_pool_grad and x.grad is None and x.pool_grad is None and argmax is not None:
            # the producer gathers this gradient while it loads (resnet_layers.root_block): nothing is written here
            x.pool_grad = (out.grad, argmax, k, stride, (pt, pl))
            out.grad = None
            return
        acc = x.grad is not None
        if not acc:
            x.grad = g.empty(x.shape)
        # (with the forward's index tensor the backward never reads x: do not force a deferred x into existence)
        ops.maxpool_bwd(x._data if argmax is not None else x.data, out.grad, k, stride, (pt, pl), x.grad, acc,
                        argmax=argmax, in_shape=x.shape)
        out.grad = None
    g.record(backward)
    return out


def prep_images(g, images, means=(123.68, 116.78, 103.94), div=1.0):
    """mean_image_subtraction (nets/model.py:18-31) + f16 cast into the [n,h,w,4] layout; `div`
    folds the PixelLink pipeline's `(x - 120) / 60` into the same pass."""
    if images.shape[-1] != len(means):
        raise ValueError("len(means) must match the number of channels")
    if g.precision == "f32":
        from . import layers_f32
        if div != 1.0:
            raise NotImplementedError("input normalisation is folded into the f16 path only")
        return layers_f32.prep_images(g, images, means)
    n, h, w, _ = images.shape
    x4 = g.empty((n, h, w, 4))
    ops.prep_images(images, x4, means, div)
    return Act(x4, requires_grad=False, name="images")


# ---------------------------------------------------------------------------- fuse heads
class SmallAct:
    """f32 [n,h,w,C] head tensor."""
    __slots__ = ("data", "grad")

    def __init__(self, data):
        self.data = data
        self.grad = None


def head_conv_bn(g, feat, names, couts, is_training=True, relu=True):
    """The BN'd 1x1 fuse convs that several heads apply to ONE feature map, evaluated in a single
    pass over the feature: slim.conv2d(feat, c, 1) for c in couts (nets/model_vgg_16.py:160-172).
    Returns (z, scale, shift, ctx): z = raw conv output [n,h,w,sum(couts)] f32."""
    n, h, w, cin = feat.shape
    C = sum(couts)
    ws, ws_small = g.workspace(), g.ws_small
    # one merged parameter per source; columns map onto the reference variables `names`
    with g.variable_scope("+".join(names)):
        wv = g.get_variable("weights", (cin, C), _merged_init(g, cin, couts), regularized=True)
        gamma, beta, mm, mv = _bn_vars(g, C)

    def mk(old):
        if old is None:
            old = (g.empty((32, cin)), g.empty((cin, 32)))
        ops.pack_weights_small(wv.data, old[0], old[1])
        return old
    w_kc32, w_ck32 = g.packed(wv, "small", mk)
    z = g.empty((n, h, w, C), F32)
    if g.precision == "f32":
        from . import layers_f32
        layers_f32.head_conv(g, feat, wv, C, z)
    else:
        ops.conv1x1_small(feat.data, w_kc32, C, z)
    P = n * h * w
    scale, shift = g.empty((C,), F32), g.empty((C,), F32)
    mean, invstd = g.empty((C,), F32), g.empty((C,), F32)
    if is_training:
        T = ops.sc_num_partials(P, C)
        part, stage = ws_small.two(T * 2 * C * 4, ops.bn_reduce_workspace(T, C))
        ops.sc_stats(z, C, part)
        ops.bn_finalize(part, T, C, float(P), gamma.data, beta.data, BN_EPS, BN_DECAY, mm.data, mv.data,
                        scale, shift, mean, invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, scale, shift)
    out = SmallAct(z)   # grad of out = gradient w.r.t. relu(bn(z))

    def backward():
        if out.grad is None:
            return
        dz = g.empty(z.shape, F32)
        ops.sc_bn_bwd(z, scale, shift, mean, invstd, out.grad, relu, gamma.grad, beta.grad, dz, ws)
        ops.conv1x1_small_wgrad(feat.data, dz, C, wv.grad, ws)
        if feat.requires_grad:
            acc = feat.grad is not None
            if not acc:
                feat.grad = g.empty(feat.shape)
            ops.conv1x1_small_dgrad(dz, w_ck32, C, feat.grad, acc)
        out.grad = None
    g.record(backward, (wv, gamma, beta))
    return out, scale, shift


def head_conv_bias(g, feat, names, couts, initializer=None):
    """PixelLinkNet's fuse convs on one feature map — slim.conv2d(feat, c, [1,1], activation_fn=None)
    with biases, no normaliser (nets/pixellink.py:58-67) — merged into one pass over the feature.
    Returns (z, None, None) in the triple form `fuse` consumes."""
    n, h, w, cin = feat.shape
    C = sum(couts)
    ws = g.workspace()
    with g.variable_scope("+".join(names)):
        init = _merged_init(g, cin, couts, initializer or xavier_uniform)
        wv = g.get_variable("weights", (cin, C), init, regularized=True)
        bias = g.get_variable("biases", (C,), constant(0.0))

    def mk(old):
        if old is None:
            old = (g.empty((32, cin)), g.empty((cin, 32)))
        ops.pack_weights_small(wv.data, old[0], old[1])
        return old
    w_kc32, w_ck32 = g.packed(wv, "small", mk)
    z = g.empty((n, h, w, C), F32)
    if g.precision == "f32":
        from . import layers_f32
        layers_f32.head_conv(g, feat, wv, C, z, bias)
    else:
        ops.conv1x1_small(feat.data, w_kc32, C, z, bias.data)
    out = SmallAct(z)

    def backward():
        if out.grad is None:
            return
        ops.sc_colsum(out.grad, C, bias.grad, ws)
        ops.conv1x1_small_wgrad(feat.data, out.grad, C, wv.grad, ws)
        if feat.requires_grad:
            acc = feat.grad is not None
            if not acc:
                feat.grad = g.empty(feat.shape)
            ops.conv1x1_small_dgrad(out.grad, w_ck32, C, feat.grad, acc)
        out.grad = None
    g.record(backward, (wv, bias))
    return out, None, None


BATCH_HEADS = __import__("os").environ.get("OCR_BATCH_HEADS", "1") == "1"    # one launch per kernel kind over the head sources


def head_group(g, feats, names_list, couts, *, mode="bn", is_training=True, relu=True, initializer=None):
    """The fuse convs of ALL head sources (nets/model_vgg_16.py:160-172: fc7, conv5_3, conv4_3, conv3_3; nets/pixellink.py:
    58-67; nets/model.py:129-141) with one launch per kernel kind over the sources instead of one per source: per feature map
    the same merged 1x1 convolution (+ batch-norm statistics | + biases) as head_conv_bn / head_conv_bias, same variables in
    the same order.  Returns the list of (z SmallAct, scale, shift) triples `fuse` consumes (scale = shift = None for
    mode="bias").  OCR_BATCH_HEADS=0 (or the f32 verification precision) runs the per-source forms."""
    if not BATCH_HEADS or g.precision == "f32" or len(feats) > 4:
        out = []
        for f, nm in zip(feats, names_list):
            out.append(head_conv_bn(g, f, nm, couts, is_training=is_training, relu=relu) if mode == "bn" else
                       head_conv_bias(g, f, nm, couts, initializer=initializer))
        return out
    C = sum(couts)
    ws = g.workspace()
    srcs = []
    for feat, names in zip(feats, names_list):
        n, h, w, cin = feat.shape
        with g.variable_scope("+".join(names)):
            if mode == "bn":
                wv = g.get_variable("weights", (cin, C), _merged_init(g, cin, couts), regularized=True)
                gamma, beta, mm, mv = _bn_vars(g, C)
                bias = None
            else:
                wv = g.get_variable("weights", (cin, C), _merged_init(g, cin, couts, initializer or xavier_uniform),
                                    regularized=True)
                bias = g.get_variable("biases", (C,), constant(0.0))
                gamma = beta = mm = mv = None

        def mk(old, wv=wv, cin=cin):
            if old is None:
                old = (g.empty((32, cin)), g.empty((cin, 32)))
            ops.pack_weights_small(wv.data, old[0], old[1])
            return old
        w_kc32, w_ck32 = g.packed(wv, "small", mk)
        srcs.append(dict(feat=feat, P=n * h * w, cin=cin, wv=wv, bias=bias, gamma=gamma, beta=beta, mm=mm, mv=mv,
                         w_kc32=w_kc32, w_ck32=w_ck32, z=g.empty((n, h, w, C), F32)))
    train_bn = mode == "bn" and is_training
    for s in srcs:
        if mode == "bn":
            s["scale"], s["shift"] = g.empty((C,), F32), g.empty((C,), F32)
            s["mean"], s["invstd"] = g.empty((C,), F32), g.empty((C,), F32)
        if train_bn:
            s["T"] = ops.conv1x1_small_batch_rows(s["P"])
            s["part"] = g.empty((s["T"], 2, C), F32)
    ops.conv1x1_small_batch([(s["feat"].data, s["w_kc32"], s["bias"].data if s["bias"] is not None else None, s["z"],
                              s.get("part")) for s in srcs])
    if train_bn:
        ops.bn_finalize_batch([(s["part"], s["T"], C, float(s["P"]), s["gamma"].data, s["beta"].data, s["mm"].data,
                                s["mv"].data, s["scale"], s["shift"], s["mean"], s["invstd"]) for s in srcs],
                              BN_EPS, BN_DECAY)
    elif mode == "bn":
        for s in srcs:
            ops.bn_inference_params(s["gamma"].data, s["beta"].data, s["mm"].data, s["mv"].data, BN_EPS, s["scale"], s["shift"])
    for s in srcs:
        s["out"] = SmallAct(s["z"])          # grad of out = gradient w.r.t. relu(bn(z)) (bn) / z (bias)

    def backward():
        live = [s for s in srcs if s["out"].grad is not None]
        if not live:
            return
        if mode == "bn":
            if not is_training:
                raise NotImplementedError("backward through inference-mode batch norm")
            items = []
            for s in live:
                s["dz"] = g.empty(s["z"].shape, F32)
                T = ops.sc_num_partials(s["P"], C)
                items.append((s["z"], s["scale"], s["shift"], s["mean"], s["invstd"], s["out"].grad, s["gamma"].grad,
                              s["beta"].grad, s["dz"], g.empty((T, 2, C), F32), relu))
            ops.sc_bn_bwd_batch(items)
        else:
            items = []
            for s in live:
                s["dz"] = s["out"].grad
                T = ops.sc_num_partials(s["P"], C)
                items.append((s["dz"], s["bias"].grad, g.empty((T + 1, 2, C), F32)))
            ops.sc_colsum_batch(items)
        # the batched MFMA kernel: 128-channel blocks, 32-bit buffer offsets (heads.hip: P * cin * 2 and P * C * 4 below
        # 2 GiB); anything else — a 256-channel map beyond 4.19 M pixels, say — keeps the per-map kernel, which has no limit
        def is_wide(s):
            return s["cin"] % 128 == 0 and s["P"] * s["cin"] * 2 < (1 << 31) and s["P"] * C * 4 < (1 << 31)
        wide = [s for s in live if is_wide(s)]
        if wide:
            ops.conv1x1_small_wgrad_batch([
                (s["feat"].data, s["dz"], s["wv"].grad,
                 g.empty((ops.conv1x1_small_wgrad_batch_slab_bytes(s["P"], s["cin"]),), torch.uint8)) for s in wide])
        for s in live:
            if not is_wide(s):
                ops.conv1x1_small_wgrad(s["feat"].data, s["dz"], C, s["wv"].grad, ws)
        items = []
        for s in live:
            feat = s["feat"]
            if feat.requires_grad:
                acc = feat.grad is not None
                if not acc:
                    feat.grad = g.empty(feat.shape)
                items.append((s["dz"], s["w_ck32"], feat.grad, acc))
        if items:
            ops.conv1x1_small_dgrad_batch(items)
        for s in live:
            s["out"].grad = None
            s["dz"] = None
    produces = []
    for s in srcs:
        produces += [s["wv"]] + ([s["gamma"], s["beta"]] if mode == "bn" else [s["bias"]])
    g.record(backward, produces)
    return [(s["out"], s.get("scale") if mode == "bn" else None, s.get("shift") if mode == "bn" else None) for s in srcs]


def pointwise_pair(g, x, scopes, *, mode="bn", is_training=True, relu=True, initializer=None):
    """The two predication convolutions on the fused 18-channel head tensor — pixel (2 -> 2 on channels 0..1) and link
    (16 -> 16 on channels 2..17): `pointwise_bn` x 2 (nets/model_vgg_16.py:166,173: conv + BN + ReLU) or `pointwise_bias`
    x 2 (nets/pixellink.py:61,67, nets/model.py:139-141: conv + biases, linear) — as ONE pass over the tensor per
    direction, same variables in the same order.  Returns (pixel SmallAct [n,h,w,2], link SmallAct [n,h,w,16])."""
    n, h, w, C = x.data.shape
    if not BATCH_HEADS or g.precision == "f32" or C != 18:
        if mode == "bn":
            return (pointwise_bn(g, x, 0, 2, scopes[0], is_training=is_training, relu=relu),
                    pointwise_bn(g, x, 2, 16, scopes[1], is_training=is_training, relu=relu))
        return (pointwise_bias(g, x, 0, 2, scopes[0], initializer=initializer),
                pointwise_bias(g, x, 2, 16, scopes[1], initializer=initializer))
    P = n * h * w
    ws = g.workspace()
    hd = []
    for scope, c in zip(scopes, (2, 16)):
        with g.variable_scope(scope):
            if mode == "bn":
                wv = g.get_variable("weights", (1, 1, c, c), variance_scaling(g.rng), regularized=True)
                gamma, beta, mm, mv = _bn_vars(g, c)
                bias = None
            else:
                wv = g.get_variable("weights", (1, 1, c, c), (initializer or xavier_uniform)(g.rng), regularized=True)
                bias = g.get_variable("biases", (c,), constant(0.0))
                gamma = beta = mm = mv = None
        hd.append(dict(c=c, wv=wv, bias=bias, gamma=gamma, beta=beta, mm=mm, mv=mv, z=g.empty((n, h, w, c), F32)))
    a, b = hd
    train_bn = mode == "bn" and is_training
    if train_bn:
        T = ops.sc_pointwise_pair_num_partials(P)
        for d in hd:
            d["part"] = g.empty((T, 2, d["c"]), F32)
    ops.sc_pointwise_pair_fwd(x.data, a["wv"].data, a["bias"].data if a["bias"] is not None else None, b["wv"].data,
                              b["bias"].data if b["bias"] is not None else None, a["z"], b["z"], a.get("part"), b.get("part"))
    if mode == "bn":
        for d in hd:
            d["scale"], d["shift"] = g.empty((d["c"],), F32), g.empty((d["c"],), F32)
            d["mean"], d["invstd"] = g.empty((d["c"],), F32), g.empty((d["c"],), F32)
        if train_bn:
            ops.bn_finalize_batch([(d["part"], T, d["c"], float(P), d["gamma"].data, d["beta"].data, d["mm"].data,
                                    d["mv"].data, d["scale"], d["shift"], d["mean"], d["invstd"]) for d in hd],
                                  BN_EPS, BN_DECAY)
        else:
            for d in hd:
                ops.bn_inference_params(d["gamma"].data, d["beta"].data, d["mm"].data, d["mv"].data, BN_EPS, d["scale"], d["shift"])
        for d in hd:
            d["out"] = SmallAct(g.empty((n, h, w, d["c"]), F32))
        ops.sc_act_batch([(d["z"], d["scale"], d["shift"], d["out"].data) for d in hd], relu)
    else:
        for d in hd:
            d["out"] = SmallAct(d["z"])

    def backward():
        if a["out"].grad is None and b["out"].grad is None:
            return
        for d in hd:
            if d["out"].grad is None:                       # a head the loss did not read: its gradient is zero
                d["out"].grad = g.empty(d["out"].data.shape, F32)
                ops.fill_(d["out"].grad, 0.0)
        if mode == "bn":
            if not is_training:
                raise NotImplementedError("backward through inference-mode batch norm")
            items = []
            for d in hd:
                d["dz"] = g.empty(d["z"].shape, F32)
                Tb = ops.sc_num_partials(P, d["c"])
                items.append((d["z"], d["scale"], d["shift"], d["mean"], d["invstd"], d["out"].grad, d["gamma"].grad,
                              d["beta"].grad, d["dz"], g.empty((Tb, 2, d["c"]), F32), relu))
            ops.sc_bn_bwd_batch(items)
        else:
            for d in hd:
                d["dz"] = d["out"].grad
        if x.grad is not None:
            raise RuntimeError("the fused head tensor already carries a gradient (pointwise_pair writes all of it)")
        x.grad = g.empty(x.data.shape, F32)
        ops.sc_pointwise_pair_bwd(x.data, a["dz"], b["dz"], a["wv"].data, b["wv"].data, x.grad, a["wv"].grad,
                                  a["bias"].grad if a["bias"] is not None else None, b["wv"].grad,
                                  b["bias"].grad if b["bias"] is not None else None, ws)
        for d in hd:
            d["out"].grad = None
            d["dz"] = None
    produces = []
    for d in hd:
        produces += [d["wv"]] + ([d["gamma"], d["beta"]] if mode == "bn" else [d["bias"]])
    g.record(backward, produces)
    return a["out"], b["out"]


def pointwise_bias(g, x, xo, c, scope, initializer=None):
    """text_predication / link_predication: 1x1 conv with biases and no activation on a channel
    slice of the fused head tensor (nets/pixellink.py:61,67).  Returns contiguous logits."""
    n, h, w, C = x.data.shape
    ws = g.workspace()
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (1, 1, c, c), (initializer or xavier_uniform)(g.rng), regularized=True)
        bias = g.get_variable("biases", (c,), constant(0.0))
    out = SmallAct(g.empty((n, h, w, c), F32))
    ops.sc_pointwise_fwd(x.data, xo, c, wv.data, out.data, 0, c, bias.data)

    def backward():
        if out.grad is None:
            return
        ops.sc_pointwise_wgrad(x.data, xo, c, out.grad, 0, c, wv.grad, bias.grad, ws)
        if x.grad is None:
            x.grad = g.empty(x.data.shape, F32)
            ops.fill_(x.grad, 0.0)
        ops.sc_pointwise_dgrad(out.grad, 0, c, wv.data, x.grad, xo, c)
        out.grad = None
    g.record(backward, (wv, bias))
    return out


def _merged_init(g, cin, couts, base=None):
    base = base or variance_scaling

    def init(shape):
        import numpy as np
        cols = [base(g.rng)((1, 1, cin, c)).reshape(cin, c) for c in couts]
        return np.concatenate(cols, axis=1)
    return init


def fuse(g, shape, a=None, b=None, prev=None, relu=True):
    """out = relu(bn(a.z)) + relu(bn(b.z)) + unpool(prev)   (any subset) on f32 head tensors.
    a, b: (SmallAct z, scale, shift) triples from head_conv_bn; prev: SmallAct at half resolution."""
    out = SmallAct(g.empty(shape, F32))
    za, sa, ha = (a[0].data, a[1], a[2]) if a is not None else (None, None, None)
    zb, sb, hb = (b[0].data, b[1], b[2]) if b is not None else (None, None, None)
    ops.sc_fuse(out.data, za, sa, ha, zb, sb, hb, prev.data if prev is not None else None, relu)

    def backward():
        if out.grad is None:
            return
        # the add fans the same gradient out to every branch; the BN/ReLU part of a and b is
        # differentiated inside head_conv_bn's closure
        for br in (a, b):
            if br is not None:
                if br[0].grad is not None:
                    raise RuntimeError("head tensor used twice")
                br[0].grad = out.grad
        if prev is not None:
            dprev = g.empty(prev.data.shape, F32)
            ops.sc_unpool_bwd(out.grad, dprev)
            prev.grad = dprev
        out.grad = None
    g.record(backward)
    return out


def pointwise_bn(g, x, xo, c, scope, is_training=True, relu=True):
    """Final predication conv on a channel slice of a head tensor:
    out = relu(bn(conv1x1(x[..., xo:xo+c], c)))   (pixel_cls / link_cls,
    nets/model_vgg_16.py:166,173).  Returns a contiguous SmallAct [n,h,w,c]."""
    n, h, w, C = x.data.shape
    P = n * h * w
    ws = g.workspace()
    with g.variable_scope(scope):
        wv = g.get_variable("weights", (1, 1, c, c), variance_scaling(g.rng), regularized=True)
        gamma, beta, mm, mv = _bn_vars(g, c)
    z = g.empty((n, h, w, c), F32)
    ops.sc_pointwise_fwd(x.data, xo, c, wv.data, z, 0, c)
    scale, shift = g.empty((c,), F32), g.empty((c,), F32)
    mean, invstd = g.empty((c,), F32), g.empty((c,), F32)
    if is_training:
        T = ops.sc_num_partials(P, c)
        part, stage = g.ws_small.two(T * 2 * c * 4, ops.bn_reduce_workspace(T, c))
        ops.sc_stats(z, c, part)
        ops.bn_finalize(part, T, c, float(P), gamma.data, beta.data, BN_EPS, BN_DECAY, mm.data, mv.data,
                        scale, shift, mean, invstd, stage)
    else:
        ops.bn_inference_params(gamma.data, beta.data, mm.data, mv.data, BN_EPS, scale, shift)
    out = SmallAct(g.empty((n, h, w, c), F32))
    ops.sc_fuse(out.data, z, scale, shift, relu=relu)

    def backward():
        if out.grad is None:
            return
        dz = g.empty(z.shape, F32)
        ops.sc_bn_bwd(z, scale, shift, mean, invstd, out.grad, relu, gamma.grad, beta.grad, dz, ws)
        ops.sc_pointwise_wgrad(x.data, xo, c, dz, 0, c, wv.grad, None, ws)
        if x.grad is None:
            x.grad = g.empty(x.data.shape, F32)
            ops.fill_(x.grad, 0.0)
        ops.sc_pointwise_dgrad(dz, 0, c, wv.data, x.grad, xo, c)
        out.grad = None
    g.record(backward, (wv, gamma, beta))
    return out
