"""CPU restatement (oracle) of the BowieHsu/tensorflow_ocr hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker.  The product path (tensorflow_ocr_amd/) never imports it and has no
CPU fallback.

PARITY UNPINNED: the reference is Python-2 / TensorFlow-1.4 graph code that can be neither
imported nor run here (no TF, no cv2) and ships no tests, fixtures or golden
vectors.  This file restates the cited reference lines with the TF-1.4 op semantics listed in
SURVEY.md §3.5 on top of PyTorch-CPU / NumPy arithmetic.  The single known-answer value the
reference holds (example.py:13-21, softmax([1,2]) = [0.268941, 0.731059]) is checked in
tests/test_oracle.py.

Two arithmetic modes:
  * fp32 (default): every tensor f32 — the reference's arithmetic.
  * mixed=True: emulates the device pipeline's storage precision — conv inputs/outputs,
    activations and activation gradients are rounded to the 16-bit storage type (IEEE f16, or
    bfloat16 when the environment says OCR_STORAGE=bf16, the libocr_hip_bf16.so build) at exactly
    the points where the HIP path stores 16-bit tensors (reductions, BN statistics and parameters
    stay f32/f64).
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

torch.set_grad_enabled(True)
STORAGE = torch.bfloat16 if os.environ.get("OCR_STORAGE", "f16") == "bf16" else torch.float16


# ----------------------------------------------------------------------------- rounding
class _Q(torch.autograd.Function):
    """forward: round to f16 storage; backward: straight through."""
    @staticmethod
    def forward(ctx, x):
        return x.to(STORAGE).float()

    @staticmethod
    def backward(ctx, g):
        return g


class _QG(torch.autograd.Function):
    """forward: identity; backward: gradient rounded to f16 storage."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(STORAGE).float()


def q(x, mixed):
    return _Q.apply(x) if mixed else x


def qg(x, mixed):
    return _QG.apply(x) if mixed else x


# ------------------------------------------------------------------------ TF primitives
def tf_same_pad(size, k, stride=1, rate=1):
    """TF 'SAME': pad_total = max((ceil(n/s)-1)*s + k_eff - n, 0); before = total//2 (§3.5-2)."""
    k_eff = (k - 1) * rate + 1
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    return out, total // 2, total - total // 2


def conv2d(x, w_hwio, stride=1, rate=1, padding="SAME"):
    """slim.conv2d's convolution (no bias/activation).  x NHWC, w HWIO (nets/vgg.py:14)."""
    n, h, wd, c = x.shape
    kh, kw, ci, co = w_hwio.shape
    xn = x.permute(0, 3, 1, 2)
    if padding == "SAME":
        _, pt, pb = tf_same_pad(h, kh, stride, rate)
        _, pl, pr = tf_same_pad(wd, kw, stride, rate)
        xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w_hwio.permute(3, 2, 0, 1), stride=stride, dilation=rate)
    return y.permute(0, 2, 3, 1)


def conv2d_same(x, w_hwio, stride, rate=1):
    """resnet_utils.conv2d_same (nets/resnet_utils.py:77-122): explicit pad + VALID for stride>1."""
    if stride == 1:
        return conv2d(x, w_hwio, 1, rate, "SAME")
    k = w_hwio.shape[0]
    k_eff = k + (k - 1) * (rate - 1)
    pad_total = k_eff - 1
    pb = pad_total // 2
    pe = pad_total - pb
    xn = F.pad(x.permute(0, 3, 1, 2), (pb, pe, pb, pe)).permute(0, 2, 3, 1)
    return conv2d(xn, w_hwio, stride, rate, "VALID")


def max_pool(x, k, stride):
    """slim.max_pool2d(padding='SAME'): -inf padding, first maximum wins the gradient."""
    n, h, w, c = x.shape
    oh, pt, pb = tf_same_pad(h, k, stride)
    ow, pl, pr = tf_same_pad(w, k, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf"))
    win = xn.unfold(2, k, stride).unfold(3, k, stride)          # n,c,oh,ow,k,k
    win = win.reshape(n, c, oh, ow, k * k)
    idx = torch.argmax(win.detach(), dim=-1, keepdim=True)      # first max
    return torch.gather(win, -1, idx).squeeze(-1).permute(0, 2, 3, 1)


def resize_bilinear_x2(x):
    """tf.image.resize_bilinear(size=2x, align_corners=False) of TF 1.4: src = dst*0.5, no
    half-pixel centres (nets/model.py:14-15)."""
    n, h, w, c = x.shape

    def idx(sz):
        o = torch.arange(2 * sz)
        lo = o // 2
        hi = torch.clamp(lo + 1, max=sz - 1)
        frac = (o % 2).to(x.dtype) * 0.5
        return lo, hi, frac
    ylo, yhi, yf = idx(h)
    xlo, xhi, xf = idx(w)
    top = x[:, ylo]
    bot = x[:, yhi]
    tl, tr = top[:, :, xlo], top[:, :, xhi]
    bl, br = bot[:, :, xlo], bot[:, :, xhi]
    xf = xf.view(1, 1, -1, 1)
    yf = yf.view(1, -1, 1, 1)
    t = tl + (tr - tl) * xf
    b = bl + (br - bl) * xf
    return t + (b - t) * yf


def batch_norm(x, gamma, beta, moving_mean, moving_var, training, decay=0.997, eps=1e-5):
    """slim.batch_norm (fused): normalise with the biased batch variance; the moving average gets
    the unbiased one.  Returns (y, new_moving_mean, new_moving_var)."""
    c = x.shape[-1]
    flat = x.reshape(-1, c)
    if training:
        m = flat.shape[0]
        mean = flat.double().mean(0)
        var = (flat.double() ** 2).mean(0) - mean ** 2
        var = torch.clamp(var, min=0.0)
        invstd = (1.0 / torch.sqrt(var + eps)).float()
        mean = mean.float()
        nm = moving_mean * decay + mean.detach() * (1 - decay)
        unb = var.detach().float() * (m / max(m - 1, 1))
        nv = moving_var * decay + unb * (1 - decay)
    else:
        mean = moving_mean
        invstd = 1.0 / torch.sqrt(moving_var + eps)
        nm, nv = moving_mean, moving_var
    scale = gamma * invstd
    shift = beta - mean * scale
    return x * scale + shift, nm, nv


def mean_image_subtraction(images, means=(123.68, 116.78, 103.94)):
    """nets/model.py:18-31."""
    if images.shape[-1] != len(means):
        raise ValueError('len(means) must match the number of channels')
    return images - torch.tensor(means, dtype=images.dtype)


def softmax(x):
    """slim.softmax over the last axis (example.py:13-21 known answer)."""
    return torch.softmax(x, dim=-1)


# ------------------------------------------------------------------------------- VGG
VGG_CFG = [("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 3, 256), ("conv4", 3, 512), ("conv5", 3, 512)]


def vgg_layer_names():
    names = []
    for block, reps, cout in VGG_CFG:
        for r in range(1, reps + 1):
            names.append(("%s/%s_%d" % (block, block, r), 3, cout, 1))
    names.append(("fc6", 3, 1024, 6))
    names.append(("fc7", 1, 1024, 1))
    return names


def init_vgg_params(rng, prefix="", normalizer="bn", width_div=1, cin=3):
    """He-normal weights (seeded), BN gamma=1 beta=0 or zero biases; names as slim creates them."""
    p = {}
    c_prev = cin
    for name, k, cout, _ in vgg_layer_names():
        cout = max(cout // width_div, 8)
        std = math.sqrt(2.0 / (k * k * c_prev))
        p[prefix + name + "/weights"] = (rng.standard_normal((k, k, c_prev, cout)) * std).astype(np.float32)
        if normalizer == "bn":
            p[prefix + name + "/BatchNorm/gamma"] = np.ones(cout, np.float32)
            p[prefix + name + "/BatchNorm/beta"] = np.zeros(cout, np.float32)
            p[prefix + name + "/BatchNorm/moving_mean"] = np.zeros(cout, np.float32)
            p[prefix + name + "/BatchNorm/moving_variance"] = np.ones(cout, np.float32)
        else:
            p[prefix + name + "/biases"] = np.zeros(cout, np.float32)
        c_prev = cout
    return p


def _conv_block(x, p, name, rate, normalizer, mixed, updates, bn_training=True):
    w = p[name + "/weights"]
    y = conv2d(q(x, mixed), q(w, mixed), 1, rate)       # f16 operands, f32 accumulate
    if normalizer == "bn":
        y = qg(q(y, mixed), mixed)                       # stored f16; its gradient is stored f16
        y, nm, nv = batch_norm(y, p[name + "/BatchNorm/gamma"], p[name + "/BatchNorm/beta"],
                               p[name + "/BatchNorm/moving_mean"], p[name + "/BatchNorm/moving_variance"],
                               bn_training)
        updates[name + "/BatchNorm/moving_mean"] = nm
        updates[name + "/BatchNorm/moving_variance"] = nv
    else:
        y = y + p[name + "/biases"]
    a = q(torch.relu(y), mixed)
    return a


def vgg_basenet(x, p, prefix="", normalizer="bn", mixed=False, updates=None, taps=None):
    """nets/vgg.py:6-42.  x: mean-subtracted NHWC image.  Returns (net, end_points).
    taps (dict, optional): per conv layer `name -> {"x": input, "a": relu(bn(conv)) output,
    "pool": the 2x2-pooled output where one follows}`, every tensor with its gradient retained —
    what a layer-by-layer check of the device path needs (tests/test_gpu_fullsize_nets.py)."""
    updates = {} if updates is None else updates
    end_points = {}

    def keep(t):
        if taps is not None and t.requires_grad:
            t.retain_grad()
        return t

    def block(net, name, rate, first=False):
        xin = net if first else keep(qg(net, mixed))
        a = keep(_conv_block(xin, p, name, rate, normalizer, mixed, updates))
        if taps is not None:
            taps[name] = {"x": xin, "a": a}
        return a
    net = q(x, mixed)
    for bi, (blk, reps, _) in enumerate(VGG_CFG):
        for r in range(1, reps + 1):
            name = "%s%s/%s_%d" % (prefix, blk, blk, r)
            net = block(net, name, 1, first=(bi == 0 and r == 1))
        end_points["%s_%d" % (blk, reps)] = net
        if bi < 4:
            net = keep(max_pool(net, 2, 2))
            if taps is not None:
                taps[name]["pool"] = net
        else:
            net = max_pool(net, 3, 1)
    net = block(net, prefix + "fc6", 6)
    end_points["fc6"] = net
    net = block(net, prefix + "fc7", 1)
    end_points["fc7"] = net
    return net, end_points


# ----------------------------------------------------------------- model_vgg (BN heads)
def init_model_vgg_params(rng, width_div=1):
    p = init_vgg_params(rng, "", "bn", width_div)
    chans = {"fc7": max(1024 // width_div, 8), "conv5_3": max(512 // width_div, 8),
             "conv4_3": max(512 // width_div, 8), "conv3_3": max(256 // width_div, 8)}
    # slim auto-names inside feature_fusion: pixel Conv..Conv_4, link Conv_5..Conv_9
    order = ["fc7", "conv5_3", "conv4_3", "conv3_3"]
    for base, cout in ((0, 2), (5, 16)):
        for i, key in enumerate(order):
            nm = "feature_fusion/Conv" + ("_%d" % (base + i) if base + i else "")
            cin = chans[key]
            p[nm + "/weights"] = (rng.standard_normal((1, 1, cin, cout)) * math.sqrt(2.0 / cin)).astype(np.float32)
            _bn_init(p, nm, cout)
        nm = "feature_fusion/Conv_%d" % (base + 4)
        p[nm + "/weights"] = (rng.standard_normal((1, 1, cout, cout)) * math.sqrt(2.0 / cout)).astype(np.float32)
        _bn_init(p, nm, cout)
    return p


def _bn_init(p, nm, c):
    p[nm + "/BatchNorm/gamma"] = np.ones(c, np.float32)
    p[nm + "/BatchNorm/beta"] = np.zeros(c, np.float32)
    p[nm + "/BatchNorm/moving_mean"] = np.zeros(c, np.float32)
    p[nm + "/BatchNorm/moving_variance"] = np.ones(c, np.float32)


def _head(feat, p, nm, is_training, mixed, updates):
    """slim.conv2d(feat, c, 1) with BN + ReLU (nets/model_vgg_16.py:160-172)."""
    z = conv2d(q(feat, mixed), q(p[nm + "/weights"], mixed), 1, 1)
    z, nmn, nv = batch_norm(z, p[nm + "/BatchNorm/gamma"], p[nm + "/BatchNorm/beta"],
                            p[nm + "/BatchNorm/moving_mean"], p[nm + "/BatchNorm/moving_variance"],
                            is_training)
    updates[nm + "/BatchNorm/moving_mean"] = nmn
    updates[nm + "/BatchNorm/moving_variance"] = nv
    return torch.relu(z)


def _head_f32(x, p, nm, is_training, updates):
    z = conv2d(x, p[nm + "/weights"], 1, 1)
    z, nmn, nv = batch_norm(z, p[nm + "/BatchNorm/gamma"], p[nm + "/BatchNorm/beta"],
                            p[nm + "/BatchNorm/moving_mean"], p[nm + "/BatchNorm/moving_variance"],
                            is_training)
    updates[nm + "/BatchNorm/moving_mean"] = nmn
    updates[nm + "/BatchNorm/moving_variance"] = nv
    return torch.relu(z)


def model_vgg(images, p, is_training=True, mixed=False, updates=None, taps=None):
    """nets/model_vgg_16.py:138-177.  Returns (pixel_cls, link_cls, end_points)."""
    updates = {} if updates is None else updates
    x = mean_image_subtraction(images)
    _, ep = vgg_basenet(x, p, "", "bn", mixed, updates, taps)
    outs = []
    for base, c in ((0, 2), (5, 16)):
        def nm(i):
            return "feature_fusion/Conv" + ("_%d" % (base + i) if base + i else "")
        f = {k: qg(ep[k], mixed) for k in ("fc7", "conv5_3", "conv4_3", "conv3_3")}
        s1 = _head(f["fc7"], p, nm(0), is_training, mixed, updates) + \
            _head(f["conv5_3"], p, nm(1), is_training, mixed, updates)
        s2 = resize_bilinear_x2(s1) + _head(f["conv4_3"], p, nm(2), is_training, mixed, updates)
        s3 = resize_bilinear_x2(s2) + _head(f["conv3_3"], p, nm(3), is_training, mixed, updates)
        outs.append(_head_f32(s3, p, nm(4), is_training, updates))
    return outs[0], outs[1], ep


# ------------------------------------------------------------------------------ losses
def dice_coefficient(y_true_cls, y_pred_cls, training_mask):
    """nets/model_vgg_16.py:179-193 (== nets/model.py:145-159), TF broadcasting included."""
    eps = 1e-5
    inter = torch.sum(y_true_cls * y_pred_cls * training_mask)
    union = torch.sum(y_true_cls * training_mask) + torch.sum(y_pred_cls * training_mask) + eps
    return 1.0 - (2 * inter / union)


def dice_loss(y_true_pixel, y_pred_pixel, y_true_link, y_pred_link, training_mask):
    """nets/model_vgg_16.py:196-225."""
    cls = dice_coefficient(y_true_pixel, y_pred_pixel, training_mask) * 2
    gts = torch.split(y_true_link, y_true_link.shape[-1] // 8, dim=3)
    prs = torch.split(y_pred_link, y_pred_link.shape[-1] // 8, dim=3)
    link = sum(dice_coefficient(g, pr, training_mask) for g, pr in zip(gts, prs))
    return link + cls


# ----------------------------------------------------------------------- optimiser step
def l2_regularizer_grad(w, scale):
    """slim.l2_regularizer(s)(w) = s*sum(w^2)/2 -> gradient s*w."""
    return scale * w


def exponential_decay(lr, step, decay_steps=5000, rate=0.94, staircase=True):
    """tf.train.exponential_decay (multigpu_train.py:104)."""
    e = step // decay_steps if staircase else step / decay_steps
    return lr * rate ** e


def adam_update(w, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps outside the bias correction."""
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    w = w - lr_t * m / (np.sqrt(v) + eps)
    return w, m, v


def momentum_update(w, g, acc, lr, momentum=0.9):
    """tf.train.MomentumOptimizer(lr, momentum) (train_pixellink.py:243), use_nesterov=False:
    accum = momentum * accum + grad; var -= lr * accum."""
    acc = momentum * acc + g
    w = w - lr * acc
    return w, acc


def pixellink_lr(step, base_lr=0.01):
    """train_pixellink.py:222-237: base_lr * tf.case{step<20k: .1, <40k: .01, <60k: .001, default 1}."""
    return base_lr * (0.1 if step < 20000 else 0.01 if step < 40000 else 0.001 if step < 60000 else 1.0)


def ema_decay(decay, num_updates):
    """tf.train.ExponentialMovingAverage(decay, num_updates): min(d, (1+n)/(10+n))."""
    return min(decay, (1.0 + num_updates) / (10.0 + num_updates))


# ----------------------------------------------------------------------------- helpers
def to_torch_params(p, requires_grad=True):
    out = {}
    for k, v in p.items():
        t = torch.from_numpy(np.array(v, dtype=np.float32))
        trainable = not (k.endswith("moving_mean") or k.endswith("moving_variance"))
        t.requires_grad_(requires_grad and trainable)
        out[k] = t
    return out


def synthetic_batch(rng, n, size, rects=8):
    """Synthetic training batch of SURVEY §8d: images U[0,255), pixel label = union of random
    axis-aligned rectangles at 1/4 resolution, link label = 1 where the neighbour in that direction
    (order of §3.4: left, left_down, left_up, right, right_down, right_up, up, down) shares a
    rectangle (border = 1), mask = 1."""
    q4 = size // 4
    images = rng.uniform(0, 255, size=(n, size, size, 3)).astype(np.float32)
    ids = np.zeros((n, q4, q4), np.int32)
    for b in range(n):
        for k in range(rects):
            hh = int(rng.integers(max(2, q4 // 16), max(3, q4 * 3 // 8)))
            ww = int(rng.integers(max(2, q4 // 16), max(3, q4 * 3 // 8)))
            y0 = int(rng.integers(0, q4 - hh + 1))
            x0 = int(rng.integers(0, q4 - ww + 1))
            ids[b, y0:y0 + hh, x0:x0 + ww] = k + 1
    pixel = (ids > 0).astype(np.float32)[..., None]
    offs = [(-1, 0), (-1, 1), (-1, -1), (1, 0), (1, 1), (1, -1), (0, -1), (0, 1)]   # (dx, dy)
    link = np.zeros((n, q4, q4, 8), np.float32)
    pad = np.pad(ids, ((0, 0), (1, 1), (1, 1)), constant_values=-1)
    for d, (dx, dy) in enumerate(offs):
        nb = pad[:, 1 + dy:1 + dy + q4, 1 + dx:1 + dx + q4]
        same = (nb == ids) | (nb == -1)
        link[..., d] = ((ids > 0) & same).astype(np.float32)
    mask = np.ones((n, q4, q4, 1), np.float32)
    return images, pixel, link, mask


# ------------------------------------------------------- softmax / OHNM / focal losses
def _ce2(logits, labels):
    """tf.nn.sparse_softmax_cross_entropy_with_logits for 2 classes; logits [..., 2], labels int."""
    lse = torch.logsumexp(logits, dim=-1)
    return lse - torch.gather(logits, -1, labels.long().unsqueeze(-1)).squeeze(-1)


def det_exp_f32(x):
    """exp(x) as a fixed sequence of IEEE float32 operations (Cody-Waite reduction, degree-6 Taylor,
    2^n scaling; ~1 ulp).  The device evaluates the SAME sequence (loss_softmax.hip: det_exp), so the
    OHNM mining scores — and the threshold / mask that follow, which are index work — are bit-exact
    on both sides.  TF's own softmax bits are out of reach either way (parity unpinned)."""
    f = np.float32
    x = np.clip(np.asarray(x, dtype=f), f(-80.0), f(80.0))
    n = np.rint(x * f(1.44269504))
    r = x - n * f(0.693145751953125)
    r = r - n * f(1.42860677e-06)
    p = np.full_like(r, f(1.3888889e-03))
    for c in (8.3333338e-03, 4.1666668e-02, 1.6666667e-01, 0.5, 1.0, 1.0):
        p = p * r + f(c)
    return np.ldexp(p, n.astype(np.int32)).astype(f)


def neg_score_f32(l0, l1):
    """softmax(logits)[..., 0] = 1 / (1 + exp(l1 - l0)) in float32 (nets/model.py:216-217)."""
    f = np.float32
    l0, l1 = np.asarray(l0, dtype=f), np.asarray(l1, dtype=f)
    return (f(1.0) / (f(1.0) + det_exp_f32(l1 - l0))).astype(f)


def ohnm_single_image(scores, n_pos, neg_mask, ratio=3):
    """nets/model.py:161-184.  scores: P(neg) per pixel (numpy 1-D), neg_mask bool.
    Returns the selected-negative mask (float).  n_pos > 0 with no negatives raises in TF
    (vals[-1] of an empty top_k); the build returns an empty selection."""
    if n_pos <= 0:
        return np.zeros_like(scores, dtype=np.float32)
    n_neg = int(min(n_pos * ratio, int(neg_mask.sum())))
    if n_neg <= 0:
        return np.zeros_like(scores, dtype=np.float32)
    neg_conf = scores[neg_mask]
    vals = np.sort(-neg_conf)[::-1][:n_neg]          # tf.nn.top_k(-neg_conf, k): descending
    threshold = vals[-1]
    return (neg_mask & (scores <= -threshold)).astype(np.float32)


def ohnm_batch(neg_scores, pos_mask, neg_mask, ratio=3):
    """nets/model.py:186-197 (batch size from the tensor, not the constant 14)."""
    sel = [ohnm_single_image(neg_scores[b], int(pos_mask[b].sum()), neg_mask[b], ratio)
           for b in range(neg_scores.shape[0])]
    return pos_mask.astype(np.float32) + np.stack(sel)


def model_loss_ohnm(y_true_pixel, y_pred_pixel, y_true_link, y_pred_link, training_mask=None):
    """nets/model.py:204-261 (`loss`, the one multigpu_train.py:32 calls).  training_mask is unused
    by the reference.  Returns (total, pixel_term, link_terms[8], selected_mask)."""
    n = y_pred_pixel.shape[0]
    label = y_true_pixel.reshape(n, -1)
    pred = y_pred_pixel.reshape(n, -1, 2)
    pn = pred.detach().numpy()
    scores = neg_score_f32(pn[..., 0], pn[..., 1])
    pos = (label == 1).numpy()
    neg = (label == 0).numpy()
    sel = torch.from_numpy(ohnm_batch(scores, pos, neg))
    n_seg_pos = float(pos.sum())
    ce = _ce2(pred, label)
    cls = (ce * sel).sum() / n_seg_pos if n_seg_pos > 0 else torch.tensor(0.0)
    w_pixel = sel.reshape(-1)
    links = []
    for i in range(8):
        ll = y_true_link[..., i].reshape(-1)
        lp = y_pred_link[..., 2 * i:2 * i + 2].reshape(-1, 2)
        lce = _ce2(lp, ll)
        wp = (ll == 1).float() * w_pixel
        wn = (ll == 0).float() * w_pixel
        links.append((lce * wp).sum() / wp.sum() + (lce * wn).sum() / wn.sum())
    total = sum(links) + 2 * cls
    return total, cls, links, sel


def cal_link_loss(link_gt, link_pred, w_pixel):
    """nets/model_vgg_16.py:227-241 for one direction (torch tensors; differentiable in link_pred)."""
    ll = link_gt.reshape(-1)
    lce = _ce2(link_pred.reshape(-1, 2), ll)
    wp = (ll == 1).float() * w_pixel
    wn = (ll == 0).float() * w_pixel
    return (lce * wp).sum() / wp.sum() + (lce * wn).sum() / wn.sum()


def get_pos_and_neg_masks(labels):
    """nets/model.py:199-201."""
    labels = np.asarray(labels)
    return labels == 1, labels == 0


def ohem_loss(y_true_pixel, y_pred_pixel, y_true_link, y_pred_link, training_mask=None):
    """nets/model_vgg_16.py:243-282 (+ cal_link_loss :227-241): positives-only weights."""
    label = y_true_pixel.reshape(-1)
    pred = y_pred_pixel.reshape(-1, 2)
    w = (label == 1).float()
    l_pixel = (_ce2(pred, label) * w).sum() / w.sum()
    links = []
    for i in range(8):
        ll = y_true_link[..., i].reshape(-1)
        lp = y_pred_link[..., 2 * i:2 * i + 2].reshape(-1, 2)
        lce = _ce2(lp, ll)
        wp = (ll == 1).float() * w
        wn = (ll == 0).float() * w
        links.append((lce * wp).sum() / wp.sum() + (lce * wn).sum() / wn.sum())
    return l_pixel * 2 + sum(links), l_pixel, links


def pixellink_build_loss(pixel_cls, link_cls, pixel_labels, link_labels, focal=None):
    """nets/pixellink.py:88-263: the two LOSSES entries (2*mean pixel CE, link total).
    focal=(alpha, gamma) swaps the link CE for the focal loss (build-defined)."""
    seg_pos = (pixel_labels > 0)
    pixel_cls_loss = _ce2(pixel_cls.reshape(-1, 2), seg_pos.reshape(-1)).mean()
    links = []
    for i in range(8):
        logits = link_cls[..., 2 * i:2 * i + 2].reshape(-1, 2)
        lab = (link_labels[..., i] > 0).reshape(-1)
        if focal is None:
            l = _ce2(logits, lab)
        else:
            alpha, gamma = focal
            logp = torch.log_softmax(logits, dim=-1).gather(-1, lab.long().unsqueeze(-1)).squeeze(-1)
            p = logp.exp()
            a = torch.where(lab, torch.tensor(alpha), torch.tensor(1.0 - alpha))
            l = -a * (1 - p) ** gamma * logp
        pw, nw = lab.float(), (~lab).float()
        pn, nn = pw.sum(), nw.sum()
        pos_l = (l * pw).sum() / pn if pn > 0 else torch.tensor(0.0)
        neg_l = (l * nw).sum() / nn if nn > 0 else torch.tensor(0.0)
        links.append(pos_l + neg_l)
    return pixel_cls_loss * 2, sum(links), links


# ------------------------------------------------------------------------------ decode
LINK_OFFSETS = [(-1, 0), (-1, 1), (-1, -1), (1, 0), (1, 1), (1, -1), (0, -1), (0, 1)]   # (dx, dy), §3.4


def pixel_detect(score_map, geo_map, score_map_thresh=0.8, link_thresh=0.8):
    """tool/pixellink_fn.py:120-154.  score_map [N,h,w,1], geo_map [8,N,h,w,2] -> uint8 [h,w] of batch
    element 0: score > thr and, for every direction, NOT (link[...,1] < link_thresh)."""
    score = np.asarray(score_map)
    geo = np.asarray(geo_map)
    if score.ndim == 4:
        score = score[0, :, :, 0]
        geo = geo[:, 0]
    res = (score > score_map_thresh).astype(np.uint8)
    for i in range(8):
        res[geo[i, :, :, 1] < link_thresh] = 0
    return res


def py27_dict_key_order(keys):
    """Iteration order (`graph.keys()`, test_pixellink_fast.py:171) of a CPython-2.7 dict whose keys are the
    DISTINCT non-negative ints `keys`, inserted in that sequence and never deleted.  The interpreter is absent
    here; this restates its published algorithm (Objects/dictobject.c, 2.7): hash(int) = the int;
    open addressing in a power-of-two table, first slot `hash & mask`, then `i = 5*i + perturb + 1` with
    `perturb = hash` shifted right by 5 AFTER each probe; after an insertion that leaves fill*3 >= size*2 the
    table is rebuilt at the smallest power of two > (4 if used <= 50000 else 2) * used, re-inserting the old
    slots in slot order; keys() walks the slots in index order.  (With a table larger than every key the order
    is ascending; a 192 x 320 map with a few thousand keys sits in a 8192- or 32768-slot table, so keys wrap
    and the order is NOT ascending.)"""
    mask = 7
    table = [-1] * 8
    used = 0

    def place(tab, msk, k):
        i = k & msk
        if tab[i] >= 0:
            perturb = k
            while True:
                i = (i << 2) + i + perturb + 1
                perturb >>= 5
                if tab[i & msk] < 0:
                    i &= msk
                    break
        tab[i] = k

    for k in keys:
        k = int(k)
        place(table, mask, k)
        used += 1
        if used * 3 >= (mask + 1) * 2:
            minused = (2 if used > 50000 else 4) * used
            newsize = 8
            while newsize <= minused:
                newsize <<= 1
            old, table, mask = table, [-1] * newsize, newsize - 1
            for e in old:
                if e >= 0:
                    place(table, mask, e)
    return [e for e in table if e >= 0]


def _link_graph(pixel_score, link_scores, pixel_thresh, link_thresh):
    """Neighbour lists of the INTERIOR segment pixels in the script's insertion sequence: x outer, y inner
    (test_pixellink_fast.py:119-150)."""
    h, w = pixel_score.shape
    seg = pixel_score > pixel_thresh
    graph = {}
    for x in range(1, w - 1):
        for y in range(1, h - 1):
            if seg[y, x]:
                nb = []
                for d, (dx, dy) in enumerate(LINK_OFFSETS):
                    if link_scores[d][y, x] > link_thresh and seg[y + dy, x + dx]:
                        nb.append((y + dy) * w + x + dx)
                graph[y * w + x] = nb
    return graph


def reference_key_order(pixel_score, pixel_thresh=0.8, key_order="py27"):
    """The order in which the script's `for i in graph.keys()` meets the keys of one map."""
    h, w = pixel_score.shape
    seg = pixel_score > pixel_thresh
    inserted = [y * w + x for x in range(1, w - 1) for y in range(1, h - 1) if seg[y, x]]
    if key_order == "ascending":
        return sorted(inserted)
    if key_order != "py27":
        raise ValueError("key_order must be 'py27' or 'ascending'")
    return py27_dict_key_order(inserted)


def link_cc_reference_dfs(pixel_score, link_scores, pixel_thresh=0.8, link_thresh=0.9, min_size=10, key_order="py27"):
    """Literal restatement of test_pixellink_fast.py:110-178 for one map: neighbour graph of the
    INTERIOR pixels, directed-edge DFS, keep components with len > min_size.  Seeds are met in the
    iteration order of the script's Python-2 dict (`py27_dict_key_order`; key_order="ascending" = the
    order rounds 1-3 used instead).  Returns int32 labels [h,w] (gid from 1)."""
    h, w = pixel_score.shape
    graph = _link_graph(pixel_score, link_scores, pixel_thresh, link_thresh)
    group = np.zeros(h * w, np.int32)
    gid = 1
    for key in reference_key_order(pixel_score, pixel_thresh, key_order):
        if group[key] != 0:
            continue
        stack, label, seen = [key], [], set()
        while stack:
            v = stack.pop()
            if v not in seen:
                seen.add(v)
                label.append(v)
                for e in graph.get(v, []):
                    if group[e] == 0:
                        stack.append(e)
        if len(label) > min_size:
            group[label] = gid
            gid += 1
    return group.reshape(h, w)


def link_cc_directed_rounds(pixel_score, link_scores, pixel_thresh=0.8, link_thresh=0.9, min_size=10, key_order="py27"):
    """The schedule `ocr_link_cc_directed` runs (csrc/decode.hip: cc_directed_kernel), restated on the CPU so that
    its equivalence with the literal script (`link_cc_reference_dfs`) is checked without a GPU: inside every
    weakly-connected component, rounds of (seed = the key of smallest RANK in the script's key order that is
    unassigned and not known to fail; R = forward-reachable set through unassigned pixels; |R| > min_size ? assign :
    all of R known to fail), gids in ascending rank of the seeds.  Returns int32 labels [h,w]."""
    h, w = pixel_score.shape
    seg = pixel_score > pixel_thresh
    ulab, _ = link_cc_union(pixel_score, link_scores, pixel_thresh, link_thresh, min_size)
    ulab = ulab.ravel()
    order = reference_key_order(pixel_score, pixel_thresh, key_order)
    edges = {}
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            if seg[y, x] and ulab[y * w + x] > 0:
                edges[y * w + x] = [(y + dy) * w + x + dx for d, (dx, dy) in enumerate(LINK_OFFSETS)
                                    if link_scores[d][y, x] > link_thresh and seg[y + dy, x + dx]]
    state = np.where(ulab > 0, 1, 0)
    dead = np.zeros(h * w, bool)
    group = np.zeros(h * w, np.int64)
    while True:
        seeds = {}
        for k in order:
            if k in edges and state[k] == 1 and not dead[k] and ulab[k] not in seeds:
                seeds[ulab[k]] = k
        if not seeds:
            break
        for c, s0 in seeds.items():
            R, frontier = {s0}, [s0]
            while frontier:
                nxt = []
                for v in frontier:
                    for q in edges.get(v, []):
                        if state[q] == 1 and q not in R:
                            R.add(q)
                            nxt.append(q)
                frontier = nxt
            R = np.fromiter(R, np.int64)
            if len(R) > min_size:
                group[R] = s0 + 1
                state[R] = 2
            else:
                dead[R] = True
    ids = np.zeros(h * w + 1, np.int32)
    nid = 0
    for k in order:                       # gids in the order the script meets its successful seeds
        if group[k] == k + 1:
            nid += 1
            ids[k + 1] = nid
    return ids[group].reshape(h, w).astype(np.int32)


def link_cc_union(pixel_score, link_scores, pixel_thresh=0.8, link_thresh=0.9, min_size=10):
    """Weakly-connected components of the same graph (every directed edge joins its two pixels),
    dense ids in ascending order of each component's smallest pixel index — what the HIP kernel
    computes.  Equals link_cc_reference_dfs whenever the link predictions are symmetric."""
    h, w = pixel_score.shape
    seg = pixel_score > pixel_thresh
    parent = np.arange(h * w)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            if not seg[y, x]:
                continue
            for d, (dx, dy) in enumerate(LINK_OFFSETS):
                if link_scores[d][y, x] > link_thresh and seg[y + dy, x + dx]:
                    a, b = find(y * w + x), find((y + dy) * w + x + dx)
                    if a != b:
                        parent[max(a, b)] = min(a, b)
    roots = np.array([find(i) if seg.flat[i] else -1 for i in range(h * w)])
    labels = np.zeros(h * w, np.int32)
    comps = []
    for r in sorted(set(roots[roots >= 0])):
        members = np.nonzero(roots == r)[0]
        if len(members) > min_size:
            comps.append((int(r), len(members)))
            labels[members] = len(comps)
    return labels.reshape(h, w), comps


def link_cc_union_fast(pixel_score, link_scores, pixel_thresh=0.8, link_thresh=0.9, min_size=10):
    """`link_cc_union` for maps too large for its Python loops (16 x 256^2, BASELINE configs[4]): the
    same edge set built with NumPy, weakly-connected components by scipy.sparse.csgraph, the same
    canonical numbering (ascending smallest pixel index, size filter).  Held equal to link_cc_union
    on small maps by tests/test_oracle.py."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    h, w = pixel_score.shape
    seg = pixel_score > pixel_thresh
    idx = np.arange(h * w).reshape(h, w)
    src, dst = [], []
    inner = np.zeros((h, w), bool)
    inner[1:h - 1, 1:w - 1] = True
    for d, (dx, dy) in enumerate(LINK_OFFSETS):
        nb_seg = np.zeros((h, w), bool)
        nb_seg[1:h - 1, 1:w - 1] = seg[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx]
        e = inner & seg & (link_scores[d] > link_thresh) & nb_seg
        src.append(idx[e])
        dst.append(idx[e] + dy * w + dx)
    src, dst = np.concatenate(src), np.concatenate(dst)
    ncomp, lab = connected_components(coo_matrix((np.ones(len(src), np.int8), (src, dst)), shape=(h * w, h * w)),
                                      directed=True, connection="weak")
    segf = seg.ravel()
    first = np.full(ncomp, h * w, np.int64)
    np.minimum.at(first, lab[segf], np.nonzero(segf)[0])
    size = np.bincount(lab[segf], minlength=ncomp)
    order = [c for c in np.argsort(first, kind="stable") if first[c] < h * w and size[c] > min_size]
    new_id = np.zeros(ncomp, np.int32)
    comps = []
    for c in order:
        comps.append((int(first[c]), int(size[c])))
        new_id[c] = len(comps)
    labels = np.where(segf, new_id[lab], 0).astype(np.int32)
    return labels.reshape(h, w), comps


def synthetic_decode_maps(rng, n, q4, strength=3.0):
    """SURVEY §8d decode inputs: logits = N(0,1) + 2*(label-0.5)*strength from rectangle labels."""
    _, pixel, link, _ = synthetic_batch(rng, n, q4 * 4)
    pl = np.zeros((n, q4, q4, 2), np.float32)
    pl[..., 1] = rng.standard_normal((n, q4, q4)) + 2 * (pixel[..., 0] - 0.5) * strength
    pl[..., 0] = rng.standard_normal((n, q4, q4)) - 2 * (pixel[..., 0] - 0.5) * strength
    ll = np.zeros((n, q4, q4, 16), np.float32)
    for d in range(8):
        ll[..., 2 * d + 1] = rng.standard_normal((n, q4, q4)) + 2 * (link[..., d] - 0.5) * strength
        ll[..., 2 * d] = rng.standard_normal((n, q4, q4)) - 2 * (link[..., d] - 0.5) * strength
    return pl, ll


# -------------------------------------------------------------------- PixelLinkNet (bias VGG)
PL_STAGES = [("fc7", "stage_6"), ("conv5_3", "stage_5"), ("conv4_3", "stage_4"), ("conv3_3", "stage_3")]


def init_pixellink_params(rng, width_div=1):
    """Variables of nets/pixellink.py: `vgg/...` trunk with biases, `pixellink_layers/...` heads."""
    p = init_vgg_params(rng, "vgg/", None, width_div)
    for k in list(p):
        if k.endswith("biases"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    chans = {"fc7": max(1024 // width_div, 8), "conv5_3": max(512 // width_div, 8),
             "conv4_3": max(512 // width_div, 8), "conv3_3": max(256 // width_div, 8)}
    for kind, c in (("pixel", 2), ("link", 16)):
        for key, st in PL_STAGES:
            nm = "pixellink_layers/%s_%s_fuse" % (st, kind)
            p[nm + "/weights"] = (rng.standard_normal((1, 1, chans[key], c)) * math.sqrt(1.0 / chans[key])).astype(np.float32)
            p[nm + "/biases"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
        nm = "pixellink_layers/%s_predication" % ("text" if kind == "pixel" else "link")
        p[nm + "/weights"] = (rng.standard_normal((1, 1, c, c)) * math.sqrt(1.0 / c)).astype(np.float32)
        p[nm + "/biases"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
    return p


def pixellink_net(inputs, p, mixed=False, taps=None):
    """nets/pixellink.py:40-86.  inputs: preprocessed NHWC image.  Returns (pixel_cls, link_cls, end_points).
    taps: as in `vgg_basenet` (per trunk conv: input, output, pooled output, gradients retained)."""
    _, ep = vgg_basenet(inputs, p, "vgg/", None, mixed, taps=taps)
    outs = []
    for kind, c in (("pixel", 2), ("link", 16)):
        def conv(key, st):
            nm = "pixellink_layers/%s_%s_fuse" % (st, kind)
            return conv2d(q(qg(ep[key], mixed), mixed), q(p[nm + "/weights"], mixed), 1, 1) + p[nm + "/biases"]
        s1 = conv("fc7", "stage_6") + conv("conv5_3", "stage_5")
        s2 = resize_bilinear_x2(s1) + conv("conv4_3", "stage_4")
        s3 = resize_bilinear_x2(s2) + conv("conv3_3", "stage_3")
        nm = "pixellink_layers/%s_predication" % ("text" if kind == "pixel" else "link")
        outs.append(conv2d(s3, p[nm + "/weights"], 1, 1) + p[nm + "/biases"])
    return outs[0], outs[1], ep


# ------------------------------------------------------------------------ ResNet-v1-50
RESNET50_BLOCKS = [("block1", [(256, 64, 1)] * 2 + [(256, 64, 2)]),
                   ("block2", [(512, 128, 1)] * 3 + [(512, 128, 2)]),
                   ("block3", [(1024, 256, 1)] * 5 + [(1024, 256, 2)]),
                   ("block4", [(2048, 512, 1)] * 3)]


def _he(rng, shape):
    fan_in = shape[0] * shape[1] * shape[2]
    return (rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32)


def init_resnet50_params(rng, scope="resnet_v1_50", blocks=None):
    blocks = blocks or RESNET50_BLOCKS
    p = {}

    def conv(name, k, cin, cout):
        p[name + "/weights"] = _he(rng, (k, k, cin, cout))
        _bn_init(p, name, cout)
        p[name + "/BatchNorm/gamma"] = (1 + 0.1 * rng.standard_normal(cout)).astype(np.float32)
        p[name + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    conv(scope + "/conv1", 7, 3, 64)
    cin = 64
    for bname, units in blocks:
        for i, (depth, db, stride) in enumerate(units):
            u = "%s/%s/unit_%d/bottleneck_v1" % (scope, bname, i + 1)
            if depth != cin:
                conv(u + "/shortcut", 1, cin, depth)
            conv(u + "/conv1", 1, cin, db)
            conv(u + "/conv2", 3, db, db)
            conv(u + "/conv3", 1, db, depth)
            cin = depth
    return p


def _conv_bn(x, p, name, stride, relu, is_training, mixed, updates, k=None):
    w = p[name + "/weights"]
    y = conv2d_same(q(x, mixed), q(w, mixed), stride)
    y = qg(q(y, mixed), mixed)
    y, nm, nv = batch_norm(y, p[name + "/BatchNorm/gamma"], p[name + "/BatchNorm/beta"],
                           p[name + "/BatchNorm/moving_mean"], p[name + "/BatchNorm/moving_variance"],
                           is_training)
    updates[name + "/BatchNorm/moving_mean"] = nm
    updates[name + "/BatchNorm/moving_variance"] = nv
    return torch.relu(y) if relu else y


def bottleneck(x, p, u, depth, stride, is_training, mixed, updates):
    """nets/resnet_v1.py:68-111."""
    depth_in = x.shape[-1]
    if depth == depth_in:
        shortcut = x if stride == 1 else max_pool(x, 1, stride)
    else:
        shortcut = q(_conv_bn(qg(x, mixed), p, u + "/shortcut", stride, False, is_training, mixed, updates), mixed)
    r = q(_conv_bn(qg(x, mixed), p, u + "/conv1", 1, True, is_training, mixed, updates), mixed)
    r = q(_conv_bn(qg(r, mixed), p, u + "/conv2", stride, True, is_training, mixed, updates), mixed)
    r = q(_conv_bn(qg(r, mixed), p, u + "/conv3", 1, False, is_training, mixed, updates), mixed)
    return q(torch.relu(shortcut + r), mixed)


def resnet_v1_50(x, p, is_training=True, scope="resnet_v1_50", mixed=False, updates=None, blocks=None, taps=None):
    """nets/resnet_v1.py:114-259 (root 7x7/2 + 3x3/2 max-pool, blocks, end points pool2..pool5).
    taps (dict, optional): per bottleneck unit `<scope>/<block>/unit_<i>/bottleneck_v1 -> {"x": input, "out": output}`
    with gradients retained (unit-by-unit checks of the device path, tests/test_gpu_batch_parity.py)."""
    updates = {} if updates is None else updates
    blocks = blocks or RESNET50_BLOCKS
    ep = {}
    net = q(_conv_bn(q(x, mixed), p, scope + "/conv1", 2, True, is_training, mixed, updates), mixed)
    net = max_pool(net, 3, 2)
    ep["pool2"] = net
    for bname, units in blocks:
        for i, (depth, db, stride) in enumerate(units):
            uname = "%s/%s/unit_%d/bottleneck_v1" % (scope, bname, i + 1)
            xin = net
            if taps is not None and xin.requires_grad:
                xin.retain_grad()
            net = bottleneck(xin, p, uname, depth, stride, is_training, mixed, updates)
            if taps is not None:
                if net.requires_grad:
                    net.retain_grad()
                taps[uname] = {"x": xin, "out": net, "depth": depth, "depth_bottleneck": db, "stride": stride}
        ep[scope + "/" + bname] = net
    ep["pool3"], ep["pool4"], ep["pool5"] = ep[scope + "/block1"], ep[scope + "/block2"], net
    return net, ep


def init_model_resnet_params(rng, blocks=None):
    p = init_resnet50_params(rng, blocks=blocks)
    blocks = blocks or RESNET50_BLOCKS
    chans = [blocks[-1][1][-1][0], blocks[1][1][-1][0], blocks[0][1][-1][0], 64]   # pool5, pool4, pool3, pool2
    for base, c in ((0, 2), (4, 16)):
        for i, cin in enumerate(chans):
            nm = "feature_fusion/Conv" + ("_%d" % (base + i) if base + i else "")
            p[nm + "/weights"] = _he(rng, (1, 1, cin, c))
            _bn_init(p, nm, c)
            p[nm + "/BatchNorm/beta"] = (0.2 + 0.1 * rng.standard_normal(c)).astype(np.float32)
    for nm, c in (("feature_fusion/Conv_8", 2), ("feature_fusion/Conv_9", 16)):
        p[nm + "/weights"] = _he(rng, (1, 1, c, c))
        p[nm + "/biases"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
    return p


def model_resnet(images, p, is_training=True, mixed=False, updates=None, blocks=None):
    """nets/model.py:84-143 -> (pixel_4, link_4, end_points)."""
    updates = {} if updates is None else updates
    x = mean_image_subtraction(images)
    _, ep = resnet_v1_50(x, p, is_training, "resnet_v1_50", mixed, updates, blocks)
    f = [ep["pool5"], ep["pool4"], ep["pool3"], ep["pool2"]]
    outs = []
    for base, c, last in ((0, 2, "Conv_8"), (4, 16, "Conv_9")):
        def nm(i):
            return "feature_fusion/Conv" + ("_%d" % (base + i) if base + i else "")
        s = resize_bilinear_x2(_head(qg(f[0], mixed), p, nm(0), is_training, mixed, updates)) + \
            _head(qg(f[1], mixed), p, nm(1), is_training, mixed, updates)
        s = resize_bilinear_x2(s) + _head(qg(f[2], mixed), p, nm(2), is_training, mixed, updates)
        s = resize_bilinear_x2(s) + _head(qg(f[3], mixed), p, nm(3), is_training, mixed, updates)
        outs.append(conv2d(s, p["feature_fusion/%s/weights" % last], 1, 1) + p["feature_fusion/%s/biases" % last])
    return outs[0], outs[1], ep


# --------------------------------------------------------------- EAST merge branch (ResNet)
def init_model_east_params(rng, blocks=None):
    """nets/model_vgg_16.py:85-136 variables: resnet_v1_50/... + feature_fusion/Conv..Conv_8."""
    blocks = blocks or RESNET50_BLOCKS
    p = init_resnet50_params(rng, blocks=blocks)
    f = [blocks[-1][1][-1][0], blocks[1][1][-1][0], blocks[0][1][-1][0], 64]
    outs = [None, 128, 64, 32]
    idx = 0

    def nm():
        nonlocal idx
        s = "feature_fusion/Conv" + ("_%d" % idx if idx else "")
        idx += 1
        return s

    def conv(name, k, cin, cout):
        p[name + "/weights"] = _he(rng, (k, k, cin, cout))
        _bn_init(p, name, cout)
        p[name + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    cg = f[0]
    for i in range(1, 4):
        conv(nm(), 1, cg + f[i], outs[i])
        conv(nm(), 3, outs[i], outs[i])
        cg = outs[i]
    conv(nm(), 3, 32, 32)
    for c in (1, 8):
        n_ = nm()
        p[n_ + "/weights"] = _he(rng, (1, 1, 32, c))
        p[n_ + "/biases"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
    return p


def model_east(images, p, is_training=True, mixed=False, updates=None, blocks=None, taps=None):
    """nets/model_vgg_16.py:85-136 -> (F_score [N,H/4,W/4,1], geo_map [N,H/4,W/4,8], end_points).
    taps: per bottleneck unit, see `resnet_v1_50`."""
    updates = {} if updates is None else updates
    x = mean_image_subtraction(images)
    _, ep = resnet_v1_50(x, p, is_training, "resnet_v1_50", mixed, updates, blocks, taps=taps)
    f = [ep["pool5"], ep["pool4"], ep["pool3"], ep["pool2"]]
    idx = [0]

    def nm():
        s = "feature_fusion/Conv" + ("_%d" % idx[0] if idx[0] else "")
        idx[0] += 1
        return s
    h = f[0]
    g = q(resize_bilinear_x2(qg(h, mixed)), mixed)
    for i in range(1, 4):
        cat = torch.cat([qg(g, mixed), qg(f[i], mixed)], dim=-1)
        c1 = q(_conv_bn(cat, p, nm(), 1, True, is_training, mixed, updates), mixed)
        h = q(_conv_bn(qg(c1, mixed), p, nm(), 1, True, is_training, mixed, updates), mixed)
        if i <= 2:
            g = q(resize_bilinear_x2(qg(h, mixed)), mixed)
        else:
            g = q(_conv_bn(qg(h, mixed), p, nm(), 1, True, is_training, mixed, updates), mixed)
    outs = []
    for _ in range(2):
        n_ = nm()
        z = conv2d(q(qg(g, mixed), mixed), q(p[n_ + "/weights"], mixed), 1, 1) + p[n_ + "/biases"]
        outs.append(torch.sigmoid(z))
    return outs[0], outs[1], ep
