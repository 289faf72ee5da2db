"""GPU: the product library's f32 inference precision ON THE MATRIX CORES —
`ocr_conv2d_f32_mfma` (v_mfma_f32_32x32x2_f32, csrc/f32_infer.hip) against the independent plain direct convolution of
libocr_verify.so and against float64, over the convolution shapes of the three nets; then whole nets at the north star's
1e-3 at 512^2 and 1024^2 through libocr_hip.so (slim.conv2d: nets/vgg.py:14-39, nets/resnet_v1.py:97-105)."""
import time

import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu
F32 = torch.float32

# (n, h, w, cin, cout, k, stride, rate, flags)   flags: 1 bias, 2 relu, 8 accumulate
SHAPES = [
    (2, 32, 32, 3, 64, 3, 1, 1, 0),          # conv1_1 (cin = 3: the scalar staging path)
    (2, 24, 40, 64, 64, 3, 1, 1, 0),
    (1, 17, 23, 64, 128, 3, 1, 1, 3),        # odd map, bias + ReLU
    (2, 16, 16, 256, 256, 3, 1, 1, 0),
    (1, 16, 16, 512, 1024, 3, 1, 6, 0),      # fc6: dilation 6
    (1, 16, 16, 1024, 1024, 1, 1, 1, 1),     # fc7
    (2, 33, 31, 64, 256, 1, 1, 1, 8),        # 1x1 accumulate (concat-free merge conv)
    (1, 30, 30, 64, 64, 3, 2, 1, 0),         # stride 2 (conv2d_same's SAME form is resolved by the host: plain strided here)
    (1, 20, 20, 3, 64, 7, 2, 1, 0),          # ResNet root 7x7/2
    (1, 9, 9, 128, 18, 1, 1, 1, 1),          # head conv: cout 18 (not a multiple of 4 couts per lane group... scalar stores)
    (1, 8, 8, 130, 66, 3, 1, 1, 2),          # ragged cin / cout
]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv_f32_mfma_vs_direct_and_float64(device, shape):
    from tensorflow_ocr_amd import ops
    n, h, w, cin, cout, k, stride, rate, flags = shape
    rng = np.random.default_rng(sum(shape))
    x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).to(device)
    wt = torch.from_numpy((rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)).to(device)
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(device)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, stride, rate)
    d.flags = flags
    y0 = torch.from_numpy(rng.standard_normal((n, d.oh, d.ow, cout)).astype(np.float32)).to(device)
    ya, yb = y0.clone(), y0.clone()
    ops.conv2d_f32(d, x, wt, ya, bias if flags & 1 else None, route="mfma")
    ops.conv2d_f32(d, x, wt, yb, bias if flags & 1 else None, route="direct")
    torch.cuda.synchronize()
    # float64 reference through torch on the CPU
    xt = x.double().cpu().permute(0, 3, 1, 2)
    wtt = wt.double().cpu().permute(3, 2, 0, 1)
    pad = (d.pad_left, max(0, (d.ow - 1) * stride + (k - 1) * rate + 1 - w - d.pad_left),
           d.pad_top, max(0, (d.oh - 1) * stride + (k - 1) * rate + 1 - h - d.pad_top))
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, pad), wtt, stride=stride, dilation=rate).permute(0, 2, 3, 1)
    if flags & 1:
        ref = ref + bias.double().cpu()
    if flags & 2:
        ref = ref.clamp_min(0)
    if flags & 8:
        ref = ref + y0.double().cpu()
    scale = float(ref.abs().max())
    ea, eb = float((ya.double().cpu() - ref).abs().max()), float((yb.double().cpu() - ref).abs().max())
    print("mfma vs f64 %.2e | direct vs f64 %.2e | mfma vs direct %.2e (scale %.2f)" % (ea, eb, float((ya - yb).abs().max()), scale))
    assert ea <= 4e-6 * scale and eb <= 4e-6 * scale


def _moving(p, rng):
    for k in p:
        if k.endswith('moving_mean'):
            p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
        if k.endswith('moving_variance'):
            p[k] = rng.uniform(0.5, 1.5, p[k].shape).astype(np.float32)
    return p


@pytest.mark.parametrize("size,n", [(512, 1), (1024, 1)])
def test_model_vgg_f32_precision_within_1e3_at_full_size(device, size, n):
    """Inference mode (test.py: is_training=False) on the matrix-core f32 path of libocr_hip.so: logits and softmax score
    maps within 1e-3 of the f32 oracle at the headline and at the decode config's resolution."""
    import os
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    rng = np.random.default_rng(0)
    p = _moving(O.init_model_vgg_params(rng), rng)
    images, _, _, _ = O.synthetic_batch(rng, n, size)
    g = Graph(device, precision="f32")
    M.model_vgg(images[:, :64, :64], is_training=False, graph=g)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    px, lk = M.model_vgg(images, is_training=False, graph=g)
    g.reset_tape()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    px, lk = M.model_vgg(images, is_training=False, graph=g)
    g.reset_tape()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with torch.no_grad():
        fpx, flk, _ = O.model_vgg(torch.from_numpy(images), O.to_torch_params(p, requires_grad=False), False, mixed=False)
    e1 = float((px.data.cpu() - fpx).abs().max())
    e2 = float((lk.data.cpu() - flk).abs().max())
    s = float((torch.softmax(px.data.cpu(), -1) - torch.softmax(fpx, -1)).abs().max())
    print("f32 precision %dx%d n=%d: pixel logits Linf %.2e, link logits Linf %.2e, P(text) Linf %.2e (logit range %.2f); forward %.1f ms" % (
        size, size, n, e1, e2, s, float(fpx.abs().max()), dt * 1e3))
    assert max(e1, e2, s) < 1e-3


def test_pixellink_f32_precision_1024(device):
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import pixellink
    import os
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    rng = np.random.default_rng(2)
    p = O.init_pixellink_params(rng)
    images, _, _, _ = O.synthetic_batch(rng, 1, 1024)
    x = ((images - 120.0) / 60.0).astype(np.float32)
    g = Graph(device, precision="f32")
    pixellink.PixelLinkNet(x[:, :64, :64], graph=g)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    net = pixellink.PixelLinkNet(x, graph=g)
    g.reset_tape()
    with torch.no_grad():
        opx, olk, _ = O.pixellink_net(torch.from_numpy(x), O.to_torch_params(p, requires_grad=False), mixed=False)
    e1 = float((net.pixel_cls.data.cpu() - opx).abs().max())
    e2 = float((net.link_cls.data.cpu() - olk).abs().max())
    e3 = float((net.pixel_scores.cpu() - torch.softmax(opx, -1)).abs().max())
    print("PixelLinkNet f32 precision 1024^2: pixel_cls %.2e link_cls %.2e pixel_scores %.2e (range %.2f)" % (e1, e2, e3, float(olk.abs().max())))
    assert max(e1, e2, e3) < 1e-3


def test_whole_net_direct_route_agrees(device, monkeypatch):
    """The same f32 graph with the convolutions on the plain direct checker kernel (OCR_F32_CONV=direct): the two routes
    agree to f32 summation-order noise end to end."""
    from tensorflow_ocr_amd import checkpoint, ops
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    rng = np.random.default_rng(1)
    p = O.init_model_vgg_params(rng)
    images, _, _, _ = O.synthetic_batch(rng, 2, 64)
    outs = {}
    for route in ("mfma", "direct"):
        monkeypatch.setattr(ops, "F32_CONV", route)
        g = Graph(device, precision="f32")
        M.model_vgg(images, graph=g)
        g.reset_tape()
        g.store.reset_non_trainable()
        g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
        px, lk = M.model_vgg(images, graph=g)
        outs[route] = (px.data.clone(), lk.data.clone())
    assert float((outs["mfma"][0] - outs["direct"][0]).abs().max()) < 1e-4
    assert float((outs["mfma"][1] - outs["direct"][1]).abs().max()) < 1e-4
