"""GPU parity of single layers (forward + backward through the C ABI) vs the CPU oracle on
identical f16-representable inputs.  These are the well-conditioned checks: one layer deep, so the
only differences are f32 summation order and one f16 rounding of the result."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu
# bars below are stated for f16 storage; the bfloat16 build (OCR_STORAGE=bf16, run by test_gpu_bf16.py)
# rounds 8x coarser at every storage point
TOL = 8.0 if O.STORAGE == torch.bfloat16 else 1.0


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-20))


def _h(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(O.STORAGE).float().numpy()      # round to the 16-bit storage type


@pytest.mark.parametrize("n,h,w,cin,cout,k,rate,pool", [
    (2, 16, 40, 64, 64, 3, 1, 0),
    (1, 24, 32, 64, 128, 3, 1, 2),
    (2, 9, 11, 128, 64, 3, 1, 2),      # odd sizes: ragged tiles and SAME pooling edge
    (1, 12, 12, 64, 128, 3, 6, 0),     # fc6-style dilation
    (2, 8, 8, 128, 128, 1, 1, 0),      # fc7-style 1x1
])
def test_conv_bn_relu_pool(device, n, h, w, cin, cout, k, rate, pool):
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Act, Graph
    rng = np.random.default_rng(1)
    x = _h(rng.standard_normal((n, h, w, cin)))
    wt = _h(rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin)))
    gamma = (1 + 0.1 * rng.standard_normal(cout)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    oh, ow = (h + 1) // 2 if pool else h, (w + 1) // 2 if pool else w
    gout = _h(rng.standard_normal((n, oh, ow, cout)) * 0.1)
    gfull = _h(rng.standard_normal((n, h, w, cout)) * 0.1) if pool else None

    # ---- device
    g = Graph(device, loss_scale=1.0)
    xa = Act(torch.from_numpy(x).to(O.STORAGE).to(device))
    full, pooled = layers.conv2d(g, xa, cout, k, "L", rate=rate, pool=pool)
    g.reset_tape()
    g.store.load_state_dict({"L/weights": wt, "L/BatchNorm/gamma": gamma, "L/BatchNorm/beta": beta,
                             "L/BatchNorm/moving_mean": np.zeros(cout, np.float32),
                             "L/BatchNorm/moving_variance": np.ones(cout, np.float32)})
    full, pooled = layers.conv2d(g, xa, cout, k, "L", rate=rate, pool=pool)
    out_d = (pooled if pool else full).data.float().cpu().numpy()
    if pool:
        pooled.grad = torch.from_numpy(gout).to(O.STORAGE).to(device)
        full.grad = torch.from_numpy(gfull).to(O.STORAGE).to(device)
    else:
        full.grad = torch.from_numpy(gout).to(O.STORAGE).to(device)
    g.backward()
    torch.cuda.synchronize()
    dv = g.store.vars
    d_dw = dv["L/weights"].grad.cpu().numpy()
    d_dg = dv["L/BatchNorm/gamma"].grad.cpu().numpy()
    d_db = dv["L/BatchNorm/beta"].grad.cpu().numpy()
    d_dx = xa.grad.float().cpu().numpy()
    d_mm = dv["L/BatchNorm/moving_mean"].data.cpu().numpy()
    d_mv = dv["L/BatchNorm/moving_variance"].data.cpu().numpy()

    # ---- oracle (mixed: identical storage roundings)
    p = {"L/weights": wt, "L/BatchNorm/gamma": gamma, "L/BatchNorm/beta": beta,
         "L/BatchNorm/moving_mean": np.zeros(cout, np.float32),
         "L/BatchNorm/moving_variance": np.ones(cout, np.float32)}
    tp = O.to_torch_params(p)
    xt = torch.from_numpy(x).requires_grad_(True)
    upd = {}
    a = O._conv_block(xt, tp, "L", rate, "bn", True, upd)
    if pool:
        o = O.max_pool(a, 2, 2)
        (o * torch.from_numpy(gout)).sum().backward(retain_graph=True)
        (a * torch.from_numpy(gfull)).sum().backward()
    else:
        o = a
        (o * torch.from_numpy(gout)).sum().backward()
    o_np = o.detach().numpy()
    assert np.abs(out_d - o_np).max() <= 2e-3 * TOL * max(1.0, np.abs(o_np).max())
    assert _rel(d_mm, upd["L/BatchNorm/moving_mean"].numpy()) < 1e-4
    assert np.abs(d_mv - upd["L/BatchNorm/moving_variance"].numpy()).max() < 1e-5
    assert _rel(d_db, tp["L/BatchNorm/beta"].grad.numpy()) < 5e-3 * TOL
    assert _rel(d_dg, tp["L/BatchNorm/gamma"].grad.numpy()) < 5e-3 * TOL
    assert _rel(d_dw, tp["L/weights"].grad.numpy()) < 1e-2 * TOL
    assert _rel(d_dx, xt.grad.numpy()) < 1e-2 * TOL


def test_first_conv(device):
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Graph
    rng = np.random.default_rng(2)
    n, h, w, cout = 2, 20, 45, 64
    img = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    wt = _h(rng.standard_normal((3, 3, 3, cout)) * 0.05)
    gout = _h(rng.standard_normal((n, h, w, cout)) * 0.1)
    g = Graph(device, loss_scale=1.0)
    x4 = layers.prep_images(g, torch.from_numpy(img).to(device))
    layers.conv2d(g, x4, cout, 3, "c", first=True)
    g.reset_tape()
    g.store.load_state_dict({"c/weights": wt})
    full, _ = layers.conv2d(g, x4, cout, 3, "c", first=True)
    full.grad = torch.from_numpy(gout).to(O.STORAGE).to(device)
    g.backward()
    torch.cuda.synchronize()
    p = {"c/weights": wt, "c/BatchNorm/gamma": np.ones(cout, np.float32), "c/BatchNorm/beta": np.zeros(cout, np.float32),
         "c/BatchNorm/moving_mean": np.zeros(cout, np.float32), "c/BatchNorm/moving_variance": np.ones(cout, np.float32)}
    tp = O.to_torch_params(p)
    xm = O.q(O.mean_image_subtraction(torch.from_numpy(img)), True)
    a = O._conv_block(xm, tp, "c", 1, "bn", True, {})
    (a * torch.from_numpy(gout)).sum().backward()
    assert np.abs(full.data.float().cpu().numpy() - a.detach().numpy()).max() < 4e-3 * TOL
    assert _rel(g.store.vars["c/weights"].grad.cpu().numpy(), tp["c/weights"].grad.numpy()) < 1e-2 * TOL
    assert _rel(g.store.vars["c/BatchNorm/gamma"].grad.cpu().numpy(), tp["c/BatchNorm/gamma"].grad.numpy()) < 5e-3 * TOL


def test_maxpool3x3s1(device):
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Act, Graph
    rng = np.random.default_rng(3)
    x = _h(np.round(rng.standard_normal((2, 7, 9, 16)) * 2) / 2)     # many ties
    gout = _h(rng.standard_normal((2, 7, 9, 16)))
    g = Graph(device, loss_scale=1.0)
    xa = Act(torch.from_numpy(x).to(O.STORAGE).to(device))
    y = layers.max_pool2d(g, xa, 3, 1)
    y.grad = torch.from_numpy(gout).to(O.STORAGE).to(device)
    g.backward()
    xt = torch.from_numpy(x).requires_grad_(True)
    yo = O.max_pool(xt, 3, 1)
    (yo * torch.from_numpy(gout)).sum().backward()
    assert np.array_equal(y.data.float().cpu().numpy(), yo.detach().numpy())
    assert np.abs(xa.grad.float().cpu().numpy() - xt.grad.numpy()).max() < 2e-2 * TOL


def test_heads_and_dice(device):
    """fuse heads (1x1 small convs + BN + ReLU + unpool/add pyramid + predication convs) and dice."""
    from tensorflow_ocr_amd import checkpoint, layers
    from tensorflow_ocr_amd.graph import Act, Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    rng = np.random.default_rng(4)
    n, h = 2, 8       # fc7..conv4_3 take the MFMA weight-gradient route, conv3_3 (cin 32) the VALU one
    chans = {"fc7": 128, "conv5_3": 64, "conv4_3": 64, "conv3_3": 32}
    sizes = {"fc7": h, "conv5_3": h, "conv4_3": 2 * h, "conv3_3": 4 * h}
    feats = {k: _h(np.abs(rng.standard_normal((n, sizes[k], sizes[k], c)))) for k, c in chans.items()}
    order = ["fc7", "conv5_3", "conv4_3", "conv3_3"]
    p = {}
    for base, cout in ((0, 2), (5, 16)):
        for i, key in enumerate(order):
            nm = "feature_fusion/Conv" + ("_%d" % (base + i) if base + i else "")
            p[nm + "/weights"] = _h(rng.standard_normal((1, 1, chans[key], cout)) * np.sqrt(2.0 / chans[key]))
            O._bn_init(p, nm, cout)
            p[nm + "/BatchNorm/gamma"] = (1 + 0.1 * rng.standard_normal(cout)).astype(np.float32)
            p[nm + "/BatchNorm/beta"] = (0.2 * rng.standard_normal(cout)).astype(np.float32)
        nm = "feature_fusion/Conv_%d" % (base + 4)
        p[nm + "/weights"] = (rng.standard_normal((1, 1, cout, cout)) * np.sqrt(2.0 / cout)).astype(np.float32)
        O._bn_init(p, nm, cout)
        p[nm + "/BatchNorm/beta"] = (0.3 + 0.2 * rng.standard_normal(cout)).astype(np.float32)
    q4 = 4 * h
    pixel = (rng.uniform(size=(n, q4, q4, 1)) < 0.3).astype(np.float32)
    link = (rng.uniform(size=(n, q4, q4, 8)) < 0.3).astype(np.float32)
    mask = (rng.uniform(size=(n, q4, q4, 1)) < 0.9).astype(np.float32)

    def build(g, acts):
        with g.variable_scope('feature_fusion'):
            srcs = [('fc7', ('Conv', 'Conv_5')), ('conv5_3', ('Conv_1', 'Conv_6')),
                    ('conv4_3', ('Conv_2', 'Conv_7')), ('conv3_3', ('Conv_3', 'Conv_8'))]
            hd = {k: layers.head_conv_bn(g, acts[k], nm, (2, 16)) for k, nm in srcs}
            s1 = layers.fuse(g, (n, h, h, 18), a=hd['fc7'], b=hd['conv5_3'])
            s2 = layers.fuse(g, (n, 2 * h, 2 * h, 18), a=hd['conv4_3'], prev=s1)
            s3 = layers.fuse(g, (n, 4 * h, 4 * h, 18), a=hd['conv3_3'], prev=s2)
            return layers.pointwise_bn(g, s3, 0, 2, 'Conv_4'), layers.pointwise_bn(g, s3, 2, 16, 'Conv_9')

    g = Graph(device, loss_scale=64.0)
    acts = {k: Act(torch.from_numpy(v).to(O.STORAGE).to(device)) for k, v in feats.items()}
    build(g, acts)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    px, lk = build(g, acts)
    L = M.loss(pixel, px, link, lk, mask, graph=g)
    g.backward()
    torch.cuda.synchronize()

    tp = O.to_torch_params(p)
    ft = {k: torch.from_numpy(v).requires_grad_(True) for k, v in feats.items()}
    outs = []
    upd = {}
    for base, c in ((0, 2), (5, 16)):
        def nm(i):
            return "feature_fusion/Conv" + ("_%d" % (base + i) if base + i else "")
        s1 = O._head(ft["fc7"], tp, nm(0), True, True, upd) + O._head(ft["conv5_3"], tp, nm(1), True, True, upd)
        s2 = O.resize_bilinear_x2(s1) + O._head(ft["conv4_3"], tp, nm(2), True, True, upd)
        s3 = O.resize_bilinear_x2(s2) + O._head(ft["conv3_3"], tp, nm(3), True, True, upd)
        outs.append(O._head_f32(s3, tp, nm(4), True, upd))
    Lo = O.dice_loss(torch.from_numpy(pixel), outs[0], torch.from_numpy(link), outs[1], torch.from_numpy(mask))
    Lo.backward()
    assert np.abs(px.data.cpu().numpy() - outs[0].detach().numpy()).max() < 1e-4
    assert np.abs(lk.data.cpu().numpy() - outs[1].detach().numpy()).max() < 1e-4
    assert abs(L.item() - float(Lo)) < 1e-5
    gr = checkpoint.internal_to_tf({nm_: (v.grad / 64.0).cpu().numpy() for nm_, v in g.store.vars.items() if v.trainable})
    for k in sorted(gr):
        r = _rel(gr[k], tp[k].grad.numpy())
        print("%-45s %.3e" % (k, r))
        assert r < 2e-3 * TOL, k
    for k in order:
        r = _rel(acts[k].grad.float().cpu().numpy() / 64.0, ft[k].grad.numpy())
        print("dfeat %-10s %.3e" % (k, r))
        assert r < 1e-2 * TOL


def test_storage_dtype_matches_library(device):
    """The loaded library, the host's tensor dtype and the oracle's rounding agree on the 16-bit
    storage type (f16 by default; bf16 in the child run of tests/test_gpu_bf16.py)."""
    import os
    from tensorflow_ocr_amd import _lib
    from tensorflow_ocr_amd.graph import F16
    want = os.environ.get("OCR_STORAGE", "f16")
    lib = _lib.load()
    assert lib.ocr_storage_dtype().decode() == want == _lib.STORAGE
    assert F16 == O.STORAGE == (torch.bfloat16 if want == "bf16" else torch.float16)
    assert os.path.basename(_lib.LIB_PATH) == ("libocr_hip_bf16.so" if want == "bf16" else "libocr_hip.so")


@pytest.mark.parametrize("n,h,w,c,relu", [(2, 16, 24, 64, True), (1, 15, 9, 128, True), (3, 8, 8, 32, False)])
def test_pooled_bn_backward_with_stored_argmax_equals_recomputing_path(device, n, h, w, c, relu):
    """ocr_bn_relu_pool_idx_f16 / ocr_bn_relu_pool_bwd_idx_f16 (first-max position stored by the forward)
    give bit-identical activations, BN gradients and dy to the recomputing path (pool=2, da_full=NULL),
    including odd sizes whose edge windows are partial."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    rng = np.random.default_rng(c + h)
    y = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(F16).to(device)
    y[0, :2, :2, :8] = 0.5                               # ties: the FIRST maximum must win
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).to(device)
    shift = torch.from_numpy(rng.normal(0, 0.3, c).astype(np.float32)).to(device)
    mean = torch.from_numpy(rng.normal(0, 0.2, c).astype(np.float32)).to(device)
    invstd = torch.from_numpy(rng.uniform(0.7, 1.3, c).astype(np.float32)).to(device)
    oh, ow = (h + 1) // 2, (w + 1) // 2
    da = torch.from_numpy(rng.standard_normal((n, oh, ow, c)).astype(np.float32)).to(F16).to(device)
    ws = ops.Workspace(device, 8 << 20)
    p1 = torch.empty((n, oh, ow, c), dtype=F16, device=device)
    p2 = torch.empty_like(p1)
    am = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=device)
    ops.bn_relu(y, scale, shift, relu, 2, None, p1)
    ops.bn_relu_pool_idx(y, scale, shift, relu, None, p2, am)
    assert torch.equal(p1, p2) and int(am.max()) <= 7 and int((am[0, 0, 0, :8] & 3).max()) == 0
    assert torch.equal((am & 4) != 0, p2.float() > 0)
    outs = []
    for which in (0, 1):
        dg = torch.zeros(c, device=device)
        db = torch.zeros(c, device=device)
        dy = torch.empty_like(y)
        if which == 0:
            ops.bn_relu_bwd(y, scale, shift, mean, invstd, None, da, relu, 2, dg, db, dy, ws)
        else:
            ops.bn_relu_pool_bwd_idx(y, scale, mean, invstd, p2, am, da, relu, dg, db, dy, ws)
        torch.cuda.synchronize()
        outs.append((dg.clone(), db.clone(), dy.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])
    assert float(outs[0][0].abs().sum()) > 0


@pytest.mark.parametrize("n,h,w", [(2, 40, 72), (1, 21, 45)])
def test_first_conv_recomputed_instead_of_read_is_bit_identical(device, n, h, w):
    """conv1_1's second pass (ocr_conv2d_first_bn_relu_f16: the activation from the convolution evaluated AGAIN) and its
    weight gradient with y recomputed from the image (ocr_conv2d_first_wgrad_bn_f16 with w_first): the same MFMA
    sequence on the same operands, so bit-identical to the passes that read the stored y; ragged tile edges included."""
    from tensorflow_ocr_amd import layers, ops
    from tensorflow_ocr_amd.graph import Graph, F32
    rng = np.random.default_rng(h)
    cout = 64
    g = Graph(device, loss_scale=1.0)
    img = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    x4 = layers.prep_images(g, torch.from_numpy(img).to(device)).data
    wt = torch.from_numpy((rng.standard_normal((3, 3, 3, cout)) * 0.05).astype(np.float32)).to(device)
    wf = torch.empty((3, cout, 16), dtype=O.STORAGE, device=device)
    ops.pack_weights_first(wt, wf)
    y = torch.empty((n, h, w, cout), dtype=O.STORAGE, device=device)
    ops.conv2d_first(x4, wf, y)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(device)
    scale[3] = -0.7
    shift = torch.from_numpy(rng.normal(0, 0.3, cout).astype(np.float32)).to(device)
    for relu in (True, False):
        a_ref, a_new = torch.empty_like(y), torch.empty_like(y)
        ops.bn_relu(y, scale, shift, relu, 0, a_ref, None)
        ops.conv2d_first_bn_relu(x4, wf, scale, shift, relu, a_new)
        assert torch.equal(a_ref, a_new), int((a_ref != a_new).sum())
    da = torch.from_numpy((rng.standard_normal((n, h, w, cout)) * 0.1).astype(np.float32)).to(O.STORAGE).to(device)
    coef = tuple(torch.from_numpy(rng.normal(0, 0.5, cout).astype(np.float32)).to(device) for _ in range(3))
    coef = (scale, coef[1], coef[2])                      # A = the layer's scale (the mask reads it)
    ws = ops.Workspace(device, 16 << 20)
    dw0 = torch.zeros((3, 3, 3, cout), dtype=F32, device=device)
    dw1 = torch.zeros_like(dw0)
    ops.conv2d_first_wgrad_bn(x4, da, y, shift, coef, True, dw0, ws)
    ops.conv2d_first_wgrad_bn(x4, da, None, shift, coef, True, dw1, ws, w_first=wf)
    torch.cuda.synchronize()
    assert float(dw0.abs().sum()) > 0 and torch.equal(dw0, dw1), float((dw0 - dw1).abs().max())


@pytest.mark.parametrize("cb", [64, 128])
def test_first_conv_y_not_stored_and_the_fallback_that_evaluates_it(device, cb):
    """conv1_1's y is never stored (ops.LazyFirstY): with a 64-channel consumer the persistent kernel's epilogue recomputes
    it (nobody calls .tensor()); a 128-channel consumer runs another kernel family, which cannot — conv2d_bnred asks the
    LazyFirstY for the tensor, conv1_1's weight gradient then reads that tensor.  Either way the gradients equal those of
    the run that stores y (bit for bit: recomputed y is the stored y)."""
    from tensorflow_ocr_amd import layers, ops
    from tensorflow_ocr_amd.graph import Graph
    rng = np.random.default_rng(31 + cb)
    n, h, w = 2, 100, 130
    img = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    gout = _h(rng.standard_normal((n, h, w, cb)) * 0.1)
    seen = {}

    def run(drop):
        old = layers.FIRST_DROP_Y, layers.FIRST_MOMENTS
        layers.FIRST_DROP_Y = drop
        layers.FIRST_MOMENTS = False           # (this test is about the recomputed y: both runs take the evaluating statistics pass)
        try:
            g = Graph(device, loss_scale=1.0, seed=9)
            x4 = layers.prep_images(g, torch.from_numpy(img).to(device))
            a, _ = layers.conv2d(g, x4, 64, 3, "a", first=True)
            lazy = a.bn_ctx[0]
            assert isinstance(lazy, ops.LazyFirstY) == drop
            b, _ = layers.conv2d(g, a, cb, 3, "b")
            b.grad = torch.from_numpy(gout).to(O.STORAGE).to(device)
            g.backward()
            torch.cuda.synchronize()
            if drop:
                seen["materialised"] = lazy.t is not None
            return {k: v.grad.cpu().numpy().copy() for k, v in g.store.vars.items() if v.trainable}
        finally:
            layers.FIRST_DROP_Y, layers.FIRST_MOMENTS = old
    gd, gs = run(True), run(False)
    assert seen["materialised"] == (cb != 64)
    for k in gs:
        assert np.abs(gs[k]).max() > 0 and np.array_equal(gd[k], gs[k]), (k, np.abs(gd[k] - gs[k]).max())


def test_first_conv_moments_on_uncentred_correlated_images_and_zero_sum_filters(device):
    """the moments form cancels large terms when sum(w) ~ 0 meets strongly correlated patches with a DC
    component (w^T M w is then the small difference of sums ~|w|^2 * 255^2 * pixels, accumulated in f32 over a workgroup's
    pixels before the f64 stages).  Worst case built on purpose: images in 0..255 with NO mean subtracted, smooth (a plane
    + a low-frequency wave + 2 grey levels of noise), every filter made zero-sum per input channel, 8 x 512 x 512 (2 048
    pixels per f32 accumulation; the headline's 32 images make it 8 192: the f32 error grows with its square root, x 2).
    Against float64 on the same 16-bit operands.  Measured on MI355X (cancellation |w|^2 x^2 / var = 3.9e3): variance within
    7.6e-6 relative, mean within 4e-9 of a standard deviation — the evaluating pass, which sums 16-bit ROUNDINGS of y, is at
    1.0e-5 / 1.1e-6 on the same data.  Bar: 1e-4 / 1e-5 (csrc/conv_first.hip states this bound)."""
    from tensorflow_ocr_amd import layers, ops
    from tensorflow_ocr_amd.graph import F16, Graph
    n, h, w = 8, 512, 512
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 60.0 + 120.0 * (xx / w) + 40.0 * np.sin(yy / 37.0) * np.cos(xx / 53.0)
    img = np.clip(base[None, :, :, None] + rng.normal(0, 2.0, (n, h, w, 3)) + rng.uniform(0, 30, (n, 1, 1, 3)), 0, 255).astype(np.float32)
    g = Graph(device, loss_scale=1.0, seed=4)
    x4 = layers.prep_images(g, torch.from_numpy(img).to(device), means=(0.0, 0.0, 0.0))
    wt = rng.standard_normal((3, 3, 3, 64)) * 0.05
    wt -= wt.mean((0, 1), keepdims=True)                       # zero-sum over the nine taps of every (cin, cout) pair
    wt = _h(wt)
    wdev = torch.from_numpy(wt).to(device)
    wf = torch.empty((3, 64, 16), dtype=F16, device=device)
    ops.pack_weights_first(wdev, wf)
    ws = ops.Workspace(device, 16 << 20)
    row = torch.zeros((1, 2, 64), dtype=torch.float32, device=device)
    ops.conv2d_first_moments(x4.data, wf, row, 64, ws)
    mt = ops.conv2d_first_num_mtiles(n, h, w)
    part = torch.zeros((mt, 2, 64), dtype=torch.float32, device=device)
    from tensorflow_ocr_amd._lib import CONV_STATS
    ops.conv2d_first(x4.data, wf, None, CONV_STATS, None, part, cout=64)
    torch.cuda.synchronize()
    x = x4.data[..., :3].double().cpu().numpy()
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    s1, s2 = np.zeros(64), np.zeros(64)
    for b in range(n):                                         # (image by image: 8 x 512^2 x 64 doubles at once is 1 GiB)
        y = np.zeros((h, w, 64))
        for ky in range(3):
            for kx in range(3):
                y += xp[b, ky:ky + h, kx:kx + w, :] @ wt[ky, kx].astype(np.float64)
        s1 += y.sum((0, 1))
        s2 += (y * y).sum((0, 1))
    N = n * h * w
    mean, var = s1 / N, s2 / N - (s1 / N) ** 2
    got = row.double().cpu().numpy()[0]
    gmean, gvar = got[0] / N, got[1] / N - (got[0] / N) ** 2
    ev = part.double().sum(0).cpu().numpy()
    emean, evar = ev[0] / N, ev[1] / N - (ev[0] / N) ** 2
    e_mean, e_var = np.abs(gmean - mean).max() / np.sqrt(var).min(), np.abs(gvar / var - 1).max()
    print("moments vs float64: mean %.2e sigma, variance %.2e relative | evaluating pass: mean %.2e sigma, variance %.2e | "
          "cancellation |w|^2 x^2 / var ~ %.1e" % (e_mean, e_var, np.abs(emean - mean).max() / np.sqrt(var).min(),
                                                  np.abs(evar / var - 1).max(), float((wt ** 2).sum((0, 1, 2)).max() * (x ** 2).mean() / var.min())))
    assert e_mean <= 1e-5 and e_var <= 1e-4


@pytest.mark.parametrize("n,h,w", [(2, 100, 130), (1, 64, 96), (3, 37, 45)])
def test_first_conv_statistics_from_the_image_moments(device, n, h, w):
    """conv1_1's batch-norm statistics without evaluating the convolution (ocr_conv2d_first_moments_f16): sum y_c = w_c . m,
    sum y_c^2 = w_c^T M w_c from the 28 x 28 second moments of the image patches.  Against float64 on the same 16-bit
    operands (ragged tiles, image borders: zero padding enters the patches), against the evaluating pass (which sums the
    16-bit ROUNDINGS of y: equal to ~1e-4 of a standard deviation / 1e-6 relative in the variance), and through the
    layer: activations and gradients within the layer test's bars of the run that evaluates."""
    from tensorflow_ocr_amd import layers, ops
    from tensorflow_ocr_amd.graph import F16, Graph
    rng = np.random.default_rng(n * 100 + h)
    img = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    g = Graph(device, loss_scale=1.0, seed=4)
    x4 = layers.prep_images(g, torch.from_numpy(img).to(device))
    wt = _h(rng.standard_normal((3, 3, 3, 64)) * 0.02)
    wdev = torch.from_numpy(wt).to(device)
    wf = torch.empty((3, 64, 16), dtype=F16, device=device)
    ops.pack_weights_first(wdev, wf)
    ws = ops.Workspace(device, 16 << 20)
    row = torch.zeros((1, 2, 64), dtype=torch.float32, device=device)
    ops.conv2d_first_moments(x4.data, wf, row, 64, ws)
    mt = ops.conv2d_first_num_mtiles(n, h, w)
    part = torch.zeros((mt, 2, 64), dtype=torch.float32, device=device)
    from tensorflow_ocr_amd._lib import CONV_STATS
    ops.conv2d_first(x4.data, wf, None, CONV_STATS, None, part, cout=64)
    torch.cuda.synchronize()
    # float64 convolution of the 16-bit operands
    x = x4.data[..., :3].double().cpu().numpy()
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    y = np.zeros((n, h, w, 64))
    for ky in range(3):
        for kx in range(3):
            y += xp[:, ky:ky + h, kx:kx + w, :] @ wt[ky, kx].astype(np.float64)
    s1, s2 = y.sum((0, 1, 2)), (y * y).sum((0, 1, 2))
    got = row.double().cpu().numpy()[0]
    N = n * h * w
    mean, var = s1 / N, s2 / N - (s1 / N) ** 2
    gmean, gvar = got[0] / N, got[1] / N - (got[0] / N) ** 2
    assert np.abs(gmean - mean).max() <= 1e-5 * np.sqrt(var).max() and np.abs(gvar / var - 1).max() <= 1e-5
    ev = part.double().sum(0).cpu().numpy()
    emean, evar = ev[0] / N, ev[1] / N - (ev[0] / N) ** 2
    tol = 16.0 if O.STORAGE == torch.bfloat16 else 1.0
    assert np.abs(gmean - emean).max() <= 1e-3 * tol * np.sqrt(var).max() and np.abs(gvar / evar - 1).max() <= 1e-4 * tol * tol
    # through the layer
    gout = _h(rng.standard_normal((n, h, w, 64)) * 0.1)

    def run(moments):
        old = layers.FIRST_MOMENTS
        layers.FIRST_MOMENTS = moments
        try:
            gg = Graph(device, loss_scale=1.0, seed=9)
            xx = layers.prep_images(gg, torch.from_numpy(img).to(device))
            a, _ = layers.conv2d(gg, xx, 64, 3, "a", first=True)
            b, _ = layers.conv2d(gg, a, 64, 3, "b")
            b.grad = torch.from_numpy(gout).to(O.STORAGE).to(device)
            gg.backward()
            torch.cuda.synchronize()
            return a.data.float().cpu().numpy(), {k: v.grad.cpu().numpy().copy() for k, v in gg.store.vars.items() if v.trainable}
        finally:
            layers.FIRST_MOMENTS = old
    a1, g1 = run(True)
    a0, g0 = run(False)
    assert np.abs(a1 - a0).max() <= 2e-3 * TOL * max(1.0, np.abs(a0).max())
    # (two 16-bit evaluations of one graph: a 1e-7 change of the statistics flips the last bit of some activations and with
    # it a few ReLU decisions of layer b — sparse differences of a percent of the tensor maximum)
    for k in g0:
        a_, b_ = g1[k].ravel().astype(np.float64), g0[k].ravel().astype(np.float64)
        assert np.linalg.norm(a_ - b_) <= 2e-2 * TOL * np.linalg.norm(b_), k
        assert a_ @ b_ / (np.linalg.norm(a_) * np.linalg.norm(b_) + 1e-30) >= (0.99 if O.STORAGE == torch.bfloat16 else 0.9995), k


def test_pooled_bn_layer_backward_sums_from_the_consumer_convolution(device):
    """conv a (+BN+ReLU, 2x2 pool its only reader) -> conv b: b's input-gradient kernel sums a's BN-backward terms over
    the POOLED positions (bn_y = y_pool, the conv output at each window's first maximum) and a's backward is the
    apply-only ocr_bn_relu_pool_bwd_idx_apply_f16.  Same gradients as the path with its own reduction pass over y, up
    to the grouping of the f32 sums; odd sizes: partial edge windows."""
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Graph
    rng = np.random.default_rng(21)
    n, h, w, cin, ca, cb = 2, 30, 66, 64, 64, 128
    x = _h(rng.standard_normal((n, h, w, cin)))
    gout = _h(rng.standard_normal((n, (h + 1) // 2, (w + 1) // 2, cb)) * 0.1)

    def run(fused):
        old = layers.FUSE_BN_POOL_REDUCE
        layers.FUSE_BN_POOL_REDUCE = fused
        try:
            g = Graph(device, loss_scale=1.0, seed=5)
            from tensorflow_ocr_amd.graph import Act, F16
            xa = Act(torch.from_numpy(x).to(F16).to(device), name="x")
            xa.requires_grad = True
            _, pa = layers.conv2d(g, xa, ca, 3, "a", pool=2, keep_full=False)
            assert (pa.bn_ctx is not None) == fused
            fb, _ = layers.conv2d(g, pa, cb, 3, "b")
            fb.grad = torch.from_numpy(gout).to(F16).to(device)
            g.backward()
            torch.cuda.synchronize()
            names = ("a/weights", "a/BatchNorm/gamma", "a/BatchNorm/beta", "b/weights", "b/BatchNorm/gamma")
            return {k: g.store.vars[k].grad.float().cpu().numpy().copy() for k in names}, xa.grad.float().cpu().numpy().copy()
        finally:
            layers.FUSE_BN_POOL_REDUCE = old
    gf, dxf = run(True)
    gu, dxu = run(False)
    for k in gf:
        ref = np.abs(gu[k]).max()
        assert ref > 0 and np.abs(gf[k] - gu[k]).max() <= 2e-3 * TOL * ref, (k, np.abs(gf[k] - gu[k]).max(), ref)
    # (the fused path's apply step is the guest form dy = A*dz + B*y + C, csrc/guest_bn.hip: one 16-bit rounding of dy
    # apart from the general kernel's sc*(dz - k_dz - xhat*k_dzx) — 2^-8 relative in the bfloat16 build)
    assert np.abs(dxf - dxu).max() <= 2e-3 * TOL * np.abs(dxu).max()


def test_first_conv_weight_gradient_with_bn_backward_applied_on_load(device):
    """conv1_1 -> conv1_2 (both conv + BN + ReLU): conv1_2's input-gradient kernel leaves conv1_1's BN-backward sums,
    and conv1_1 — which has no input gradient — computes dy = A*dz + B*y + C inside its weight-gradient kernel
    (ocr_conv2d_first_wgrad_bn_f16) instead of in a separate apply pass.  Against the oracle, and against the
    unfused device path (same numbers up to the grouping of the f32 operations)."""
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Graph
    rng = np.random.default_rng(12)
    n, h, w, cout = 2, 40, 72, 64
    img = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    p = {"a/weights": _h(rng.standard_normal((3, 3, 3, cout)) * 0.05),
         "b/weights": _h(rng.standard_normal((3, 3, cout, cout)) * np.sqrt(2.0 / (9 * cout)))}
    for nm in ("a", "b"):
        O._bn_init(p, nm, cout)
        p[nm + "/BatchNorm/gamma"] = (1 + 0.1 * rng.standard_normal(cout)).astype(np.float32)
        p[nm + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    gout = _h(rng.standard_normal((n, h, w, cout)) * 0.1)

    def run(fused):
        old = layers.FUSE_FIRST_WGRAD, layers.FIRST_MOMENTS
        layers.FUSE_FIRST_WGRAD = fused
        layers.FIRST_MOMENTS = False       # (both runs on the evaluating statistics pass: the test is about the weight gradient)
        try:
            g = Graph(device, loss_scale=1.0)
            x4 = layers.prep_images(g, torch.from_numpy(img).to(device))

            def net():
                a, _ = layers.conv2d(g, x4, cout, 3, "a", first=True)
                b, _ = layers.conv2d(g, a, cout, 3, "b")
                return a, b
            net()
            g.reset_tape()
            g.store.load_state_dict(p)
            a, b = net()
            b.grad = torch.from_numpy(gout).to(O.STORAGE).to(device)
            g.backward()
            torch.cuda.synchronize()
            return {k: v.grad.cpu().numpy().copy() for k, v in g.store.vars.items() if v.trainable}
        finally:
            layers.FUSE_FIRST_WGRAD, layers.FIRST_MOMENTS = old
    gf, gu = run(True), run(False)
    tp = O.to_torch_params(p)
    xm = O.q(O.mean_image_subtraction(torch.from_numpy(img)), True)
    a = O._conv_block(xm, tp, "a", 1, "bn", True, {})
    b = O._conv_block(O.qg(a, True), tp, "b", 1, "bn", True, {})
    (b * torch.from_numpy(gout)).sum().backward()
    for k in ("a/weights", "a/BatchNorm/gamma", "a/BatchNorm/beta"):
        ref = tp[k].grad.numpy()
        assert _rel(gf[k], gu[k]) < 2e-3 * TOL, (k, _rel(gf[k], gu[k]))           # fused vs unfused device paths
        assert _rel(gf[k], ref) < 1e-2 * TOL, (k, _rel(gf[k], ref))               # vs the oracle (bars of test_first_conv)
    assert np.array_equal(gf["a/BatchNorm/gamma"], gu["a/BatchNorm/gamma"])       # same reduction either way
