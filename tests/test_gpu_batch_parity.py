"""GPU: parity AT THE BATCH SIZES BASELINE.json quotes, with DISTINCT images.

(a) the three MFMA convolution entry points through the C ABI at n = 32 distinct images on the conv1_2 / conv2_2 /
    conv3_3 / conv4_3 shapes of configs[1] (an addressing error with an even image period — the wrong image of the same
    parity, a split-K slab of image k summed into k+2 — is invisible to a replicated pair; here every image differs);
(b) PixelLinkNet + build_loss (+ focal) + backward at 512^2: n = 2 end to end and layer by layer on the oracle's own layer
    inputs, then n = 32 replicated = n = 2 (configs[2]: bias epilogue, bias_relu_bwd, softmax / OHNM-summary loss);
(c) ResNet-v1-50 EAST at 640^2 in f16: n = 2 unit by unit (every bottleneck on the oracle's input and output gradient,
    the bars of test_gpu_resnet.py) and end to end, then n = 64 replicated = n = 2 (configs[3]'s one-GPU share: flat-tile
    conv_pw over 1.6 M pixels, the 256-cout tail-fusion epilogue, wgrad_pw<64,256,1>);
(d) bench.py's own losses: `bench.py --batch 2 --loss-trace` against the oracle's Adam steps on the same data and
    initial weights.

The oracle runs each of these in seconds to tens of seconds on the GPU box's host cores."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = O.STORAGE == torch.bfloat16
TOL = 8.0 if BF else 1.0


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-20))


def _l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _cos(a, b):
    a, b = np.asarray(a).ravel().astype(np.float64), np.asarray(b).ravel().astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


# ------------------------------------------------------------------------------------------------ (a)
@pytest.mark.parametrize("name,hw,cin,cout", [("conv1_2", 512, 64, 64), ("conv2_2", 256, 128, 128),
                                              ("conv3_3", 128, 256, 256), ("conv4_3", 64, 512, 512)])
def test_conv_abi_batch32_distinct_images(device, name, hw, cin, cout):
    """ocr_conv2d_f16 (forward, input gradient) and ocr_conv2d_wgrad_f16 on 32 DIFFERENT images at the layer's real
    resolution against the oracle's convolution (torch-CPU f32 on the same 16-bit-exact operands).  Per-image errors are
    checked one by one: every image must match ITS OWN reference."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.ops import Workspace
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    n, k = 32, 3
    gen = torch.Generator().manual_seed(hw + cin)
    x = torch.randn((n, hw, hw, cin), generator=gen).to(O.STORAGE)
    dy = (torch.randn((n, hw, hw, cout), generator=gen) * 0.25).to(O.STORAGE)
    wt = (torch.randn((k, k, cin, cout), generator=gen) * float(np.sqrt(2.0 / (k * k * cin)))).to(O.STORAGE).float()
    # image b carries its own signature so that a swapped image cannot pass: a per-image scale on x and dy
    sig = (1.0 + 0.03 * torch.arange(n, dtype=torch.float32)).view(n, 1, 1, 1)
    x = (x.float() * sig).to(O.STORAGE)
    dy = (dy.float() * sig.flip(0)).to(O.STORAGE)
    xt = x.float().requires_grad_(True)
    wtt = wt.clone().requires_grad_(True)
    yo = O.conv2d(xt, wtt, 1, 1)
    yo.backward(dy.float())
    xd, dyd = x.to(device), dy.to(device)
    wm = wt.to(device)
    w_kc = torch.empty((k * k, cout, cin), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((k * k, cin, cout), dtype=O.STORAGE, device=device)
    ops.pack_weights(wm, w_kc, w_ck)
    d = ops.conv_desc((n, hw, hw, cin), cout, k, k, 1, 1)
    d.flags = 0
    y = torch.empty((n, hw, hw, cout), dtype=O.STORAGE, device=device)
    ops.conv2d(d, xd, w_kc, y)
    dg = ops.ConvDesc(n, hw, hw, cout, hw, hw, cin, k, k, 1, 1, k - 1 - d.pad_top, k - 1 - d.pad_left, 1, 0)
    dx = torch.empty((n, hw, hw, cin), dtype=O.STORAGE, device=device)
    ops.conv2d(dg, dyd, w_ck, dx)
    dw = torch.zeros((k, k, cin, cout), dtype=torch.float32, device=device)
    ws = Workspace(device, 256 << 20)
    dd = ops.ConvDesc(n, hw, hw, cin, hw, hw, cout, k, k, 1, 1, d.pad_top, d.pad_left, 0, 0)
    ops.conv2d_wgrad(dd, xd, dyd, dw, ws)
    torch.cuda.synchronize()
    yh, dxh = y.float().cpu(), dx.float().cpu()
    yo_d, dxo = yo.detach(), xt.grad
    tol = 8e-3 if BF else 1e-3
    worst_y = worst_dx = 0.0
    for b in range(n):
        ey = float((yh[b] - yo_d[b]).abs().max() / yo_d[b].abs().max())
        ex = float((dxh[b] - dxo[b]).abs().max() / dxo[b].abs().max())
        worst_y, worst_dx = max(worst_y, ey), max(worst_dx, ex)
        assert ey < tol and ex < tol, (name, "image", b, ey, ex)
    e_dw = _rel(dw.cpu().numpy(), wtt.grad.numpy())
    print("%s n=32 distinct: variant %s | worst per-image y %.2e dx %.2e | dw %.2e" % (
        name, ops.conv2d_variant(d), worst_y, worst_dx, e_dw))
    # f32 accumulation over 32 x hw^2 pixels in different orders on the two sides (split-K slabs vs oneDNN blocking)
    assert e_dw < 2e-5, e_dw                          # measured 1e-6 .. 5.4e-6


# ------------------------------------------------------------------------------------------------ (b)
def _device_pixellink(device, p, x, pixel, link, focal=None, S=256.0):
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import pixellink
    g = Graph(device, loss_scale=S)
    pixellink.PixelLinkNet(x[:1], graph=g)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    net = pixellink.PixelLinkNet(x, graph=g)
    kw = {} if focal is None else {"focal": focal}
    L = net.build_loss(pixel, link, **kw)
    terms = [t.item() for t in g.collections["losses"]]
    g.backward()
    torch.cuda.synchronize()
    grads = checkpoint.internal_to_tf({n: (v.grad / S).cpu().numpy() for n, v in g.store.vars.items() if v.trainable})
    return net.pixel_cls.data.cpu().numpy(), net.link_cls.data.cpu().numpy(), L.item(), terms, grads


@pytest.fixture(scope="module")
def pl512():
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    S = 256.0
    rng = np.random.default_rng(0)
    p = O.init_pixellink_params(rng)
    images, pixel, link, _ = O.synthetic_batch(rng, 2, 512)
    x = ((images - 120.0) / 60.0).astype(np.float32)
    out = {"p": p, "x": x, "pixel": np.ascontiguousarray(pixel[..., 0]), "link": link, "S": S}
    for key, focal in (("plain", None), ("focal", (0.25, 2.0))):
        tp = O.to_torch_params(p)
        taps = {} if focal is None else None
        opx, olk, _ = O.pixellink_net(torch.from_numpy(x), tp, mixed=True, taps=taps)
        p2, ltot, _ = O.pixellink_build_loss(opx, olk, torch.from_numpy(out["pixel"]), torch.from_numpy(link), focal=focal)
        ((p2 + ltot) * S).backward()
        out[key] = dict(tp=tp, taps=taps, px=opx.detach().numpy(), lk=olk.detach().numpy(), l2p=float(p2), link=float(ltot))
    return out


@pytest.mark.parametrize("key", ["plain", "focal"])
def test_pixellinknet_512_end_to_end(device, pl512, key):
    """configs[2] at its real resolution, n = 2: logits, the two loss terms (2 x pixel, link total; train_pixellink.py:263)
    and every weight gradient against the oracle, with and without the focal link weighting."""
    o, r = pl512, pl512[key]
    S = o["S"]
    dpx, dlk, dL, terms, dgr = _device_pixellink(device, o["p"], o["x"], o["pixel"], o["link"],
                                                  focal=None if key == "plain" else (0.25, 2.0), S=S)
    assert dpx.shape == (2, 128, 128, 2) and dlk.shape == (2, 128, 128, 16)
    sc = max(1.0, float(np.abs(r["px"]).max()))
    ogr = {k: (v.grad / S).numpy() for k, v in r["tp"].items() if v.grad is not None}
    cs = sorted((_cos(dgr[k], ogr[k]), k) for k in ogr if ogr[k].size >= 64)
    print("PixelLink 512^2 n=2 (%s): loss terms %.5f / %.5f vs %.5f / %.5f | pixel_cls Linf %.3e link_cls Linf %.3e "
          "(scale %.2f) | lowest gradient cosines %s" % (key, terms[0], terms[1], r["l2p"], r["link"],
                                                          np.abs(dpx - r["px"]).max(), np.abs(dlk - r["lk"]).max(), sc, cs[:3]))
    assert np.abs(dpx - r["px"]).max() < 2e-2 * sc * TOL and np.abs(dlk - r["lk"]).max() < 2e-2 * sc * TOL
    assert abs(terms[0] - r["l2p"]) < 2e-3 * TOL and abs(terms[1] - r["link"]) < 5e-3 * TOL
    assert cs[0][0] > (0.9 if BF else 0.98)


def test_pixellinknet_512_layer_by_layer(device, pl512):
    """Every trunk convolution of PixelLinkNet (bias + ReLU epilogue, bias_relu_bwd; + 2x2 pool where one follows) at
    512-row resolution on the ORACLE's input of that layer and the oracle's gradient of its output."""
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Act, Graph
    o = pl512
    p, tp, taps = o["p"], o["plain"]["tp"], o["plain"]["taps"]
    for name, t in taps.items():
        first = name.endswith("conv1_1")
        pool = 2 if "pool" in t else 0
        rate = 6 if name.endswith("fc6") else 1
        w = p[name + "/weights"]
        k, cout = w.shape[0], w.shape[3]
        g = Graph(device, loss_scale=1.0)
        if first:
            xa = layers.prep_images(g, torch.from_numpy(o["x"]).to(device), means=(0.0, 0.0, 0.0))
        else:
            xa = Act(t["x"].detach().contiguous().to(O.STORAGE).to(device))
        sd = {"L/weights": w, "L/biases": p[name + "/biases"]}
        kw = dict(rate=rate, pool=pool, first=first, normalizer=None,
                  keep_full=(not pool) or name.split("/")[-1].startswith(("conv3", "conv4")))
        layers.conv2d(g, xa, cout, k, "L", **kw)
        g.reset_tape()
        g.store.load_state_dict(sd)
        full, pooled = layers.conv2d(g, xa, cout, k, "L", **kw)
        out = pooled if pool else full
        ref = t["pool"] if pool else t["a"]
        if pool:
            pooled.grad = t["pool"].grad.contiguous().to(O.STORAGE).to(device)
            if full is not None:
                a2 = t["a"].detach().requires_grad_(True)
                through = torch.autograd.grad(O.max_pool(a2, 2, 2), a2, t["pool"].grad)[0]
                full.grad = (t["a"].grad - through).contiguous().to(O.STORAGE).to(device)
        else:
            full.grad = t["a"].grad.contiguous().to(O.STORAGE).to(device)
        g.backward()
        torch.cuda.synchronize()
        e_out = np.abs(out.data.float().cpu().numpy() - ref.detach().numpy()).max() / max(1.0, float(ref.abs().max()))
        dv = g.store.vars
        dwd, dwo = dv["L/weights"].grad.cpu().numpy(), tp[name + "/weights"].grad.numpy()
        e_db = _rel(dv["L/biases"].grad.cpu().numpy(), tp[name + "/biases"].grad.numpy())
        l2_dw = _l2(dwd, dwo)
        l2_dx = bad = 0.0
        if not first:
            dxd, dxo = xa.grad.float().cpu().numpy(), t["x"].grad.numpy()
            l2_dx = _l2(dxd, dxo)
            bad = float((np.abs(dxd - dxo) > 2e-3 * TOL * np.abs(dxo).max()).mean())
        print("%-18s out %.2e | dw L2 %.2e | dbias %.2e | dx L2 %.2e, %.1e of the elements off" % (name, e_out, l2_dw, e_db, l2_dx, bad))
        # bars as for model_vgg's layers (test_gpu_fullsize_nets.py): element-wise on outputs and bias gradients,
        # relative L2 + fraction of moved elements where a 16-bit ReLU / arg-max decision may flip
        assert e_out <= (4e-3 if first else 2e-3) * TOL, (name, e_out)
        assert e_db < 5e-3 * TOL, (name, e_db)
        assert l2_dw < 2e-2 * TOL and l2_dx < 3e-2 * TOL and bad < 3e-3, (name, l2_dw, l2_dx, bad)
        del g, xa, full, pooled
        torch.cuda.empty_cache()


def test_pixellinknet_512_batch32_replicated_equals_n2(device, pl512):
    """configs[2]'s batch: the pair replicated 16x.  Without batch norm every image is independent, the losses are
    batch means / ratios of batch sums, so logits, loss terms and weight gradients equal the n = 2 run's."""
    o = pl512
    S = o["S"]
    px2, lk2, L2, t2, g2 = _device_pixellink(device, o["p"], o["x"], o["pixel"], o["link"], S=S)
    rep = lambda a: np.ascontiguousarray(np.concatenate([a] * 16, axis=0))
    px32, lk32, L32, t32, g32 = _device_pixellink(device, o["p"], rep(o["x"]), rep(o["pixel"]), rep(o["link"]), S=S)
    d = np.abs(px32.reshape(16, 2, 128, 128, 2) - px2[None]).max()
    dl = np.abs(lk32.reshape(16, 2, 128, 128, 16) - lk2[None]).max()
    cs = sorted((_cos(g32[k], g2[k]), k) for k in g2 if g2[k].size >= 64)
    print("PixelLink n=32 vs n=2: losses %s vs %s | logits Linf %.3e / %.3e | lowest gradient cosines %s" % (t32, t2, d, dl, cs[:3]))
    assert d == 0.0 and dl == 0.0                     # no cross-image term in the forward pass: bit-identical per image
    assert abs(t32[0] - t2[0]) < 1e-4 and abs(t32[1] - t2[1]) < 1e-4
    assert cs[0][0] > (0.98 if BF else 0.999)
    assert abs(t32[0] - o["plain"]["l2p"]) < 2e-3 * TOL and abs(t32[1] - o["plain"]["link"]) < 5e-3 * TOL


# ------------------------------------------------------------------------------------------------ (c)
def _device_east(device, p, images, score, geo, mask, S=1024.0):
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as MV
    g = Graph(device, loss_scale=S)
    MV.model(images[:1], graph=g)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    a, b = MV.model(images, graph=g)
    L = MV.loss(score, a, geo, b, mask, graph=g)
    g.backward()
    torch.cuda.synchronize()
    grads = checkpoint.internal_to_tf({n: (v.grad / S).cpu().numpy() for n, v in g.store.vars.items() if v.trainable})
    return a.data.cpu().numpy(), b.data.cpu().numpy(), L.item(), grads


@pytest.fixture(scope="module")
def east640():
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    S = 1024.0
    rng = np.random.default_rng(3)
    p = O.init_model_east_params(rng)
    images, pixel, link, mask = O.synthetic_batch(rng, 2, 640)
    tp = O.to_torch_params(p)
    taps = {}
    F, G, _ = O.model_east(torch.from_numpy(images), tp, True, mixed=True, taps=taps)
    L = O.dice_loss(torch.from_numpy(pixel), F, torch.from_numpy(link), G, torch.from_numpy(mask))
    (L * S).backward(retain_graph=True)
    return dict(p=p, tp=tp, taps=taps, images=images, pixel=pixel, link=link, mask=mask, S=S,
                F=F.detach().numpy(), G=G.detach().numpy(), loss=float(L))


@pytest.mark.skipif(BF, reason="the f16 build's test; the bf16 build has its own full-depth test (test_gpu_fullsize_nets.py)")
def test_resnet50_east_640_unit_by_unit(device, east640):
    """Every bottleneck of ResNet-v1-50 at configs[3]'s resolution (640^2 input, n = 2) on the oracle's input of that unit
    and the oracle's gradient of its output: output to 1e-2, input / weight / BN gradients by relative L2 to 3e-2 — the
    bars of test_gpu_resnet.py::test_bottleneck, at full size and full depth (2048-channel units included)."""
    from tensorflow_ocr_amd import resnet_layers
    from tensorflow_ocr_amd.graph import Act, Graph
    o = east640
    p, tp = o["p"], o["tp"]
    for uname, t in o["taps"].items():
        scope = uname[:-len("/bottleneck_v1")]
        g = Graph(device, loss_scale=1.0)
        xa = Act(t["x"].detach().contiguous().to(O.STORAGE).to(device))
        resnet_layers.bottleneck(g, xa, t["depth"], t["depth_bottleneck"], t["stride"], "u")
        g.reset_tape()
        sd = {k.replace(scope, "u", 1): v for k, v in p.items() if k.startswith(uname + "/")}
        g.store.load_state_dict(sd)
        out = resnet_layers.bottleneck(g, xa, t["depth"], t["depth_bottleneck"], t["stride"], "u")
        out.grad = t["out"].grad.contiguous().to(O.STORAGE).to(device)
        g.backward()
        torch.cuda.synchronize()
        ref = t["out"].detach().numpy()
        e_out = float(np.abs(out.data.float().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
        dx_o = torch.autograd.grad(t["out"], t["x"], t["out"].grad, retain_graph=True)[0].numpy()   # this unit's share only
        l2_dx = _l2(xa.grad.float().cpu().numpy(), dx_o)
        worst = (0.0, "")
        for k in sd:
            if k.endswith(("weights", "gamma", "beta")):
                full = k.replace("u", scope, 1)
                r = _l2(g.store.vars[k].grad.cpu().numpy(), tp[full].grad.numpy())
                worst = max(worst, (r, k))
        print("%-48s out %.2e | dx L2 %.2e | worst parameter-gradient L2 %.2e (%s)" % (scope, e_out, l2_dx, worst[0], worst[1]))
        assert e_out < 1e-2, (uname, e_out)
        assert l2_dx < 3e-2 and worst[0] < 3e-2, (uname, l2_dx, worst)
        del g, xa, out
        torch.cuda.empty_cache()


@pytest.mark.skipif(BF, reason="f16 build")
def test_resnet50_east_640_end_to_end_and_batch64_replicated(device, east640):
    """configs[3]'s one-GPU share in f16: (1) n = 2 end to end against the oracle (chaotic BN net: the end-to-end bars of
    test_gpu_model_vgg.py); (2) n = 64 = the pair replicated 32x must reproduce the n = 2 device results — identical batch
    statistics — through everything batch 64 selects (flat-tile conv_pw over 1.6 M pixels, 256-cout tail-fusion epilogue,
    the wide pointwise weight gradients)."""
    o = east640
    F2, G2, L2, g2 = _device_east(device, o["p"], o["images"], o["pixel"], o["link"], o["mask"], o["S"])
    assert F2.shape == (2, 160, 160, 1) and G2.shape == (2, 160, 160, 8)
    ogr = {k: (v.grad / o["S"]).numpy() for k, v in o["tp"].items() if v.grad is not None}
    glob = _cos(np.concatenate([g2[k].ravel() for k in sorted(ogr)]), np.concatenate([ogr[k].ravel() for k in sorted(ogr)]))
    # what 16-bit storage alone does to this 53-layer BN net at n = 2: the oracle against ITSELF, f32 vs f16-storage mode
    # (measured 0.83 between device and oracle with every unit matching to < 1e-2 in the unit-by-unit test above: the
    # end-to-end cosine is the net's sensitivity, not a kernel property)
    tp32 = O.to_torch_params(o["p"])
    F32_, G32_, _ = O.model_east(torch.from_numpy(o["images"]), tp32, True, mixed=False)
    (O.dice_loss(torch.from_numpy(o["pixel"]), F32_, torch.from_numpy(o["link"]), G32_, torch.from_numpy(o["mask"])) * o["S"]).backward()
    g32 = {k: (v.grad / o["S"]).numpy() for k, v in tp32.items() if v.grad is not None}
    intrinsic = _cos(np.concatenate([g32[k].ravel() for k in sorted(ogr)]), np.concatenate([ogr[k].ravel() for k in sorted(ogr)]))
    print("EAST R50 640^2 n=2: loss %.5f vs %.5f | F_score mean|d| %.3e | global gradient cosine %.4f (oracle f32 vs its own "
          "f16-storage mode: %.4f)" % (L2, o["loss"], np.abs(F2 - o["F"]).mean(), glob, intrinsic))
    assert abs(L2 - o["loss"]) < 5e-3 and np.abs(F2 - o["F"]).mean() < 1e-2
    assert glob > min(0.9, intrinsic - 0.05), (glob, intrinsic)
    rep = lambda a: np.ascontiguousarray(np.concatenate([a] * 32, axis=0))
    F64, G64, L64, g64 = _device_east(device, o["p"], rep(o["images"]), rep(o["pixel"]), rep(o["link"]), rep(o["mask"]), o["S"])
    d = np.abs(F64.reshape(32, 2, 160, 160, 1) - F2[None])
    cs = sorted((_cos(g64[k], g2[k]), k) for k in g2 if g2[k].size >= 64)
    print("n=64 vs n=2: loss %.6f vs %.6f | F_score mean|d| %.3e Linf %.3e | lowest gradient cosines %s" % (
        L64, L2, d.mean(), d.max(), cs[:3]))
    # the two runs differ only in reduction ORDER (split-K schedules, BN partial rows); 53 BN layers at random init carry that
    # ulp-level difference to 3.2e-3 mean on F_score and 0.85 on the lowest per-tensor gradient cosine (measured) — the same
    # sensitivity that puts the oracle's own f32 and f16-storage modes at 0.65 globally.  A wrong tile at batch 64 shows as
    # O(0.1) on F_score; bars at 2x the measured spread.
    assert abs(L64 - L2) < 2e-3 and d.mean() < 6.5e-3
    assert cs[0][0] > 0.7


@pytest.mark.skipif(not BF, reason="the bf16 build's test (run in the bf16 child by test_gpu_bf16.py)")
def test_resnet50_east_640_bf16_batch64_replicated_equals_n2(device):
    """configs[3] AS QUOTED — bf16 storage, batch 64 at 640^2 — had no parity check of its own (VERDICT r3 Weak 3: bf16 was
    checked at n = 1, batch 64 only in f16): the n = 2 pair replicated 32x must reproduce the n = 2 device results of the
    same bf16 library (identical batch statistics; only reduction orders change) through every tile variant batch 64
    selects.  The n = 2 bf16 run itself is held to the oracle by test_gpu_fullsize_nets.py / test_gpu_resnet.py."""
    rng = np.random.default_rng(3)
    p = O.init_model_east_params(rng)
    images, pixel, link, mask = O.synthetic_batch(rng, 2, 640)
    S = 1024.0
    F2, G2, L2, g2 = _device_east(device, p, images, pixel, link, mask, S)
    assert F2.shape == (2, 160, 160, 1) and G2.shape == (2, 160, 160, 8) and np.isfinite(L2)
    rep = lambda a: np.ascontiguousarray(np.concatenate([a] * 32, axis=0))
    F64, G64, L64, g64 = _device_east(device, p, rep(images), rep(pixel), rep(link), rep(mask), S)
    d = np.abs(F64.reshape(32, 2, 160, 160, 1) - F2[None])
    dg = np.abs(G64.reshape(32, 2, 160, 160, 8) - G2[None])
    cs = sorted((_cos(g64[k], g2[k]), k) for k in g2 if g2[k].size >= 64)
    glob = _cos(np.concatenate([g64[k].ravel() for k in sorted(g2)]), np.concatenate([g2[k].ravel() for k in sorted(g2)]))
    print("bf16 n=64 vs n=2: loss %.6f vs %.6f | F_score mean|d| %.3e Linf %.3e | geo mean|d| %.3e | global gradient cosine "
          "%.4f, lowest per tensor %s" % (L64, L2, d.mean(), d.max(), dg.mean(), glob, cs[:3]))
    # 53 BN layers at random init carry the ulp-level reduction-order differences of bf16 (8x f16's rounding) into the
    # outputs; a wrong tile at batch 64 shows as O(0.1) on F_score and a global cosine near 0
    assert abs(L64 - L2) < 1.6e-2 and d.mean() < 5e-2 and dg.mean() < 5e-2
    assert glob > 0.5 and np.isfinite(d).all()


# ------------------------------------------------------------------------------------------------ (d)
@pytest.mark.skipif(BF, reason="bench.py's default build")
def test_bench_losses_against_the_oracle(device):
    """`bench.py --batch 2 --loss-trace`: the losses of its first three optimiser steps (the engine-build steps ARE
    training steps) against three Adam steps of the oracle on the same synthetic batch and the same initial weights
    (Graph(seed=1) re-created here; bench.py's data seed is 100 + rank).  Bars = the end-to-end sensitivity of this BN net
    under 16-bit storage (test_gpu_model_vgg.py), growing with the step."""
    from tensorflow_ocr_amd import checkpoint, synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    size, n = 256, 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(n), "--size", str(size), "--steps", "1",
                        "--warmup", "0", "--loss-trace", "--no-cpu-baseline", "--no-config-legs"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=540)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    trace = out["loss_trace"]
    assert len(trace) == 4 and abs(trace[-1] - out["loss"]) < 1e-5
    # the same initial weights and data as bench.py's rank 0
    g = Graph(device, loss_scale=1024.0, seed=1)
    images, pixel, link, mask = synthetic.make_batch(np.random.default_rng(100), n, size)
    M.model_vgg(images, graph=g)
    g.reset_tape()
    p = checkpoint.internal_to_tf(g.store.state_dict())
    names = sorted(k for k in p if not k.endswith(("moving_mean", "moving_variance")))
    reg = {k for k in names if k.endswith("weights")}
    m = {k: np.zeros_like(p[k]) for k in names}
    v = {k: np.zeros_like(p[k]) for k in names}
    oracle = []
    for t in range(1, 4):
        tp = O.to_torch_params(p)
        updates = {}
        px, lk, _ = O.model_vgg(torch.from_numpy(images), tp, True, mixed=True, updates=updates)
        L = O.dice_loss(torch.from_numpy(pixel), px, torch.from_numpy(link), lk, torch.from_numpy(mask))
        L.backward()
        oracle.append(float(L))
        lr = O.exponential_decay(1e-4, t - 1)
        for k in names:
            gk = tp[k].grad.numpy() if tp[k].grad is not None else np.zeros_like(p[k])
            if k in reg:
                gk = gk + 1e-5 * p[k]                        # slim.l2_regularizer(1e-5) on every conv kernel
            p[k], m[k], v[k] = O.adam_update(p[k], gk, m[k], v[k], t, lr)
        for k, val in updates.items():                       # BN moving statistics (not used in training mode)
            p[k] = val.detach().numpy() if hasattr(val, "detach") else np.asarray(val)
    print("bench.py losses %s | oracle %s" % (trace[:3], [round(x, 6) for x in oracle]))
    for i, bar in enumerate((5e-3, 1e-2, 2e-2)):
        assert abs(trace[i] - oracle[i]) < bar, (i, trace[i], oracle[i])
    assert trace[2] < trace[0]                               # and it trains
