"""GPU: the fuse heads with one launch per kernel kind over the head sources (VERDICT r3 item 1b;
nets/model_vgg_16.py:160-175, nets/pixellink.py:55-67) — every batched entry point against the per-map entry point it
replaces (bit-identical where the arithmetic is the same, a stated bar where a summation order changed), and the whole
model_vgg step with the batched heads against the per-map heads."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32 = torch.float32


def _g(device):
    from tensorflow_ocr_amd.graph import Graph
    return Graph(device)


def _feat(rng, P, cin, device, F16):
    return torch.from_numpy(rng.standard_normal((P, cin)).astype(np.float32)).to(device).to(F16)


# (P, cin): the four VGG sources at 64^2 x 2 ... plus ragged / odd pixel counts
SHAPES = [(2 * 4 * 4, 1024), (2 * 4 * 4, 512), (2 * 8 * 8, 512), (2 * 16 * 16, 256)]
ODD = [(77, 128), (1, 256), (130, 384), (4097, 128)]


@pytest.mark.parametrize("shapes", [SHAPES, ODD, SHAPES[:1], [(200000, 256)]])
@pytest.mark.parametrize("with_bias", [False, True])
def test_conv_batch_equals_per_map_and_statistics(device, shapes, with_bias):
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    g = _g(device)
    g.workspace()
    rng = np.random.default_rng(len(shapes) + 10 * with_bias)
    C = 18
    items, refs = [], []
    for P, cin in shapes:
        x = _feat(rng, P, cin, device, F16)
        w = torch.from_numpy((rng.standard_normal((cin, C)) / np.sqrt(cin)).astype(np.float32)).to(device)
        kc, ck = torch.empty((32, cin), dtype=F16, device=device), torch.empty((cin, 32), dtype=F16, device=device)
        ops.pack_weights_small(w, kc, ck)
        bias = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(device) if with_bias else None
        out = torch.empty((P, C), dtype=F32, device=device)
        T = ops.conv1x1_small_batch_rows(P)
        assert 1 <= T <= 1024
        part = torch.empty((T, 2, C), dtype=F32, device=device)
        items.append((x, kc, bias, out, part))
        ref = torch.empty((P, C), dtype=F32, device=device)
        ops.conv1x1_small(x, kc, C, ref, bias)
        refs.append(ref)
    ops.conv1x1_small_batch(items)
    torch.cuda.synchronize()
    for (x, kc, bias, out, part), ref in zip(items, refs):
        assert torch.equal(out, ref)
        z = ref.double().cpu().numpy()
        got = part.double().sum(0).cpu().numpy()
        assert np.allclose(got[0], z.sum(0), rtol=1e-5, atol=1e-4 * np.abs(z).sum(0).max())
        assert np.allclose(got[1], (z * z).sum(0), rtol=1e-5)
    # finalize: batched vs the per-map launch on the SAME partial rows -> the same f64 row sums, identical results
    outs_b, outs_r = [], []
    fin = []
    for (x, kc, bias, out, part) in items:
        P = out.shape[0]
        ga = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).to(device)
        be = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(device)
        mm0 = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(device)
        mv0 = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).to(device)
        b = [torch.empty(C, dtype=F32, device=device) for _ in range(4)] + [mm0.clone(), mv0.clone()]
        r = [torch.empty(C, dtype=F32, device=device) for _ in range(4)] + [mm0.clone(), mv0.clone()]
        fin.append((part, part.shape[0], C, float(P), ga, be, b[4], b[5], b[0], b[1], b[2], b[3]))
        stage = torch.empty(ops.bn_reduce_workspace(part.shape[0], C), dtype=torch.uint8, device=device)
        ops.bn_finalize(part, part.shape[0], C, float(P), ga, be, 1e-5, 0.997, r[4], r[5], r[0], r[1], r[2], r[3], stage)
        outs_b.append(b)
        outs_r.append(r)
    ops.bn_finalize_batch(fin, 1e-5, 0.997)
    torch.cuda.synchronize()
    for b, r in zip(outs_b, outs_r):
        for tb, tr in zip(b, r):
            assert torch.allclose(tb, tr, rtol=1e-6, atol=1e-7), (tb - tr).abs().max()


@pytest.mark.parametrize("shapes", [SHAPES, [(77, 128), (130, 384), (4097, 128)]])
def test_dgrad_batch_equals_per_map(device, shapes):
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    g = _g(device)
    rng = np.random.default_rng(3)
    C = 18
    items, refs = [], []
    for k, (P, cin) in enumerate(shapes):
        dz = torch.from_numpy(rng.standard_normal((P, C)).astype(np.float32)).to(device)
        w = torch.from_numpy((rng.standard_normal((cin, C)) / 4).astype(np.float32)).to(device)
        kc, ck = torch.empty((32, cin), dtype=F16, device=device), torch.empty((cin, 32), dtype=F16, device=device)
        ops.pack_weights_small(w, kc, ck)
        acc = k % 2 == 1
        old = _feat(rng, P, cin, device, F16)
        dx, ref = old.clone(), old.clone()
        items.append((dz, ck, dx, acc))
        ops.conv1x1_small_dgrad(dz, ck, C, ref, acc)
        refs.append(ref)
    ops.conv1x1_small_dgrad_batch(items)
    torch.cuda.synchronize()
    for (dz, ck, dx, acc), ref in zip(items, refs):
        assert torch.equal(dx, ref)


@pytest.mark.parametrize("shapes", [SHAPES, [(77, 128), (1, 256), (130, 384), (4097, 128)], [(150001, 256)]])
def test_wgrad_batch_vs_float64_and_per_map(device, shapes):
    """dw = x^T dz with x in 16-bit storage and dz rounded to it on load: against float64 on the rounded operands
    (f32 accumulation: 1e-5 of the result's scale) and against the per-map entry point (pad pass + generic kernel)."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    g = _g(device)
    g.workspace()
    rng = np.random.default_rng(5)
    C = 18
    items = []
    for P, cin in shapes:
        x = _feat(rng, P, cin, device, F16)
        dz = torch.from_numpy(rng.standard_normal((P, C)).astype(np.float32)).to(device)
        dw = torch.full((cin, C), float("nan"), dtype=F32, device=device)
        slab = torch.empty(ops.conv1x1_small_wgrad_batch_slab_bytes(P, cin), dtype=torch.uint8, device=device)
        items.append((x, dz, dw, slab))
    ops.conv1x1_small_wgrad_batch(items)
    torch.cuda.synchronize()
    for x, dz, dw, slab in items:
        want = x.double().cpu().numpy().T @ dz.to(F16).double().cpu().numpy()
        got = dw.cpu().numpy()
        scale = np.abs(want).max() + 1e-30
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= 2e-5 * scale * max(1.0, np.sqrt(x.shape[0] / 1000.0)), np.abs(got - want).max() / scale
        ref = torch.empty_like(dw)
        ops.conv1x1_small_wgrad(x, dz, C, ref, g.ws)
        # (the per-map route for odd pixel counts keeps dz in f32; the batched kernel rounds it to 16 bits like the MFMA route)
        assert np.abs(ref.cpu().numpy() - got).max() <= 2e-3 * scale


def test_sc_bn_bwd_batch_and_colsum_batch(device):
    from tensorflow_ocr_amd import ops
    g = _g(device)
    g.workspace()
    rng = np.random.default_rng(7)
    items, refs = [], []
    for P, C in ((32, 18), (128, 18), (2049, 18), (524, 2), (524, 16)):
        mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(device)
        z, dout = mk(P, C), mk(P, C)
        scale, shift, mean = mk(C), mk(C), mk(C)
        invstd = torch.from_numpy(rng.uniform(0.5, 2.0, C).astype(np.float32)).to(device)
        T = ops.sc_num_partials(P, C)
        out = [torch.empty(C, dtype=F32, device=device), torch.empty(C, dtype=F32, device=device), torch.empty((P, C), dtype=F32, device=device)]
        ref = [torch.empty(C, dtype=F32, device=device), torch.empty(C, dtype=F32, device=device), torch.empty((P, C), dtype=F32, device=device)]
        items.append((z, scale, shift, mean, invstd, dout, out[0], out[1], out[2], torch.empty((T, 2, C), dtype=F32, device=device), True))
        ops.sc_bn_bwd(z, scale, shift, mean, invstd, dout, True, ref[0], ref[1], ref[2], g.ws)
        refs.append((out, ref))
    for lo in (0, 4):                    # four items per launch
        ops.sc_bn_bwd_batch(items[lo:lo + 4])
    torch.cuda.synchronize()
    for out, ref in refs:
        for a, b in zip(out, ref):       # same partial rows; the row sums in f64 here, f32 there
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max())), (a - b).abs().max()
    # column sums
    its, want = [], []
    for P, C in ((32, 18), (2049, 18), (100, 7)):
        x = torch.from_numpy(rng.standard_normal((P, C)).astype(np.float32)).to(device)
        T = ops.sc_num_partials(P, C)
        o = torch.empty(C, dtype=F32, device=device)
        its.append((x, o, torch.empty((T + 1, 2, C), dtype=F32, device=device)))
        want.append(x.double().sum(0).cpu().numpy())
    ops.sc_colsum_batch(its)
    torch.cuda.synchronize()
    for (x, o, _), w in zip(its, want):
        assert np.allclose(o.cpu().numpy(), w, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("P", [32, 1000, 32 * 128 * 128 // 16])
@pytest.mark.parametrize("with_bias", [False, True])
def test_pointwise_pair_equals_the_two_per_head_passes(device, P, with_bias):
    from tensorflow_ocr_amd import ops
    g = _g(device)
    g.workspace()
    rng = np.random.default_rng(P)
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(device)
    x = mk(P, 18)
    w_px, w_lk = mk(2, 2), mk(16, 16)
    b_px, b_lk = (mk(2), mk(16)) if with_bias else (None, None)
    z_px, z_lk = torch.empty((P, 2), dtype=F32, device=device), torch.empty((P, 16), dtype=F32, device=device)
    T = ops.sc_pointwise_pair_num_partials(P)
    pp, pl = torch.empty((T, 2, 2), dtype=F32, device=device), torch.empty((T, 2, 16), dtype=F32, device=device)
    ops.sc_pointwise_pair_fwd(x, w_px, b_px, w_lk, b_lk, z_px, z_lk, pp, pl)
    r_px, r_lk = torch.empty_like(z_px), torch.empty_like(z_lk)
    ops.sc_pointwise_fwd(x, 0, 2, w_px, r_px, 0, 2, b_px)
    ops.sc_pointwise_fwd(x, 2, 16, w_lk, r_lk, 0, 16, b_lk)
    torch.cuda.synchronize()
    x64, r64 = x.double().cpu().numpy(), None
    for got, ref, w_, b_, sl in ((z_px, r_px, w_px, b_px, slice(0, 2)), (z_lk, r_lk, w_lk, b_lk, slice(2, 18))):
        want = x64[:, sl] @ w_.double().cpu().numpy() + (b_.double().cpu().numpy() if b_ is not None else 0.0)
        e_pair, e_ref = np.abs(got.double().cpu().numpy() - want).max(), np.abs(ref.double().cpu().numpy() - want).max()
        print("pair vs per-head: max |diff| %.3e | vs float64: pair %.3e, per-head %.3e" % (float((got - ref).abs().max()), e_pair, e_ref))
        assert e_pair <= 2e-6 * (1 + np.abs(want).max()) and e_ref <= 2e-6 * (1 + np.abs(want).max())
        assert torch.allclose(got, ref, rtol=0, atol=4e-6 * float(ref.abs().max()))
    for part, z in ((pp, z_px), (pl, z_lk)):
        zz = z.double().cpu().numpy()
        got = part.double().sum(0).cpu().numpy()
        assert np.allclose(got[0], zz.sum(0), rtol=1e-5, atol=1e-5 * np.abs(zz).sum(0).max())
        assert np.allclose(got[1], (zz * zz).sum(0), rtol=1e-5)
    # backward: input gradient of both heads in one tensor, both weight (+ bias) gradients
    dz_px, dz_lk = mk(P, 2), mk(P, 16)
    dx = torch.full((P, 18), float("nan"), dtype=F32, device=device)
    dw_px, dw_lk = torch.empty((2, 2), dtype=F32, device=device), torch.empty((16, 16), dtype=F32, device=device)
    db_px, db_lk = (torch.empty(2, dtype=F32, device=device), torch.empty(16, dtype=F32, device=device)) if with_bias else (None, None)
    ops.sc_pointwise_pair_bwd(x, dz_px, dz_lk, w_px, w_lk, dx, dw_px, db_px, dw_lk, db_lk, g.ws)
    rdx = torch.zeros((P, 18), dtype=F32, device=device)
    ops.sc_pointwise_dgrad(dz_px, 0, 2, w_px, rdx, 0, 2)
    ops.sc_pointwise_dgrad(dz_lk, 0, 16, w_lk, rdx, 2, 16)
    rw_px, rw_lk = torch.empty_like(dw_px), torch.empty_like(dw_lk)
    rb_px, rb_lk = (torch.empty(2, dtype=F32, device=device), torch.empty(16, dtype=F32, device=device)) if with_bias else (None, None)
    ops.sc_pointwise_wgrad(x, 0, 2, dz_px, 0, 2, rw_px, rb_px, g.ws)
    ops.sc_pointwise_wgrad(x, 2, 16, dz_lk, 0, 16, rw_lk, rb_lk, g.ws)
    torch.cuda.synchronize()
    assert torch.equal(dx, rdx)
    assert torch.equal(dw_px, rw_px) and torch.equal(dw_lk, rw_lk)
    if with_bias:
        assert torch.equal(db_px, rb_px) and torch.equal(db_lk, rb_lk)


def test_sc_act_batch(device):
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(9)
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(device)
    for relu in (True, False):
        items, refs = [], []
        for P, C in ((1000, 2), (1000, 16), (33, 18)):
            z, sc, sh = mk(P, C), mk(C), mk(C)
            out, ref = torch.empty_like(z), torch.empty_like(z)
            items.append((z, sc, sh, out))
            ops.sc_fuse(ref.view(1, 1, P, C), z, sc, sh, relu=relu)
            refs.append(ref)
        ops.sc_act_batch(items, relu)
        torch.cuda.synchronize()
        for (z, sc, sh, out), ref in zip(items, refs):
            assert torch.equal(out, ref)


@pytest.mark.parametrize("net", ["model_vgg", "pixellink"])
def test_whole_step_batched_heads_vs_per_map_heads(device, net, monkeypatch):
    """The same net, same weights, same batch, with layers.BATCH_HEADS on and off: outputs, loss and every gradient agree to
    the summation-order differences of the head statistics / weight gradients (the trunk is untouched)."""
    from oracle import ocr_oracle as O
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Graph
    rng = np.random.default_rng(11)
    images, pixel, link, mask = O.synthetic_batch(rng, 2, 128)
    res = {}
    for batched in (True, False):
        monkeypatch.setattr(layers, "BATCH_HEADS", batched)
        g = Graph(device, loss_scale=128.0, seed=3)
        if net == "model_vgg":
            from tensorflow_ocr_amd.nets import model_vgg_16 as M
            def fl():
                px, lk = M.model_vgg(images, graph=g)
                return px, lk, M.loss(pixel, px, link, lk, mask, graph=g)
        else:
            from tensorflow_ocr_amd.nets import pixellink as PL
            def fl():
                nt = PL.PixelLinkNet(images, graph=g, input_norm=(120.0, 60.0))
                return nt.pixel_cls, nt.link_cls, nt.build_loss(pixel[..., 0], link)
        fl()
        g.reset_tape()
        g.ensure_materialised()
        g.store.reset_non_trainable()
        px, lk, L = fl()
        g.backward()
        torch.cuda.synchronize()
        res[batched] = (px.data.clone(), lk.data.clone(), L.item(), g.store.flat_grad.clone(),
                        [n for n in g.store.order])
    a, b = res[True], res[False]
    assert a[4] == b[4]                                   # same variables in the same order
    assert torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-4) and torch.allclose(a[1], b[1], rtol=1e-4, atol=1e-4)
    assert abs(a[2] - b[2]) <= 1e-5 * max(1.0, abs(b[2]))
    ga, gb = a[3].double(), b[3].double()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert cos > 0.9999, cos
    assert float((ga - gb).norm() / gb.norm()) < 2e-2


@pytest.mark.parametrize("P,cin,C", [(1, 32, 9), (77, 32, 9), (4097, 32, 1), (200000, 32, 9), (5000, 64, 18), (3001, 16, 8), (130, 8, 2)])
def test_small_wgrad_on_narrow_maps(device, P, cin, C):
    """ocr_conv1x1_small_wgrad_f16 on cin = 8..64 (conv1x1_small_wgrad_narrow_kernel: x in 16-byte chunks, dz in f32)
    against x^T dz in float64."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    g = _g(device)
    g.workspace()
    rng = np.random.default_rng(P + cin + C)
    x = _feat(rng, P, cin, device, F16)
    dz = torch.from_numpy(rng.standard_normal((P, C)).astype(np.float32)).to(device)
    dw = torch.full((cin, C), float("nan"), dtype=F32, device=device)
    ops.conv1x1_small_wgrad(x, dz, C, dw, g.ws)
    torch.cuda.synchronize()
    want = x.double().cpu().numpy().T @ dz.double().cpu().numpy()
    got = dw.cpu().numpy()
    scale = np.abs(want).max() + 1e-30
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-5 * scale * max(1.0, np.sqrt(P / 1000.0)), np.abs(got - want).max() / scale


def test_sigmoid_split_and_its_gradient(device):
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(31)
    P, C, c0 = 1237, 9, 1
    z = torch.from_numpy(rng.standard_normal((P, C)).astype(np.float32) * 3).to(device)
    o0 = torch.empty((P, c0), dtype=F32, device=device)
    o1 = torch.empty((P, C - c0), dtype=F32, device=device)
    ops.sc_sigmoid_split(z, c0, o0, o1)
    ref = torch.empty_like(z)
    ops.sc_sigmoid(z, ref)
    torch.cuda.synchronize()
    assert torch.equal(o0, ref[:, :c0]) and torch.equal(o1, ref[:, c0:])
    d0 = torch.from_numpy(rng.standard_normal((P, c0)).astype(np.float32)).to(device)
    d1 = torch.from_numpy(rng.standard_normal((P, C - c0)).astype(np.float32)).to(device)
    dz = torch.empty_like(z)
    ops.sc_sigmoid_split_bwd(o0, d0, o1, d1, dz)
    want = torch.empty_like(z)
    ops.sc_sigmoid_bwd(ref, torch.cat([d0, d1], 1).contiguous(), want)
    torch.cuda.synchronize()
    assert torch.equal(dz, want)
    ops.sc_sigmoid_split_bwd(o0, None, o1, d1, dz)            # a head nobody differentiated: zero gradient
    torch.cuda.synchronize()
    assert float(dz[:, :c0].abs().max()) == 0.0 and torch.equal(dz[:, c0:], want[:, c0:])


def test_merged_sigmoid_heads_equal_the_two_heads(device):
    """resnet_layers.sigmoid_heads (one merged 1x1 convolution for F_score + geo_map) against two sigmoid_head calls with
    the same weights: activations bit for bit (same kernel, same per-output arithmetic), gradients to f32 summation order."""
    from tensorflow_ocr_amd import checkpoint, resnet_layers as R
    from tensorflow_ocr_amd.graph import Act, F16, Graph
    rng = np.random.default_rng(32)
    n, h, w, cin = 2, 24, 40, 32
    feat_v = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    tf_sd = {"a/weights": rng.standard_normal((1, 1, cin, 1)).astype(np.float32) * 0.2, "a/biases": rng.standard_normal(1).astype(np.float32),
             "b/weights": rng.standard_normal((1, 1, cin, 8)).astype(np.float32) * 0.2, "b/biases": rng.standard_normal(8).astype(np.float32)}
    d0 = rng.standard_normal((n, h, w, 1)).astype(np.float32)
    d1 = rng.standard_normal((n, h, w, 8)).astype(np.float32)

    def run(merge):
        old = R.MERGE_HEADS
        R.MERGE_HEADS = merge
        try:
            g = Graph(device, loss_scale=1.0)
            mk = lambda: Act(torch.from_numpy(feat_v).to(device).to(F16))
            R.sigmoid_heads(g, mk(), (1, 8), ("a", "b"))
            g.reset_tape()
            g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, tf_sd))
            feat = mk()
            o0, o1 = R.sigmoid_heads(g, feat, (1, 8), ("a", "b"))
            o0.grad = torch.from_numpy(d0).to(device)
            o1.grad = torch.from_numpy(d1).to(device)
            g.backward()
            torch.cuda.synchronize()
            grads = checkpoint.internal_to_tf({k: v.grad.cpu().numpy() for k, v in g.store.vars.items() if v.trainable})
            return o0.data.clone(), o1.data.clone(), feat.grad.float().cpu().numpy(), grads
        finally:
            R.MERGE_HEADS = old
    m0, m1, mdx, mg = run(True)
    s0, s1, sdx, sg = run(False)
    assert torch.equal(m0, s0) and torch.equal(m1, s1)
    assert np.abs(mdx - sdx).max() <= 2e-3 * np.abs(sdx).max()
    assert set(mg) == set(sg) == set(tf_sd)
    for k in sg:
        assert mg[k].shape == tf_sd[k].shape               # the merged variable splits back into the reference's two
        ref = sg[k].reshape(tf_sd[k].shape)
        assert np.abs(mg[k] - ref).max() <= 2e-3 * (np.abs(ref).max() + 1e-30), k


@pytest.mark.parametrize("pc,G,P", [(1, 1, 2 * 37 * 41), (1, 1, 300000), (2, 2, 5000), (2, 1, 777)])
def test_dice_loss_kernels_against_the_oracle(device, pc, G, P):
    """ocr_dice_loss_fwd / _bwd (the vector forms for pc = G = 1 and pc = G = 2, the scalar form otherwise) against the
    oracle's dice loss (nets/model_vgg_16.py:179-225) and its autograd gradient."""
    from oracle import ocr_oracle as O
    from tensorflow_ocr_amd import ops
    g = _g(device)
    g.workspace()
    rng = np.random.default_rng(pc * 10 + G + P)
    ytp = (rng.uniform(size=(1, P, 1, 1)) > 0.7).astype(np.float32)
    ytl = (rng.uniform(size=(1, P, 1, 8)) > 0.6).astype(np.float32)
    m = (rng.uniform(size=(1, P, 1, 1)) > 0.1).astype(np.float32)
    ypp = rng.uniform(size=(1, P, 1, pc)).astype(np.float32)
    ypl = rng.uniform(size=(1, P, 1, 8 * G)).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(device)
    sums = torch.zeros(27, dtype=F32, device=device)
    loss = torch.zeros(10, dtype=F32, device=device)
    ops.dice_loss_fwd(d(ytp), d(ypp), d(ytl), d(ypl), d(m), sums, loss, g.ws)
    dpp = torch.full((P, pc), float("nan"), dtype=F32, device=device)
    dpl = torch.full((P, 8 * G), float("nan"), dtype=F32, device=device)
    ops.dice_loss_bwd(d(ytp), d(ytl), d(m), sums, 3.0, dpp, dpl)
    torch.cuda.synchronize()
    tp_, tl_ = torch.from_numpy(ypp).requires_grad_(), torch.from_numpy(ypl).requires_grad_()
    want = O.dice_loss(torch.from_numpy(ytp), tp_, torch.from_numpy(ytl), tl_, torch.from_numpy(m))
    (want * 3.0).backward()
    assert abs(float(loss[0]) - float(want)) < 2e-5
    for got, ref in ((dpp, tp_.grad.reshape(P, pc)), (dpl, tl_.grad.reshape(P, 8 * G))):
        ref = ref.numpy()
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-12
