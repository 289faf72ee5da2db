"""GPU: conv1_2-shaped batch-norm layers (64 -> 64 channels at full resolution, read only by their 2x2 pool: nets/vgg.py:17-18
under nets/model_vgg_16.py:144) — the persistent kernel's epilogue picks the y the pool will select (ocr_conv2d_stats_pool_f16),
so bn + ReLU + pool runs over the pooled tensor.  Against the two-pass form (ocr_conv2d_f16 with statistics, then
ocr_bn_relu_pool_idx_f16 over the full-resolution y): y, statistics and the pooled activation bit-identical; the selected y
and its position identical wherever the window's 16-bit activations are distinct (equal activations with different y: the
position of the larger y is kept instead of the first)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F32 = torch.float32


@pytest.mark.parametrize("n,h,w", [(2, 128, 128), (1, 130, 253), (3, 96, 160)])
@pytest.mark.parametrize("neg_gamma", [False, True])
def test_stats_pool_epilogue_equals_two_pass(device, n, h, w, neg_gamma):
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16, Graph
    from tensorflow_ocr_amd._lib import CONV_STATS
    g = Graph(device)
    g.workspace()
    rng = np.random.default_rng(n * h + w)
    c = 64
    x = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(device).to(F16)
    wt = torch.from_numpy((rng.standard_normal((3, 3, c, c)) / 24).astype(np.float32)).to(device)
    kc, ck = torch.empty((9, c, c), dtype=F16, device=device), torch.empty((9, c, c), dtype=F16, device=device)
    ops.pack_weights(wt, kc, ck)
    d = ops.conv_desc((n, h, w, c), c, 3, 3)
    if ops.conv2d_variant(d) != "conv_c64_persist_kernel<64>":
        pytest.skip("map too small for the persistent kernel")
    d.flags = CONV_STATS
    mt = ops.conv2d_num_mtiles(d)
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
    if neg_gamma:
        gamma[::3] *= -1.0
    gamma_t = torch.from_numpy(gamma).to(device)
    beta = torch.from_numpy(rng.normal(0, 0.3, c).astype(np.float32)).to(device)
    ph, pw = (h + 1) // 2, (w + 1) // 2
    # new: one conv launch + the element-wise pass over the pooled tensor
    y1 = torch.empty((n, h, w, c), dtype=F16, device=device)
    p1 = torch.zeros((mt, 2, c), dtype=F32, device=device)
    yp1 = torch.empty((n, ph, pw, c), dtype=F16, device=device)
    ix1 = torch.empty((n, ph, pw, c), dtype=torch.uint8, device=device)
    ops.conv2d_stats_pool(d, x, kc, y1, p1, gamma_t, yp1, ix1)
    # two-pass reference
    y0 = torch.empty_like(y1)
    p0 = torch.zeros_like(p1)
    ops.conv2d(d, x, kc, y0, None, p0)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(p0, p1)
    scale, shift, mean, invstd = (torch.empty(c, dtype=F32, device=device) for _ in range(4))
    mm, mv = torch.zeros(c, device=device), torch.ones(c, device=device)
    stage = torch.empty(ops.bn_reduce_workspace(mt, c), dtype=torch.uint8, device=device)
    ops.bn_finalize(p0, mt, c, float(n * h * w), gamma_t, beta, 1e-5, 0.997, mm, mv, scale, shift, mean, invstd, stage)
    a0 = torch.empty((n, ph, pw, c), dtype=F16, device=device)
    ix0 = torch.empty_like(ix1)
    yp0 = torch.empty_like(yp1)
    ops.bn_relu_pool_idx(y0, scale, shift, True, None, a0, ix0, yp0)
    a1 = torch.empty_like(a0)
    ops.bn_relu_selected(yp1, scale, shift, True, a1, ix1)
    torch.cuda.synchronize()
    assert torch.equal(a0, a1)                                  # the pooled activation: bit-identical
    assert torch.equal(ix0 & 4, ix1 & 4)                        # ... and with it bit 2 of the positions (activation > 0)
    diff = (ix0 & 3) != (ix1 & 3)
    frac = float(diff.float().mean())
    print("positions differ on %.4f %% of the pooled elements (equal 16-bit activations inside a window)" % (100 * frac))
    # wherever the positions differ the window's activations at the two positions are equal (both zero after the ReLU, or
    # two y that round to one 16-bit activation), and the new form holds the larger (gamma >= 0) / smaller (gamma < 0) y
    yf = y0.float().cpu().numpy()
    sc, sh = scale.cpu().numpy(), shift.cpu().numpy()
    i0, i1 = (ix0 & 3).cpu().numpy(), (ix1 & 3).cpu().numpy()
    where = np.argwhere(diff.cpu().numpy())
    for b, py, px_, ch in where[:2000]:
        def at(pos):
            return yf[b, min(2 * py + pos // 2, h - 1), min(2 * px_ + pos % 2, w - 1), ch]
        ya, yb = at(i0[b, py, px_, ch]), at(i1[b, py, px_, ch])
        act = lambda v: np.float16(max(np.float32(v) * sc[ch] + sh[ch], 0.0)) if True else 0
        assert act(ya) == act(yb), (b, py, px_, ch, ya, yb)
        assert (yb >= ya) if gamma[ch] >= 0 else (yb <= ya)
    same = ~diff
    assert torch.equal(yp0[same], yp1[same])
    assert frac < 0.6                                            # (about half the windows are all-zero after the ReLU at beta ~ 0)


def test_model_vgg_step_with_and_without_the_pool_epilogue(device, monkeypatch):
    """Whole model_vgg step at 256^2 (conv1_2 runs the persistent kernel there) with layers.FUSE_BN_POOL_EPI on and off:
    forward outputs bit-identical, gradients equal up to the tie positions' routing."""
    from oracle import ocr_oracle as O
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    rng = np.random.default_rng(21)
    images, pixel, link, mask = O.synthetic_batch(rng, 2, 256)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(layers, "FUSE_BN_POOL_EPI", on)
        g = Graph(device, loss_scale=128.0, seed=5)
        M.model_vgg(images, graph=g)
        g.reset_tape()
        g.ensure_materialised()
        g.store.reset_non_trainable()
        px, lk = M.model_vgg(images, graph=g)
        L = M.loss(pixel, px, link, lk, mask, graph=g)
        g.backward()
        torch.cuda.synchronize()
        res[on] = (px.data.clone(), lk.data.clone(), L.item(), g.store.flat_grad.clone())
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
    # where two window elements round to one 16-bit activation the gradient is routed to the larger y (as the f32
    # reference would) instead of to the first of the two: a handful of relocated gradient elements per map
    ga, gb = a[3].double(), b[3].double()
    assert float((ga - gb).norm() / gb.norm()) < 1e-2
    assert float((ga * gb).sum() / (ga.norm() * gb.norm())) > 0.9999
