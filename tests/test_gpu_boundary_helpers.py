"""GPU: the helper names of the Python boundary — `model.get_pos_and_neg_masks`,
`model.OHNM_single_image`, `model.OHNM_batch` (nets/model.py:161-201), `model_vgg_16.cal_link_loss`
(nets/model_vgg_16.py:227-241), `pixellink_fn.tf_pixellink_get_rbox` (tool/pixellink_fn.py:112-118) — each against the
oracle's restatement of the cited lines."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu


def test_get_pos_and_neg_masks(device):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model as M
    g = Graph(device)
    rng = np.random.default_rng(0)
    labels = rng.choice([0.0, 1.0, 2.0, -1.0, 0.5], size=(3, 37, 41, 1)).astype(np.float32)
    pos, neg = M.get_pos_and_neg_masks(labels, graph=g)
    rp, rn = O.get_pos_and_neg_masks(labels)
    assert pos.dtype == torch.bool and tuple(pos.shape) == labels.shape
    assert np.array_equal(pos.cpu().numpy(), rp) and np.array_equal(neg.cpu().numpy(), rn)


@pytest.mark.parametrize("hw,n_pos", [(4096, 100), (4096, 0), (1000, 400), (777, 5), (64, 64)])
def test_ohnm_single_image_exact(device, hw, n_pos):
    """The selected-negative mask is index work: exact, ties at the threshold included."""
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model as M
    g = Graph(device)
    rng = np.random.default_rng(hw + n_pos)
    scores = rng.random(hw).astype(np.float32)
    scores[rng.integers(0, hw, hw // 8)] = np.float32(0.25)          # many exact ties
    neg = rng.random(hw) < 0.7
    got = M.OHNM_single_image(scores, n_pos, neg, graph=g).cpu().numpy()
    want = O.ohnm_single_image(scores, n_pos, neg)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    if n_pos > 0 and neg.sum() > 0:
        assert got.sum() >= min(3 * n_pos, int(neg.sum()))           # tie-inclusive: at least k


@pytest.mark.parametrize("hw,n_pos", [(4096, 100), (1000, 250), (333, 7)])
def test_ohnm_single_image_on_signed_scores(device, hw, n_pos):
    """the reference's top_k(-neg_conf) works on any real scores (nets/model.py:161-184); the radix select
    orders the float bits through a sign-aware key, so logits, mixed signs, -0.0 / +0.0 ties and all-negative maps give
    the oracle's selection exactly."""
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model as M
    g = Graph(device)
    rng = np.random.default_rng(hw * 3 + n_pos)
    for kind in ("logits", "all_negative", "zeros"):
        scores = (rng.standard_normal(hw) * 3).astype(np.float32)
        if kind == "all_negative":
            scores = -np.abs(scores) - np.float32(0.5)
        if kind == "zeros":
            scores[rng.integers(0, hw, hw // 3)] = np.float32(0.0)
            scores[rng.integers(0, hw, hw // 3)] = np.float32(-0.0)
        scores[rng.integers(0, hw, hw // 8)] = np.float32(-1.25)     # exact ties below zero
        neg = rng.random(hw) < 0.7
        got = M.OHNM_single_image(scores, n_pos, neg, graph=g).cpu().numpy()
        want = O.ohnm_single_image(scores, n_pos, neg)
        assert np.array_equal(got, want), kind
        assert got.sum() >= min(3 * n_pos, int(neg.sum()))


def test_ohnm_batch_exact_and_equals_the_loss_kernels_mask(device):
    """OHNM_batch on P(neg) computed with the shared deterministic exp = the mask the fused loss mines."""
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model as M
    g = Graph(device)
    rng = np.random.default_rng(3)
    n, h, w = 5, 32, 48
    logits = rng.standard_normal((n, h, w, 2)).astype(np.float32)
    label = (rng.random((n, h, w, 1)) < 0.1).astype(np.float32)
    label[3] = 0                                                      # an image without positives selects nothing
    scores = O.neg_score_f32(logits[..., 0], logits[..., 1]).reshape(n, -1)
    pos, neg = O.get_pos_and_neg_masks(label.reshape(n, -1))
    got = M.OHNM_batch(n, scores, pos, neg, graph=g).cpu().numpy()
    want = O.ohnm_batch(scores, pos, neg)
    assert np.array_equal(got, want) and got[3].sum() == 0
    # the same selection comes out of the training loss (nets/model.py: loss -> ocr_softmax_loss_selected)
    px = layers.SmallAct(torch.from_numpy(logits).to(device))
    lk = layers.SmallAct(torch.from_numpy(rng.standard_normal((n, h, w, 16)).astype(np.float32)).to(device))
    res = M.loss(label, px, (rng.random((n, h, w, 8)) < 0.5).astype(np.float32), lk, None, graph=g)
    assert np.array_equal(res.selected_mask().cpu().numpy().reshape(n, -1).astype(np.float32), want)
    # a batch_size below the tensor's (the reference passes a literal 14): the first rows only
    assert np.array_equal(M.OHNM_batch(2, scores, pos, neg, graph=g).cpu().numpy(), want[:2])


def test_cal_link_loss_value_slices_and_gradient(device):
    from tensorflow_ocr_amd import layers
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as MV
    g = Graph(device, loss_scale=1.0)
    rng = np.random.default_rng(4)
    n, h, w = 2, 24, 40
    link_gt = (rng.random((n, h, w, 8)) < 0.4).astype(np.float32)
    link_pred = rng.standard_normal((n, h, w, 16)).astype(np.float32)
    W = (rng.random(n * h * w) < 0.3).astype(np.float32)
    gt_d, pr_d = torch.from_numpy(link_gt).to(device), torch.from_numpy(link_pred).to(device)
    for d in (0, 3, 7):      # tf.split slices, read in place (strided rows)
        got = MV.cal_link_loss(gt_d[..., d:d + 1], pr_d[..., 2 * d:2 * d + 2], W, graph=g).item()
        want = float(O.cal_link_loss(torch.from_numpy(link_gt[..., d]), torch.from_numpy(link_pred[..., 2 * d:2 * d + 2]),
                                     torch.from_numpy(W)))
        assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (d, got, want)
    # a head handle: the backward seed is recorded
    pr = torch.from_numpy(link_pred[..., 4:6].copy()).requires_grad_(True)
    want = O.cal_link_loss(torch.from_numpy(link_gt[..., 2]), pr, torch.from_numpy(W))
    want.backward()
    hd = layers.SmallAct(torch.from_numpy(link_pred[..., 4:6].copy()).to(device))
    got = MV.cal_link_loss(link_gt[..., 2], hd, W, graph=g)
    g.backward()
    assert abs(got.item() - float(want)) <= 1e-5
    assert np.abs(hd.grad.cpu().numpy() - pr.grad.numpy()).max() <= 1e-6
    # ohem_loss = 2 L_pixel + sum of the eight cal_link_loss terms with W_pixel = (label == 1) (:243-282)
    label = (rng.random((n, h, w, 1)) < 0.3).astype(np.float32)
    px = layers.SmallAct(torch.from_numpy(rng.standard_normal((n, h, w, 2)).astype(np.float32)).to(device))
    lk = layers.SmallAct(pr_d)
    g.reset_tape()
    full = MV.ohem_loss(label, px, link_gt, lk, None, graph=g)
    terms = [MV.cal_link_loss(gt_d[..., d:d + 1], pr_d[..., 2 * d:2 * d + 2], label.reshape(-1), graph=g).item() for d in range(8)]
    assert np.allclose(full.terms()[1:9], terms, rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        MV.cal_link_loss(gt_d[..., 0:1], pr_d[..., 0:2], W[:-1], graph=g)


def test_tf_pixellink_get_rbox(device):
    from oracle import labels as OL
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as PF
    g = Graph(device)
    rng = np.random.default_rng(5)
    h, w, k = 128, 192, 6
    cx, cy = rng.uniform(0.15, 0.85, k), rng.uniform(0.15, 0.85, k)
    dx, dy = rng.uniform(0.03, 0.12, k), rng.uniform(0.03, 0.12, k)
    xs = np.stack([cx - dx, cx + dx, cx + dx, cx - dx], 1).astype(np.float32)
    ys = np.stack([cy - dy, cy - dy, cy + dy, cy + dy], 1).astype(np.float32)
    bboxes = np.stack([ys.min(1), xs.min(1), ys.max(1), xs.max(1)], 1).astype(np.float32)
    ignored = np.zeros(k, np.int32)
    pm, lm, sb = PF.tf_pixellink_get_rbox((h, w), xs, ys, bboxes, ignored, graph=g)
    assert tuple(pm.shape) == (h // 4, w // 4) and tuple(lm.shape) == (h // 4, w // 4, 8) and tuple(sb.shape) == (200, 4)
    s_ref, l_ref, sb_ref = OL.pixellink_generate_rbox(h, w, xs, ys, bboxes, ignored)
    assert np.array_equal(pm.cpu().numpy(), s_ref) and np.array_equal(lm.cpu().numpy(), l_ref)
    assert np.array_equal(sb.cpu().numpy(), sb_ref)
