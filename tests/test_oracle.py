"""CPU tests of the oracle: the reference's one known-answer value, the TF-1.4 semantics of
SURVEY.md §3.5 stated as hand-computed cases, and the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_softmax_known_answer_from_reference_example_py():
    # example.py:13-21: softmax over [[1,2],[3,4],[5,6],[7,8]] pairs -> [0.268941, 0.731059]
    x = torch.tensor([[[[1., 2.], [3., 4.], [5., 6.], [7., 8.]]]])
    s = O.softmax(x).numpy().reshape(-1, 2)
    assert np.allclose(s, [[0.26894142, 0.73105858]] * 4, atol=1e-6)


def test_same_padding_is_asymmetric_like_tf():
    assert O.tf_same_pad(512, 3, 1) == (512, 1, 1)
    assert O.tf_same_pad(512, 2, 2) == (256, 0, 0)
    assert O.tf_same_pad(320, 3, 2) == (160, 0, 1)      # 3x3/2 on even H pads (0,1)
    assert O.tf_same_pad(7, 2, 2) == (4, 0, 1)
    assert O.tf_same_pad(32, 3, 1, rate=6) == (32, 6, 6)


def test_conv2d_same_uses_explicit_symmetric_pad_for_stride2():
    # resnet_utils.conv2d_same: 3x3/2 pads (1,1), unlike raw SAME's (0,1)
    x = torch.arange(16.0).reshape(1, 4, 4, 1)
    w = torch.ones(3, 3, 1, 1)
    a = O.conv2d_same(x, w, 2).numpy()[0, :, :, 0]
    b = O.conv2d(x, w, 2, 1, "SAME").numpy()[0, :, :, 0]
    assert a[0, 0] == 0 + 1 + 4 + 5            # window centred on pixel (0,0) with zero halo
    assert b[0, 0] == sum([0, 1, 2, 4, 5, 6, 8, 9, 10])   # raw SAME window starts at (0,0)


def test_legacy_bilinear_x2():
    x = torch.tensor([1., 3., 7.]).reshape(1, 1, 3, 1)
    up = O.resize_bilinear_x2(x).numpy()[0, :, :, 0]
    assert np.allclose(up[0], [1, 2, 3, 5, 7, 7])      # out[2i]=in[i], out[2i+1]=(in[i]+in[min(i+1,W-1)])/2
    assert np.allclose(up[1], up[0])                    # H=1: the second row clamps


def test_max_pool_gradient_goes_to_first_maximum():
    x = torch.tensor([[2., 2.], [2., 1.]]).reshape(1, 2, 2, 1).requires_grad_(True)
    O.max_pool(x, 2, 2).sum().backward()
    assert np.array_equal(x.grad.numpy().reshape(2, 2), [[1, 0], [0, 0]])


def test_batch_norm_moving_variance_is_unbiased():
    x = torch.tensor([[0.], [2.]]).reshape(2, 1, 1, 1)
    y, mm, mv = O.batch_norm(x, torch.ones(1), torch.zeros(1), torch.zeros(1), torch.ones(1), True,
                             decay=0.5, eps=0.0)
    assert np.allclose(y.numpy().ravel(), [-1, 1])           # biased variance 1 normalises
    assert np.allclose(mm.numpy(), [0.5]) and np.allclose(mv.numpy(), [0.5 * 1 + 0.5 * 2.0])


def test_dice_broadcasting_matches_hand_sum():
    y = torch.tensor([1., 0.]).reshape(1, 1, 2, 1)
    p = torch.tensor([[.5, .25], [.5, .5]]).reshape(1, 1, 2, 2)
    m = torch.ones(1, 1, 2, 1)
    d = O.dice_coefficient(y, p, m).item()
    assert abs(d - (1 - 2 * 0.75 / (1 + 1.75 + 1e-5))) < 1e-6


def test_adam_and_ema_formulas():
    w, m, v = O.adam_update(np.float32(1.0), np.float32(0.5), 0.0, 0.0, 1, 1e-4)
    # first step: m = 0.05, v = 2.5e-4, lr_t = lr*sqrt(1-.999)/(1-.9) -> update = lr (up to eps)
    assert abs(w - (1.0 - 1e-4)) < 2e-7 and abs(m - 0.05) < 1e-8 and abs(v - 2.5e-4) < 1e-9
    assert O.ema_decay(0.997, 0) == 0.1 and O.ema_decay(0.997, 10 ** 6) == 0.997
    assert abs(O.exponential_decay(1e-4, 5000) - 0.94e-4) < 1e-12
    assert O.exponential_decay(1e-4, 4999) == 1e-4


def test_golden_primitives():
    g = np.load(os.path.join(GOLD, "primitives.npz"))
    t = torch.from_numpy
    assert np.allclose(O.resize_bilinear_x2(t(g["x"])).numpy(), g["up"], atol=1e-6)
    assert np.array_equal(O.max_pool(t(g["xp"]), 2, 2).numpy(), g["p22"])
    assert np.array_equal(O.max_pool(t(g["xp"]), 3, 1).numpy(), g["p31"])
    assert np.array_equal(O.max_pool(t(g["xp"]), 3, 2).numpy(), g["p32"])
    assert np.allclose(O.conv2d(t(g["xp"]), t(g["w"]), 1, 1).numpy(), g["c1"], atol=1e-5)
    assert np.allclose(O.conv2d(t(g["xp"]), t(g["w"]), 1, 6).numpy(), g["c6"], atol=1e-5)
    assert np.allclose(O.conv2d_same(t(g["xp"]), t(g["w"]), 2).numpy(), g["cs2"], atol=1e-5)
    d = O.dice_loss(t(g["yt"]), t(g["pp"]), t(g["yl"]), t(g["pl"]), t(g["m"])).item()
    assert abs(d - float(g["dice"])) < 1e-5


def test_golden_model_vgg_small():
    g = np.load(os.path.join(GOLD, "model_vgg_w8_64.npz"))
    rng = np.random.default_rng(7)
    p = O.init_model_vgg_params(rng, width_div=8)
    images, pixel, link, mask = O.synthetic_batch(rng, 1, 64)
    assert np.array_equal(images, g["images"]) and np.array_equal(link, g["link"])
    tp = O.to_torch_params(p)
    px, lk, _ = O.model_vgg(torch.from_numpy(images), tp, True, mixed=False)
    L = O.dice_loss(torch.from_numpy(pixel), px, torch.from_numpy(link), lk, torch.from_numpy(mask))
    assert np.allclose(px.detach().numpy(), g["pixel_cls"], atol=2e-4)
    assert np.allclose(lk.detach().numpy(), g["link_cls"], atol=2e-4)
    assert abs(L.item() - float(g["loss"])) < 1e-4


def test_momentum_update_and_pixellink_lr():
    """tf.train.MomentumOptimizer (train_pixellink.py:243) and the tf.case LR factors (:222-237)."""
    w, g, acc = np.array([1.0, -2.0]), np.array([0.5, 0.25]), np.array([0.1, -0.2])
    w1, a1 = O.momentum_update(w, g, acc, lr=0.1, momentum=0.9)
    assert np.allclose(a1, [0.59, 0.07]) and np.allclose(w1, [1.0 - 0.059, -2.0 - 0.007])
    assert [O.pixellink_lr(s) for s in (0, 19999, 20000, 40000, 60000)] == pytest.approx([1e-3, 1e-3, 1e-4, 1e-5, 1e-2])


def test_det_exp_is_an_accurate_f32_exp_and_scores_are_a_softmax():
    x = np.linspace(-30, 30, 100001).astype(np.float32)
    e = O.det_exp_f32(x)
    assert e.dtype == np.float32
    assert np.max(np.abs(e - np.exp(x.astype(np.float64))) / np.exp(x.astype(np.float64))) < 4e-7
    l = (np.random.default_rng(0).standard_normal((4096, 2)) * 3).astype(np.float32)
    s = O.neg_score_f32(l[:, 0], l[:, 1])
    assert np.abs(s - torch.softmax(torch.from_numpy(l), -1)[:, 0].numpy()).max() < 3e-7
    # example.py:13-21 (the reference's only known answer): softmax([1, 2])[0] = 0.268941
    assert abs(float(O.neg_score_f32(np.float32(1), np.float32(2))) - 0.268941) < 1e-6


def test_link_cc_union_vs_reference_dfs_on_asymmetric_links():
    """a15: the product's weakly-connected components against the reference's directed DFS
    (test_pixellink_fast.py:153-178, ascending key order) on ASYMMETRIC link predictions.  They differ
    only in fragments around the size filter (a pixel the DFS reaches from one side but not the
    other); pixels labelled by both always fall in the same groups.  Measured on 4 x 128^2 maps:
    0 / 19823 segment pixels at the survey's logit strength 3.0, 17 / 19226 (0.09 %) at 2.0,
    45 / 17367 (0.26 %) at 1.5."""
    def sm(l):
        e = np.exp(l - l.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)
    for strength, bound in ((3.0, 0.0), (1.5, 0.01)):          # 12 / 2132 on these two 64^2 maps
        rng = np.random.default_rng(0)
        pl, ll = O.synthetic_decode_maps(rng, 2, 64, strength)
        tot = differs = 0
        for b in range(2):
            ps = sm(pl[b])[..., 1]
            ls = [sm(ll[b][..., 2 * d:2 * d + 2])[..., 1] for d in range(8)]
            A = O.link_cc_reference_dfs(ps, ls, 0.8, 0.9, 10)
            B, _ = O.link_cc_union(ps, ls, 0.8, 0.9, 10)
            seg = ps > 0.8
            tot += int(seg.sum())
            differs += int(((A > 0) != (B > 0))[seg].sum())
            both = (A > 0) & (B > 0)
            for a in np.unique(A[both]):                 # every DFS group lies inside ONE union-find group
                assert len(np.unique(B[both & (A == a)])) == 1
        assert differs <= bound * tot, (strength, differs, tot)


def test_directed_rounds_schedule_equals_the_literal_dfs():
    """The kernel's schedule for the reference's directed grouping (rounds of smallest-seed reachability per
    weakly-connected component, `O.link_cc_directed_rounds`) against the literal script restatement
    (`O.link_cc_reference_dfs`) — on the survey's decode maps, on weak / asymmetric predictions where the two
    differ from the union labelling, and on near-noise maps with a small size filter where many seeds fail and
    are collected again by later ones."""
    def sm(l):
        e = np.exp(l - l.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)
    differs_from_union = 0
    for seed, q4, strength, pt, lt, ms in ((0, 48, 3.0, 0.8, 0.9, 10), (1, 48, 1.5, 0.8, 0.9, 10), (2, 40, 0.8, 0.6, 0.7, 3),
                                           (3, 40, 0.4, 0.5, 0.6, 2), (4, 32, 0.2, 0.5, 0.55, 4)):
        rng = np.random.default_rng(seed)
        pl, ll = O.synthetic_decode_maps(rng, 1, q4, strength)
        ps = sm(pl[0])[..., 1]
        ls = [sm(ll[0][..., 2 * d:2 * d + 2])[..., 1] for d in range(8)]
        for ko in ("py27", "ascending"):
            A = O.link_cc_reference_dfs(ps, ls, pt, lt, ms, key_order=ko)
            B = O.link_cc_directed_rounds(ps, ls, pt, lt, ms, key_order=ko)
            assert np.array_equal(A, B), (seed, ko, int((A != B).sum()))
        U, _ = O.link_cc_union(ps, ls, pt, lt, ms)
        differs_from_union += int(((A > 0) != (U > 0)).sum())
    assert differs_from_union > 0          # the cases do exercise the directed / union difference


def test_py27_dict_key_order_hand_derived():
    """the script meets its seeds in `graph.keys()` order of a CPython-2.7 dict
    (test_pixellink_fast.py:171).  Cases derived BY HAND from Objects/dictobject.c (2.7): 8-slot table, first slot
    key & 7, then i = 5 i + perturb + 1 with perturb = key, >>= 5 after each probe; growth to 32 slots when the 6th key
    lands (6 * 3 >= 8 * 2; 4 * 6 = 24 -> 32), old slots re-inserted in slot order."""
    f = O.py27_dict_key_order
    assert f([]) == [] and f([3]) == [3]
    assert f([4, 2, 0, 3, 1]) == [0, 1, 2, 3, 4]          # no collision: slot order = ascending
    assert f([8, 0]) == [8, 0]                            # 0 probes 5*0 + 0 + 1 = slot 1
    assert f([40, 8]) == [40, 8]                          # 8 probes 5*0 + 8 + 1 = 9 -> slot 1
    # 9: slot 1 taken, 5*1 + 9 + 1 = 15 -> slot 7; 17: 5*1 + 17 + 1 = 23 -> slot 7 taken, perturb 17 >> 5 = 0, 5*23 + 1 = 116 -> slot 4
    assert f([1, 9, 17]) == [1, 17, 9]
    # the 6th key triggers the rebuild at 32 slots; then 40 -> slot 8, 8 -> slot 8 taken, 5*8 + 8 + 1 = 49 -> slot 17
    assert f([0, 1, 2, 3, 4, 5, 40, 8]) == [0, 1, 2, 3, 4, 5, 40, 8]
    # six keys = 7 (mod 8): slots 7, 3, 0, 1, 6, 4 in the 8-slot table (by hand), rebuilt in slot order 23, 31, 15, 47, 39, 7
    # into 32 slots: 23, 31, 15 direct, 47 -> 15 taken -> 5*15 + 47 + 1 = 123 -> 27, 39 -> 7, 7 -> 7 taken -> 43 -> 11
    assert f([7, 15, 23, 31, 39, 47]) == [39, 7, 15, 23, 47, 31]
    # a table larger than every key holds them at slot = key: ascending, whatever the insertion order
    rng = np.random.default_rng(0)
    keys = rng.permutation(2000)[:400].tolist()           # 400 keys: rebuilt at the 342nd into 2048 slots > 1999
    assert f(keys) == sorted(keys)
    # ... and a 192 x 320 map with a tenth of its pixels set wraps (table 8192 or 32768 < 61440): not ascending
    seg = rng.random((192, 320)) < 0.1
    order = O.reference_key_order(seg.astype(np.float32), 0.5)
    assert sorted(order) != order and sorted(order) == sorted(y * 320 + x for y in range(1, 191) for x in range(1, 319) if seg[y, x])


def test_py27_dict_order_host_routine_equals_the_oracle():
    """The library's HOST routine (ocr_py27_dict_order: plain C, no GPU) against the oracle's restatement of the same
    algorithm, incl. the > 50 000-key growth rule (x2 instead of x4)."""
    import torch
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(1)
    for h, w, dens in ((20, 30, 0.5), (64, 64, 0.3), (192, 320, 0.1), (192, 320, 0.9), (256, 256, 0.95), (33, 47, 1.0), (3, 3, 1.0)):
        ps = rng.random((2, h, w)).astype(np.float32)
        thr = float(np.float32(1.0 - dens))
        got = ops.py27_dict_order(torch.from_numpy(ps), thr).numpy()
        for b in range(2):
            want = O.reference_key_order(ps[b], np.float32(thr))
            assert got[b, :len(want)].tolist() == want and (got[b, len(want):] == -1).all(), (h, w, b)


def test_reference_dfs_key_order_changes_the_grouping():
    """The two key orders give different groups on asymmetric links with the size filter (why item 6 matters): a one-way
    chain is collected whole from its head but fails piecewise from its tail."""
    q = 24
    ps = np.zeros((q, 160), np.float32)
    ls = [np.zeros((q, 160), np.float32) for _ in range(8)]
    ps[6, 2:150] = 0.9
    ls[3][6, 2:150] = 0.95                               # "right" links only
    A = O.link_cc_reference_dfs(ps, ls, 0.8, 0.9, 10, key_order="ascending")
    B = O.link_cc_reference_dfs(ps, ls, 0.8, 0.9, 10, key_order="py27")
    assert A.max() == 1 and (A[6, 2:150] == 1).all()     # ascending: the head is met first and reaches everything
    # dict order: 148 keys 962..1109 in a 512-slot table (rebuilt at the 86th key): 1024..1109 sit in slots 0..85 and are
    # met first -> x = 64.. becomes group 1, then key 962 (x = 2) collects what is left as group 2
    assert B.max() == 2 and (B[6, 64:150] == 1).all() and (B[6, 2:64] == 2).all()
    assert np.array_equal(O.link_cc_directed_rounds(ps, ls, 0.8, 0.9, 10, key_order="py27"), B)
