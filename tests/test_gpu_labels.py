"""GPU: ground-truth label maps and image resize vs the oracle's sequential
restatement (oracle/labels.py on oracle/cvgeom_oracle.c) — bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import cvgeom as C
from oracle import labels as OL

pytestmark = pytest.mark.gpu


def _quads(rng, k, size, lo=0, hi=None, small=False):
    hi = size - 1 if hi is None else hi
    out = []
    for _ in range(k):
        c = rng.uniform(lo + 5, hi - 5, 2)
        w, h = (rng.uniform(3, 9), rng.uniform(2, 6)) if small and rng.uniform() < 0.5 else \
            (rng.uniform(10, size / 3), rng.uniform(6, size / 8))
        th = rng.uniform(-0.6, 0.6)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        p = (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) / 2) @ R.T + c
        out.append(np.clip(p, lo, hi))
    return np.array(out, np.float32).reshape(-1, 4, 2)


@pytest.mark.parametrize("size,k,seed", [(64, 5, 0), (128, 12, 1), (96, 40, 2), (512, 9, 3)])
def test_icdar_labels_bit_exact(device, size, k, seed):
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    rng = np.random.default_rng(seed)
    polys_l, tags_l = [], []
    for b in range(3):
        kk = k if b < 2 else 0                          # last image: no polygons at all
        polys = _quads(rng, kk, size, small=True)
        if b == 1 and kk:
            polys[0] = [[0, 0], [size - 1, 0], [size - 1, 7], [0, 7]]            # touches three borders
            polys[1] = [[size - 9, 3], [size - 1, 3], [size - 1, size - 1], [size - 9, size - 1]]
            polys[2, :, 1] = polys[2, 0, 1]                                        # degenerate: one row
        polys_l.append(polys)
        tags_l.append(rng.uniform(size=kk) < 0.3)
    score, geo, mask = icdar.generate_rbox_batch((size, size), polys_l, tags_l, graph=g)
    for b in range(3):
        os_, og, om = OL.icdar_labels((size, size), polys_l[b], tags_l[b])
        assert np.array_equal(score[b].cpu().numpy(), os_)
        assert np.array_equal(geo[b].cpu().numpy(), og)
        assert np.array_equal(mask[b].cpu().numpy(), om)
    if size <= 128:      # full-resolution single-image API (the reference's generate_rbox signature)
        s, gm, m = icdar.generate_rbox((size, size), polys_l[1], tags_l[1], graph=g)
        os_, og, om = OL.icdar_generate_rbox((size, size), polys_l[1], tags_l[1])
        assert s.dtype == np.uint8 and np.array_equal(s, os_)
        assert np.array_equal(gm, og) and np.array_equal(m, om)
        assert gm.sum() > 0 and s.sum() > 0 and (m == 0).any()


@pytest.mark.parametrize("h,w,k,seed", [(64, 64, 4, 0), (128, 96, 10, 1), (100, 140, 7, 2)])
def test_pixellink_generate_rbox_bit_exact(device, h, w, k, seed):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g = Graph(device)
    rng = np.random.default_rng(seed)
    xs_l, ys_l, bb_l, ig_l = [], [], [], []
    for b in range(2):
        q = _quads(rng, k, 1000, lo=-30, hi=1030) / 1000.0          # normalised, partly outside [0,1]
        if b == 0:
            q[0] = [[0.0, 0.0], [1.0, 0.0], [1.0, 0.2], [0.0, 0.2]]  # x == w after scaling: clipLine
        xs_l.append(q[:, :, 0])
        ys_l.append(q[:, :, 1])
        bb_l.append(rng.uniform(size=(k, 4)).astype(np.float32))
        ig_l.append(np.zeros(k, np.int32))
    score, link, show = P.generate_rbox_batch(h, w, xs_l, ys_l, bb_l, ig_l, graph=g)
    for b in range(2):
        os_, ol, ob = OL.pixellink_generate_rbox(h, w, xs_l[b], ys_l[b], bb_l[b], ig_l[b])
        assert np.array_equal(score[b].cpu().numpy(), os_)
        assert np.array_equal(link[b].cpu().numpy(), ol)
        assert np.array_equal(show[b], ob)
    s1, l1, b1 = P.generate_rbox(h, w, xs_l[0], ys_l[0], bb_l[0], ig_l[0], graph=g)
    assert np.array_equal(s1, score[0].cpu().numpy()) and l1.shape == (h // 4, w // 4, 8)


def test_poly_cover_random_polygons_vs_fillpoly(device):
    """Raw cover map against the literal fillPoly on arbitrary (concave, self-intersecting, degenerate,
    partly outside) polygons with 3..8 vertices."""
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(7)
    for V, h, w in [(3, 40, 56), (4, 64, 64), (5, 33, 70), (8, 80, 48)]:
        n, P = 4, 37
        polys = rng.integers(-12, max(h, w) + 12, size=(n, P, V, 2)).astype(np.int32)
        polys[0] = np.clip(polys[0], 0, min(h, w) - 1)
        polys[1, :, 1] = polys[1, :, 0]                    # two identical vertices
        polys[2, 5, :, 1] = 9                              # horizontal degenerate
        counts = np.array([P, P, 11, 0], np.int32)
        ignore = (rng.uniform(size=(n, P)) < 0.25).astype(np.uint8)
        cover = torch.empty((n, h, w), dtype=torch.int32, device=device)
        ops.poly_cover(torch.from_numpy(polys).to(device), torch.from_numpy(counts).to(device),
                       torch.from_numpy(ignore).to(device), h, w, cover)
        cover = cover.cpu().numpy()
        for b in range(n):
            first = np.zeros((h, w), np.int64)
            last = np.zeros((h, w), np.int64)
            ign = np.zeros((h, w), np.int64)
            for i in range(counts[b]):
                m = C.fill_poly(np.zeros((h, w), np.uint8), polys[b, i], 1).astype(bool)
                first[m & (first == 0)] = i + 1
                last[m] = i + 1
                if ignore[b, i]:
                    ign[m] = 1
            assert np.array_equal(cover[b] & 255, first)
            assert np.array_equal((cover[b] >> 8) & 255, last)
            assert np.array_equal((cover[b] >> 16) & 1, ign)


@pytest.mark.parametrize("H,W,S", [(37, 53, 64), (720, 1280, 512), (1024, 1024, 512), (300, 200, 512), (64, 64, 64)])
def test_resize_linear_u8_bit_exact(device, H, W, S):
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    rng = np.random.default_rng(H)
    im = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    out = icdar.resize_images([im], S, graph=g)[0].cpu().numpy()
    want = C.resize_linear_u8(im, S, S).astype(np.float32)
    assert out.dtype == np.float32 and np.array_equal(out, want)


def test_generator_end_to_end(device, tmp_path):
    """icdar.generator on a directory of images + gt_*.txt (PNG via PIL and .npy), against the oracle
    chain: validate -> scale polygons -> resize -> labels."""
    from PIL import Image
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    rng = np.random.default_rng(3)
    S = 128
    names = []
    for i, (H, W) in enumerate([(90, 160), (128, 128), (200, 150)]):
        im = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        name = "img_%d" % i
        if i == 1:
            np.save(os.path.join(tmp_path, name + ".npy"), im)
        else:
            Image.fromarray(im).save(os.path.join(tmp_path, name + ".png"))
        q = _quads(rng, 5, min(H, W))
        q[1] = q[1][::-1]                                   # wrong orientation: flipped by validation
        q[2] = q[2, 0]                                      # zero area: dropped
        with open(os.path.join(tmp_path, "gt_%s.txt" % name), "w", encoding="utf-8") as f:
            if i == 0:
                f.write("\ufeff")
            for j, p in enumerate(q):
                f.write(",".join("%d" % v for v in p.ravel()) + "," + ("###" if j == 3 else "word,with,commas") + "\n")
        names.append(name)
    gen = icdar.generator(str(tmp_path), input_size=S, batch_size=3, graph=g, shuffle=False)
    images, fns, score, geo, mask = next(gen)
    assert images.shape == (3, S, S, 3) and score.shape == (3, S // 4, S // 4, 1) and geo.shape[-1] == 8
    for b, fn in enumerate(fns):
        im = icdar.read_image_rgb(fn)
        H, W, _ = im.shape
        polys, tags = icdar.load_annoataion(icdar.txt_name(fn))
        assert len(polys) == 5 and tags.tolist() == [False, False, False, True, False]
        polys, tags = icdar.check_and_validate_polys(polys, tags, (H, W))
        assert len(polys) == 4
        polys[:, :, 0] *= S / float(W)
        polys[:, :, 1] *= S / float(H)
        assert np.array_equal(images[b].cpu().numpy(), C.resize_linear_u8(im, S, S).astype(np.float32))
        os_, og, om = OL.icdar_labels((S, S), polys, tags)
        assert np.array_equal(score[b].cpu().numpy(), os_) and np.array_equal(geo[b].cpu().numpy(), og)
        assert np.array_equal(mask[b].cpu().numpy(), om)


def test_device_feeder_overlaps_and_matches_direct_generator(device, tmp_path):
    """icdar.get_batch (feeder thread + own HIP stream + 2 decode processes) yields exactly what the
    in-line generator yields, batch after batch, and a recorded training step is not polluted by the
    feeder's launches."""
    from tensorflow_ocr_amd.datasets import icdar
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    rng = np.random.default_rng(11)
    S, B = 64, 4
    for i in range(10):
        H, W = int(rng.integers(60, 140)), int(rng.integers(60, 140))
        np.save(os.path.join(tmp_path, "im%02d.npy" % i), rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8))
        with open(os.path.join(tmp_path, "gt_im%02d.txt" % i), "w") as f:
            for p in _quads(rng, 3, min(H, W)):
                f.write(",".join("%d" % v for v in p.ravel()) + ",w\n")
    direct = icdar.generator(str(tmp_path), input_size=S, batch_size=B, graph=g, shuffle=True, seed=5)
    feeder = icdar.get_batch(num_workers=2, training_data_path=str(tmp_path), input_size=S, batch_size=B,
                             graph=g, shuffle=True, seed=5)
    try:
        for _ in range(5):                      # crosses an epoch boundary (10 images, 2 batches each)
            a = next(direct)
            b = next(feeder)
            assert a[1] == b[1]
            for x, y in zip((a[0], a[2], a[3], a[4]), (b[0], b[2], b[3], b[4])):
                assert torch.equal(x, y)
    finally:
        feeder.close()
