"""GPU: the full-resolution decode of test_pixellink.py (:95-218) — cv2.resize(INTER_CUBIC) of the
score maps bit-exact against the CPU restatement (oracle/cvgeom_oracle.c, parity unpinned: cv2 is
not installed), then the link-gated grouping on the up-sampled maps against the union-find oracle."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from oracle import cvgeom
from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape,out", [((192, 320), (720, 1280)), ((48, 80), (180, 320)), ((37, 53), (41, 200)),
                                       ((64, 64), (32, 32)), ((50, 70), (17, 23)), ((1, 5), (4, 20)),
                                       ((5, 1), (20, 3)), ((2, 3), (9, 7)), ((33, 33), (33, 33))])
def test_resize_cubic_f32_bit_exact(device, shape, out):
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(shape[0] * 1000 + out[1])
    src = rng.uniform(0, 1, size=(3,) + shape).astype(np.float32)
    src[1] = (src[1] > 0.5).astype(np.float32)              # saturated maps: over/undershoot of the cubic
    d_src = torch.from_numpy(src).to(device)
    for pre, post in ((1.0, 1.0), (255.0, 1.0), (1.0, 255.0)):
        dst = torch.full((3,) + out, float("nan"), dtype=torch.float32, device=device)
        ops.resize_cubic_f32(d_src, dst, pre, post)
        got = dst.cpu().numpy()
        for k in range(3):
            want = cvgeom.resize_cubic_f32(src[k] * np.float32(pre), out[0], out[1]) * np.float32(post)
            assert np.array_equal(got[k], want), (k, pre, post, np.abs(got[k] - want).max())


def test_resize_cubic_properties(device):
    """Size-independent properties at the script's full size: identity when the size does not change,
    constants preserved to rounding (the four coefficients sum to 1), separability in the plane index."""
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(3)
    src = torch.from_numpy(rng.uniform(size=(9, 192, 320)).astype(np.float32)).to(device)
    same = torch.empty_like(src)
    ops.resize_cubic_f32(src, same)
    assert torch.equal(same, src)
    const = torch.full((2, 192, 320), 0.625, dtype=torch.float32, device=device)
    up = torch.empty((2, 720, 1280), dtype=torch.float32, device=device)
    ops.resize_cubic_f32(const, up)
    assert float((up - 0.625).abs().max()) < 1e-6
    all9 = torch.empty((9, 720, 1280), dtype=torch.float32, device=device)
    ops.resize_cubic_f32(src, all9)
    one = torch.empty((1, 720, 1280), dtype=torch.float32, device=device)
    ops.resize_cubic_f32(src[4:5], one)
    assert torch.equal(one[0], all9[4])
    assert float(all9.min()) < 0.0 and float(all9.max()) > 1.0        # cubic overshoot exists (A = -0.75)


def _score_maps(rng, n, h, w):
    """Blobby text-like maps: rectangles of high pixel score with mostly-on links inside."""
    ps = rng.uniform(0.0, 0.6, size=(n, h, w)).astype(np.float32)
    lk = rng.uniform(0.0, 0.7, size=(8, n, h, w, 2)).astype(np.float32)
    for i in range(n):
        for _ in range(5):
            y0, x0 = int(rng.integers(1, h - 8)), int(rng.integers(1, w - 12))
            hh, ww = int(rng.integers(3, 7)), int(rng.integers(5, 11))
            ps[i, y0:y0 + hh, x0:x0 + ww] = rng.uniform(0.85, 1.0, size=(hh, ww))
            lk[:, i, y0:y0 + hh, x0:x0 + ww, 1] = rng.uniform(0.8, 1.0, size=(8, hh, ww))
    return ps, lk


def test_full_resolution_decode_matches_oracle(device):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as P
    g = Graph(device)
    rng = np.random.default_rng(11)
    n, h, w, oh, ow = 2, 24, 40, 90, 160
    ps, lk = _score_maps(rng, n, h, w)
    up = P.resize_scores_cubic(ps, lk, oh, ow, graph=g).cpu().numpy()
    for i in range(n):
        want_p = cvgeom.resize_cubic_f32(ps[i], oh, ow) * np.float32(255)
        assert np.array_equal(up[0, i], want_p)
        for d in range(8):
            assert np.array_equal(up[1 + d, i], cvgeom.resize_cubic_f32(lk[d, i, :, :, 1] * np.float32(255), oh, ow))
    labels, ncomp, comps = P.full_resolution_decode(ps, lk, oh, ow, min_size=20, graph=g)
    labels, ncomp, comps = labels.cpu().numpy(), ncomp.cpu().numpy(), comps.cpu().numpy()
    total = 0
    for i in range(n):
        want, wc = O.link_cc_union(up[0, i], [up[1 + d, i] for d in range(8)], float(int(255 * 0.8)), 255 * 0.9,
                                   min_size=20)
        assert np.array_equal(labels[i], want)
        assert ncomp[i] == len(wc)
        assert [tuple(c) for c in comps[i, :len(wc)]] == wc
        total += len(wc)
    assert total >= 4
    # one box per group, in full-resolution coordinates (scale 1): every group pixel inside its box's hull
    res = P.min_area_rect_boxes(torch.from_numpy(labels).to(device), torch.from_numpy(ncomp).to(device), 1.0, 1.0,
                                graph=g)
    for i in range(n):
        rects, boxes = res[i]
        assert len(boxes) == ncomp[i]
        for k, r in enumerate(rects):
            ys, xs = np.nonzero(labels[i] == k + 1)
            assert abs(r[0] - (xs.min() + xs.max()) / 2) < max(r[2], r[3]) and abs(r[1] - (ys.min() + ys.max()) / 2) < max(r[2], r[3])


@pytest.mark.parametrize("script", ["test_pixellink", "test_pixellink_fast"])
def test_decode_scripts_end_to_end(device, tmp_path, capsys, script):
    """Three images of one size: the forward runs launch by launch, is captured as a HIP graph on the
    second image and replayed on the third."""
    sys.path.insert(0, ROOT)
    mod = importlib.import_module(script)
    assert mod.__file__.startswith(ROOT)
    out_dir = os.path.join(tmp_path, "out")
    old = sys.argv
    sys.argv = [script + ".py", "--synthetic", "3", "--eval_image_height", "128", "--eval_image_width", "192",
                "--checkpoint_path", os.path.join(tmp_path, "none"), "--output_dir", out_dir]
    if script == "test_pixellink":
        sys.argv += ["--decode_height", "120", "--decode_width", "192"]
    else:
        sys.argv[sys.argv.index("--checkpoint_path"):sys.argv.index("--checkpoint_path") + 2] = []
    try:
        mod.main()
    finally:
        sys.argv = old
    out = capsys.readouterr().out
    assert out.count("net+decode") == 3
    assert sorted(os.listdir(out_dir)) == ["res_synthetic_%d.txt" % i for i in range(3)]
    for fn in os.listdir(out_dir):
        for line in open(os.path.join(out_dir, fn), newline="").read().split("\r\n"):
            assert line == "" or len(line.split(",")) == 8
