"""GPU: what the PRODUCT path (16-bit storage, MFMA kernels) delivers on a score map, with a bar that
would catch a regression.

The north star's "score-map L-inf < 1e-3" is a statement about f32 arithmetic; the library's f32 inference
precision (Graph(precision="f32"): matrix-core f32 convolutions, tests/test_gpu_f32_verify.py, test_gpu_f32_mfma.py)
meets it and is measured here beside the default.  The default path stores activations
in f16 / bf16, so its score maps carry the accumulated storage rounding of 16 layers.  This file
measures that number in INFERENCE mode (`is_training=False`: the heads' batch norms use their moving
statistics; the VGG trunk's stay in batch mode as in the reference) on the softmax
score maps the decode consumes, against the f32 oracle, and asserts it at about twice the measured
value — not at the 1e-1 of the chaotic training-mode end-to-end test.

Measured on MI355X (this file's prints; L-inf / mean of |P(text) device - P(text) f32 oracle|):
see MEASURED below, updated with the round's GPU runs."""
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

# (size, n) -> (pixel-score Linf bar, link-score Linf bar, mean bar); f16 and bf16 builds
# measured (MI355X, round 2): f16 64^2 n=2  P(text) 5.2e-3 / P(link) 2.3e-2 (means 0.9e-3 / 1.5e-3);
# f16 512^2 n=1  1.3e-2 / 3.8e-2 (means 1.0e-3 / 1.5e-3); bf16 64^2  5.1e-2 / 1.8e-1 (means 6.6e-3 / 1.0e-2)
BARS_F16 = {(64, 2): (1.2e-2, 5e-2, 3e-3), (512, 1): (2.5e-2, 8e-2, 3e-3)}
BARS_BF16 = {(64, 2): (1e-1, 3.5e-1, 2e-2), (512, 1): (2e-1, 5e-1, 2e-2)}


def _params(rng):
    p = O.init_model_vgg_params(rng)
    for k in p:                         # non-trivial moving statistics for the inference-mode head BNs
        if k.endswith('moving_mean'):
            p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
        if k.endswith('moving_variance'):
            p[k] = rng.uniform(0.5, 1.5, p[k].shape).astype(np.float32)
    return p


def _scores(px, lk):
    """P(text) and the 8 P(link) maps from the 2- and 16-channel logits (test.py:142-147)."""
    px, lk = torch.as_tensor(px), torch.as_tensor(lk)
    s = torch.softmax(px, -1)[..., 1]
    l = torch.softmax(lk.reshape(lk.shape[:-1] + (8, 2)), -1)[..., 1]
    return s.numpy(), l.numpy()


@pytest.mark.parametrize("size,n", [(64, 2), (512, 1)])
def test_inference_score_maps_vs_f32_oracle(device, size, n):
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    rng = np.random.default_rng(0)
    p = _params(rng)
    images, _, _, _ = O.synthetic_batch(rng, n, size)
    g = Graph(device)
    M.model_vgg(images[:, :64, :64], is_training=False, graph=g)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    px, lk = M.model_vgg(images, is_training=False, graph=g)
    g.reset_tape()
    torch.cuda.synchronize()
    with torch.no_grad():
        fpx, flk, _ = O.model_vgg(torch.from_numpy(images), O.to_torch_params(p, requires_grad=False), False, mixed=False)
    ds, dl = _scores(px.data.cpu(), lk.data.cpu())
    fs, fl = _scores(fpx, flk)
    e_s, e_l = np.abs(ds - fs), np.abs(dl - fl)
    el = np.abs(px.data.cpu().numpy() - fpx.numpy()).max()
    print("%s %dx%d n=%d: P(text) Linf %.3e mean %.3e | P(link) Linf %.3e mean %.3e | logits Linf %.3e (range %.2f)" % (
        "bf16" if BF else "f16", size, size, n, e_s.max(), e_s.mean(), e_l.max(), e_l.mean(), el, float(fpx.abs().max())))
    bs, bl, bm = (BARS_BF16 if BF else BARS_F16)[(size, n)]
    assert e_s.max() < bs and e_l.max() < bl and e_s.mean() < bm and e_l.mean() < bm
    assert fs.std() > 1e-3                    # the maps are not degenerate
    # ... and side by side, the SAME library's f32 inference precision (Graph(precision="f32"): f32 storage, convolutions
    # on the matrix cores with v_mfma_f32_32x32x2_f32 — test.py --precision f32), which meets the north star's 1e-3
    g32 = Graph(device, precision="f32")
    M.model_vgg(images[:, :64, :64], is_training=False, graph=g32)
    g32.reset_tape()
    g32.store.load_state_dict(checkpoint.tf_to_internal(g32.store.order, p))
    qx, qk = M.model_vgg(images, is_training=False, graph=g32)
    g32.reset_tape()
    torch.cuda.synchronize()
    hs, hl = _scores(qx.data.cpu(), qk.data.cpu())
    h_s, h_l = np.abs(hs - fs).max(), np.abs(hl - fl).max()
    hlg = max(np.abs(qx.data.cpu().numpy() - fpx.numpy()).max(), np.abs(qk.data.cpu().numpy() - flk.numpy()).max())
    print("f32 precision, same library: P(text) Linf %.3e | P(link) Linf %.3e | logits Linf %.3e" % (h_s, h_l, hlg))
    assert max(h_s, h_l, hlg) < 1e-3


def test_inference_score_maps_bf16_build(device):
    """The same measurement in the bfloat16 build (libocr_hip_bf16.so), in a child interpreter."""
    if BF:
        pytest.skip("already the bf16 child")
    env = dict(os.environ, OCR_STORAGE="bf16")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_product_accuracy.py"), "-k", "vs_f32_oracle"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("bf16")))
    assert r.returncode == 0 and "2 passed" in r.stdout, tail
