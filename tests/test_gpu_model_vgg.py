"""GPU parity, end to end: model_vgg forward / dice loss / backward (through the C ABI) vs the CPU
oracle on the same seeded inputs.

Tolerances.  The device pipeline stores activations and activation gradients in f16 (BASELINE
config "fp16").  A 15-conv BN+ReLU network at random init amplifies one-ulp f16 differences
chaotically: the ORACLE ITSELF moves by ~5e-2 (score maps, L-inf) and ~0.2 relative-L2 / cosine
0.97 (weight gradients) between its f32 and its f16-storage (`mixed`) modes on these inputs
(measured, see DESIGN.md "Parity").  The end-to-end bars below are therefore set at that
intrinsic sensitivity; the TIGHT bars (2e-3 outputs, 1e-2 gradients, identical storage roundings)
are the single-layer tests in test_gpu_layers.py."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu

S = 1024.0


def _run_oracle(p, images, pixel, link, mask, mixed):
    tp = O.to_torch_params(p)
    px, lk, ep = O.model_vgg(torch.from_numpy(images), tp, True, mixed=mixed)
    L = O.dice_loss(torch.from_numpy(pixel), px, torch.from_numpy(link), lk, torch.from_numpy(mask))
    (L * S).backward()
    grads = {k: (v.grad / S).numpy() for k, v in tp.items() if v.grad is not None}
    return px.detach().numpy(), lk.detach().numpy(), float(L), grads, ep


def _run_device(device, p, images, pixel, link, mask):
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    g = Graph(device, loss_scale=S)
    M.model_vgg(images, graph=g)          # creates the variables
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))   # incl. moving stats reset
    px, lk = M.model_vgg(images, graph=g)
    L = M.loss(pixel, px, link, lk, mask, graph=g)
    g.backward()
    torch.cuda.synchronize()
    grads = {n: (v.grad / S).cpu().numpy() for n, v in g.store.vars.items() if v.trainable}
    grads = checkpoint.internal_to_tf(grads)
    return px.data.cpu().numpy(), lk.data.cpu().numpy(), L.item(), grads, g


@pytest.mark.parametrize("size,n", [(64, 2), (96, 1)])
def test_model_vgg_forward_backward(device, size, n):
    rng = np.random.default_rng(0)
    p = O.init_model_vgg_params(rng)
    images, pixel, link, mask = O.synthetic_batch(rng, n, size)
    dpx, dlk, dL, dgr, g = _run_device(device, p, images, pixel, link, mask)
    opx, olk, oL, ogr, _ = _run_oracle(p, images, pixel, link, mask, mixed=True)
    fpx, flk, fL, fgr, _ = _run_oracle(p, images, pixel, link, mask, mixed=False)
    print("loss device %.6f oracle(mixed) %.6f oracle(f32) %.6f" % (dL, oL, fL))
    print("pixel_cls Linf vs mixed %.3e vs f32 %.3e" % (np.abs(dpx - opx).max(), np.abs(dpx - fpx).max()))
    print("link_cls  Linf vs mixed %.3e vs f32 %.3e" % (np.abs(dlk - olk).max(), np.abs(dlk - flk).max()))
    # f16 storage: the bars documented in DESIGN.md section 4.  bfloat16 storage (OCR_STORAGE=bf16) rounds 8x
    # coarser at each of the ~30 storage points; there the device must stay closer to the oracle in the
    # SAME storage mode than that oracle is to its own f32 mode (the net's intrinsic sensitivity).
    bf = O.STORAGE == torch.bfloat16
    if bf:
        assert np.abs(dpx - opx).max() < np.abs(opx - fpx).max() and np.abs(dlk - olk).max() < np.abs(olk - flk).max()
        # loss (~6.5 here): within 4e-2 of the bf16 oracle, or within the band between that oracle and its f32 mode
        assert np.abs(dpx - opx).mean() < 8e-2
        assert abs(dL - oL) < 4e-2 or abs(dL - fL) + abs(dL - oL) < 1.5 * abs(oL - fL)
    else:
        assert np.abs(dpx - opx).max() < 1e-1 * max(1.0, np.abs(opx).max())
        assert np.abs(dlk - olk).max() < 1e-1 * max(1.0, np.abs(olk).max())
        assert np.abs(dpx - opx).mean() < 1e-2
        assert abs(dL - oL) < 5e-3 and abs(dL - fL) < 2e-2

    def cos(a, b):
        a, b = a.ravel().astype(np.float64), b.ravel().astype(np.float64)
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    worst = 1.0
    allc = []
    for k in sorted(ogr):
        c = cos(dgr[k], ogr[k])
        allc.append((c, k))
        if ogr[k].size >= 64:            # 2x2 / 16-element tensors are too small for a stable cosine
            worst = min(worst, c)
    print("lowest cosines:", sorted(allc)[:4])
    da = np.concatenate([dgr[k].ravel() for k in sorted(ogr)])
    oa = np.concatenate([ogr[k].ravel() for k in sorted(ogr)])
    print("global gradient cosine vs mixed oracle %.4f" % cos(da, oa))
    assert worst > (0.6 if bf else 0.9)
    assert cos(da, oa) > (0.8 if bf else 0.95)
