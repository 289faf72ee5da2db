"""Generates the golden fixtures in this directory FROM THE ORACLE (oracle/ocr_oracle.py).

The reference itself cannot run (Python 2 / TF 1.4 / cv2 absent) and ships no
vectors, so these pin the oracle against accidental change and give the GPU tests fixed
input/output pairs; they are NOT reference outputs (parity unpinned).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ocr_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def model_vgg_small():
    """Width/8 model_vgg on one 64x64 image, f32: inputs, parameters (seeded), outputs, loss, a few
    gradient norms."""
    rng = np.random.default_rng(7)
    p = O.init_model_vgg_params(rng, width_div=8)
    images, pixel, link, mask = O.synthetic_batch(rng, 1, 64)
    tp = O.to_torch_params(p)
    px, lk, _ = O.model_vgg(torch.from_numpy(images), tp, True, mixed=False)
    L = O.dice_loss(torch.from_numpy(pixel), px, torch.from_numpy(link), lk, torch.from_numpy(mask))
    L.backward()
    out = {"images": images, "pixel": pixel, "link": link, "mask": mask,
           "pixel_cls": px.detach().numpy(), "link_cls": lk.detach().numpy(),
           "loss": np.float32(L.item())}
    for k in ("conv1/conv1_1/weights", "conv5/conv5_3/weights", "fc7/BatchNorm/gamma",
              "feature_fusion/Conv_9/weights"):
        out["gradnorm:" + k] = np.float32(tp[k].grad.norm().item())
    np.savez_compressed(os.path.join(HERE, "model_vgg_w8_64.npz"), **out)


def primitives():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((1, 5, 6, 3)).astype(np.float32)
    up = O.resize_bilinear_x2(torch.from_numpy(x)).numpy()
    xp = rng.standard_normal((2, 7, 9, 4)).astype(np.float32)
    p22 = O.max_pool(torch.from_numpy(xp), 2, 2).numpy()
    p31 = O.max_pool(torch.from_numpy(xp), 3, 1).numpy()
    p32 = O.max_pool(torch.from_numpy(xp), 3, 2).numpy()
    w = rng.standard_normal((3, 3, 4, 5)).astype(np.float32)
    c1 = O.conv2d(torch.from_numpy(xp), torch.from_numpy(w), 1, 1).numpy()
    c6 = O.conv2d(torch.from_numpy(xp), torch.from_numpy(w), 1, 6).numpy()
    cs2 = O.conv2d_same(torch.from_numpy(xp), torch.from_numpy(w), 2).numpy()
    yt = (rng.uniform(size=(2, 8, 8, 1)) < 0.3).astype(np.float32)
    yl = (rng.uniform(size=(2, 8, 8, 8)) < 0.3).astype(np.float32)
    pp = rng.uniform(size=(2, 8, 8, 2)).astype(np.float32)
    pl = rng.uniform(size=(2, 8, 8, 16)).astype(np.float32)
    m = (rng.uniform(size=(2, 8, 8, 1)) < 0.9).astype(np.float32)
    dl = O.dice_loss(*(torch.from_numpy(a) for a in (yt, pp, yl, pl, m))).item()
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), x=x, up=up, xp=xp, p22=p22, p31=p31,
                        p32=p32, w=w, c1=c1, c6=c6, cs2=cs2, yt=yt, yl=yl, pp=pp, pl=pl, m=m,
                        dice=np.float32(dl))


if __name__ == "__main__":
    model_vgg_small()
    primitives()
    print("golden fixtures written to", HERE)
