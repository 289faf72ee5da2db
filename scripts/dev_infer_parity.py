"""Dev: inference-mode (moving-statistics BN) forward parity of model_vgg vs the f32 oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import ocr_oracle as O
from tensorflow_ocr_amd import checkpoint
from tensorflow_ocr_amd.graph import Graph
from tensorflow_ocr_amd.nets import model_vgg_16 as M

for size, n in [(64, 2), (128, 2), (256, 1)]:
    rng = np.random.default_rng(0)
    p = O.init_model_vgg_params(rng)
    # non-trivial moving statistics
    for k in p:
        if k.endswith('moving_mean'):
            p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
        if k.endswith('moving_variance'):
            p[k] = rng.uniform(0.5, 1.5, p[k].shape).astype(np.float32)
    images, pixel, link, mask = O.synthetic_batch(rng, n, size)
    g = Graph('cuda:0')
    M.model_vgg(images, is_training=False, graph=g)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    px, lk = M.model_vgg(images, is_training=False, graph=g)
    g.reset_tape()
    tp = O.to_torch_params(p)
    with torch.no_grad():
        fpx, flk, _ = O.model_vgg(torch.from_numpy(images), tp, False, mixed=False)
        mpx, mlk, _ = O.model_vgg(torch.from_numpy(images), tp, False, mixed=True)
    dpx, dlk = px.data.cpu().numpy(), lk.data.cpu().numpy()
    print(size, n, "pixel Linf vs f32 %.3e vs mixed %.3e | link vs f32 %.3e vs mixed %.3e | range %.3f..%.3f" % (
        np.abs(dpx - fpx.numpy()).max(), np.abs(dpx - mpx.numpy()).max(),
        np.abs(dlk - flk.numpy()).max(), np.abs(dlk - mlk.numpy()).max(), dpx.min(), dpx.max()))
