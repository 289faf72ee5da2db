"""GPU parity of PixelLinkNet (bias VGG-16 + fuse heads, nets/pixellink.py) forward, build_loss and
backward vs the oracle.  Without batch norm the net is far less chaotic than model_vgg, so the
end-to-end bars are tighter (see test_gpu_model_vgg.py for why those are loose)."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu


def test_pixellinknet_forward_loss_backward(device):
    from tensorflow_ocr_amd import checkpoint
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import pixellink
    S = 256.0
    rng = np.random.default_rng(0)
    p = O.init_pixellink_params(rng)
    images, pixel, link, _ = O.synthetic_batch(rng, 2, 64)
    x = (images - 120.0) / 60.0                           # "preprocessed" input, O(1)
    g = Graph(device, loss_scale=S)
    pixellink.PixelLinkNet(x, graph=g)
    g.reset_tape()
    g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order, p))
    net = pixellink.PixelLinkNet(x, graph=g)
    assert net.get_shape('conv3_3') == (16, 16) and net.get_shape('fc7') == (4, 4)
    assert set(net.end_points) == {'conv1_2', 'conv2_2', 'conv3_3', 'conv4_3', 'conv5_3', 'fc6', 'fc7'}
    L = net.build_loss(pixel[..., 0], link)
    assert len(g.collections["losses"]) == 2              # train_pixellink.py:263
    g.backward()
    torch.cuda.synchronize()
    grads = checkpoint.internal_to_tf({n: (v.grad / S).cpu().numpy() for n, v in g.store.vars.items() if v.trainable})

    tp = O.to_torch_params(p)
    opx, olk, _ = O.pixellink_net(torch.from_numpy(x), tp, mixed=True)
    p2, ltot, _ = O.pixellink_build_loss(opx, olk, torch.from_numpy(pixel[..., 0]), torch.from_numpy(link))
    ((p2 + ltot) * S).backward()
    dpx, dlk = net.pixel_cls.data.cpu().numpy(), net.link_cls.data.cpu().numpy()
    sc = max(1.0, float(np.abs(opx.detach().numpy()).max()))
    print("pixel_cls Linf %.3e (scale %.2f)  link_cls Linf %.3e" % (np.abs(dpx - opx.detach().numpy()).max(), sc,
                                                               np.abs(dlk - olk.detach().numpy()).max()))
    assert np.abs(dpx - opx.detach().numpy()).max() < 2e-2 * sc
    assert np.abs(dlk - olk.detach().numpy()).max() < 2e-2 * sc
    assert abs(L.item() - float(p2 + ltot)) < 5e-3
    l2p, llink = [t.item() for t in g.collections["losses"]]
    # (f16 bars; bfloat16 storage rounds 8x coarser at every storage point: tests/test_gpu_bf16.py)
    tol = 8.0 if O.STORAGE == torch.bfloat16 else 1.0
    assert abs(l2p - float(p2)) < 2e-3 * tol and abs(llink - float(ltot)) < 5e-3 * tol

    def cos(a, b):
        a, b = a.ravel().astype(np.float64), b.ravel().astype(np.float64)
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    worst = min(cos(grads[k], (tp[k].grad / S).numpy()) for k in grads if grads[k].size >= 64)
    print("worst gradient cosine", worst)
    assert worst > 0.98
    ps = net.pixel_scores.cpu().numpy()
    assert np.allclose(ps.sum(-1), 1.0, atol=1e-5)


def test_input_normalisation_inside_the_prep_kernel(device):
    """`(x - 120) / 60` of the PixelLink input pipeline folded into ocr_prep_images_norm_f16 (IEEE
    division): the net sees the same f16 pixels as with a host-side normalisation, bit for bit."""
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import pixellink
    rng = np.random.default_rng(1)
    images, _, _, _ = O.synthetic_batch(rng, 2, 64)
    x = ((images - np.float32(120.0)) / np.float32(60.0)).astype(np.float32)
    ga, gb = Graph(device, seed=7), Graph(device, seed=7)
    a = pixellink.PixelLinkNet(x, graph=ga)
    b = pixellink.PixelLinkNet(images, graph=gb, input_norm=(120.0, 60.0))
    torch.cuda.synchronize()
    assert torch.equal(a.pixel_cls.data, b.pixel_cls.data) and torch.equal(a.link_cls.data, b.link_cls.data)
