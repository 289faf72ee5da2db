"""GPU: the inference forward captured as a HIP graph (infer.GraphedForward) returns bit for bit what
the same launches give one by one — for fresh inputs, for a second input shape, and after the
variables change under it (checkpoint restore)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _resnet(gr, x):
    from tensorflow_ocr_amd.nets import model
    a, b = model.model(x, is_training=False, graph=gr)
    return a.data if hasattr(a, "data") else a, b.data if hasattr(b, "data") else b


def _pixellink(gr, x):
    from tensorflow_ocr_amd.nets import pixellink
    from tensorflow_ocr_amd.tool import pixellink_fn
    net = pixellink.PixelLinkNet(x, graph=gr)
    return net.pixel_scores, pixellink_fn.link_scores(net.link_cls, graph=gr)


@pytest.mark.parametrize("fn", [_resnet, _pixellink], ids=["resnet50_heads", "pixellink_vgg"])
def test_graphed_forward_equals_eager(device, fn):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.infer import GraphedForward
    rng = np.random.default_rng(0)
    ge, gg = Graph(device, seed=2), Graph(device, seed=2)
    fwd = GraphedForward(gg, fn, capture_after=0)

    def image(h, w):
        return torch.from_numpy(rng.uniform(-1, 1, (1, h, w, 3)).astype(np.float32)).to(device)

    def damp(sd):       # random weights under inference-mode BN (moving stats 0 / 1) overflow f16 through 50 layers
        return {k: (v * 0.25 if k.endswith("gamma") else v) for k, v in sd.items()}

    def eager(x):
        out = fn(ge, x)
        ge.reset_tape()
        return [o.clone() for o in out]

    first = True
    for shape in ((96, 128), (96, 128), (64, 64), (96, 128)):
        x = image(*shape)
        if first:       # variables exist after one forward; give both graphs the same damped set
            eager(x)
            fwd(x)
            sd0 = damp(ge.store.state_dict())
            ge.store.load_state_dict(sd0)
            gg.store.load_state_dict(sd0)
            first = False
        want = eager(x)
        got = fwd(x)
        for a, b in zip(got, want):
            assert a.shape == b.shape and torch.isfinite(b).all() and torch.equal(a, b)
    assert len(fwd.cache) == 2
    # new weights under the captured graph: scale every variable, same answer as eager again
    sd = {k: v * 0.5 for k, v in ge.store.state_dict().items()}
    ge.store.load_state_dict(sd)
    gg.store.load_state_dict(sd)
    x = image(96, 128)
    want = eager(x)
    got = fwd(x)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert float(want[0].float().abs().sum()) > 0
    # default policy: a shape is captured the second time it shows up
    lazy = GraphedForward(gg, fn)
    for i, shape in enumerate(((64, 96), (32, 32), (64, 96), (64, 96))):
        x = image(*shape)
        want = eager(x)
        got = lazy(x)
        assert all(torch.equal(a, b) for a, b in zip(got, want))
        assert len(lazy.cache) == (0 if i < 2 else 1)
