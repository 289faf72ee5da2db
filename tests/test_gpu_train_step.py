"""GPU: the full training step (forward, dice loss, backward, Adam+EMA) — eager vs recorded/replayed
execution must agree bit for bit (every kernel is deterministic: slab reductions, no atomics), the
Adam update must match the oracle's formula on the device gradients, and the loss must go down."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu


def _make(device, replay):
    from tensorflow_ocr_amd import synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    g = Graph(device, loss_scale=1024.0, seed=3)
    rng = np.random.default_rng(5)
    batch = [torch.from_numpy(a).to(device) for a in synthetic.make_batch(rng, 2, 64)]

    def fl(gr, im, px, lk, mk):
        a, b = M.model_vgg(im, graph=gr)
        return M.loss(px, a, lk, b, mk, graph=gr)
    return g, batch, TrainStep(g, fl, lambda gr: AdamOptimizer(gr, learning_rate=1e-3), replay=replay)


def test_replay_equals_eager_bitwise(device):
    ge, be, se = _make(device, False)
    gr, br, sr = _make(device, True)
    le, lr = [], []
    for i in range(7):
        le.append(se(*be).item())
        lr.append(sr(*br).item())
    assert sr.plan is not None and se.plan is None
    assert le == lr, (le, lr)
    assert le[-1] < le[0]
    assert torch.equal(ge.store.flat, gr.store.flat)
    assert torch.equal(ge.store.flat_aux, gr.store.flat_aux)
    # fresh data goes through the recorded input buffers
    rng = np.random.default_rng(9)
    from tensorflow_ocr_amd import synthetic
    nb = [torch.from_numpy(a).to(device) for a in synthetic.make_batch(rng, 2, 64)]
    assert se(*nb).item() == sr(*nb).item()


def test_adam_ema_update_matches_oracle(device):
    g, batch, step = _make(device, False)
    step(*batch)                       # creates variables + optimiser, first update
    st = g.store
    w0 = st.flat.cpu().numpy().copy()
    m0, v0 = step.opt.m.cpu().numpy().copy(), step.opt.v.cpu().numpy().copy()
    e0 = step.opt.ema.cpu().numpy().copy()
    step(*batch)
    grad = st.flat_grad.cpu().numpy() / g.loss_scale
    grad[:st.n_reg] += 1e-5 * w0[:st.n_reg]                      # slim.l2_regularizer gradient
    w1, m1, v1 = O.adam_update(w0, grad, m0, v0, 2, O.exponential_decay(1e-3, 1))
    assert np.allclose(st.flat.cpu().numpy(), w1, rtol=1e-4, atol=2e-6)
    assert np.allclose(step.opt.m.cpu().numpy(), m1, rtol=1e-4, atol=1e-7)
    d = O.ema_decay(0.997, 1)
    assert np.allclose(step.opt.ema.cpu().numpy(), e0 - (1 - d) * (e0 - w1), rtol=1e-4, atol=2e-6)
