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


def _make_pixellink(device, replay, normalise_inside=False):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import pixellink
    from tensorflow_ocr_amd.train import MomentumOptimizer, TrainStep
    g = Graph(device, seed=4)

    def fl(gr, im, px, lk):
        net = pixellink.PixelLinkNet((im - 120.0) / 60.0 if normalise_inside else im, graph=gr)
        return net.build_loss(px, lk)
    return g, TrainStep(g, fl, lambda gr: MomentumOptimizer(gr, base_lr=1e-3), replay=replay)


def _pixellink_batch(device, seed):
    from tensorflow_ocr_amd import synthetic
    im, px, lk, _ = synthetic.make_batch(np.random.default_rng(seed), 2, 64)
    return [torch.from_numpy(a).to(device) for a in (im, np.ascontiguousarray(px[..., 0]), lk)]


def test_pixellink_replay_equals_eager_on_a_fresh_batch_every_step(device):
    """Every step gets new data (as the feeder delivers it) and nothing synchronises in between:
    the replayed plan must consume the new batch, not anything left over from the recorded step."""
    ge, se = _make_pixellink(device, False)
    gr, sr = _make_pixellink(device, True)
    le, lr = [], []
    for i in range(8):
        b = _pixellink_batch(device, 20 + i)
        b[0] = (b[0] - 120.0) / 60.0
        le.append(se(*b).data.clone())
        lr.append(sr(*[t.clone() for t in b]).data.clone())
        del b
        torch.empty(1 << 20, device=device).fill_(float("nan"))       # recycle freed blocks with poison
    assert sr.plan is not None and se.plan is None
    le = [v[0].item() for v in le]
    lr = [v[0].item() for v in lr]
    assert le == lr, (le, lr)
    assert all(np.isfinite(le))
    assert torch.equal(ge.store.flat, gr.store.flat)


def test_recording_refuses_torch_operators_inside_the_step(device):
    """A torch operator on device data inside forward_loss runs once, at record time; replay would
    read its stale output.  The recording step must raise instead of producing such a plan."""
    g, step = _make_pixellink(device, True, normalise_inside=True)
    b = _pixellink_batch(device, 1)
    step(*b)
    step(*b)                                    # eager steps are fine
    with pytest.raises(RuntimeError, match="would not be replayed"):
        step(*b)
    g2, eager = _make_pixellink(device, False, normalise_inside=True)
    for _ in range(4):
        eager(*b)                               # replay=False keeps torch operators legal


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


_DP_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as td
from tensorflow_ocr_amd import dist, synthetic
from tensorflow_ocr_amd.graph import Graph
from tensorflow_ocr_amd.nets import model_vgg_16 as M
from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
os.environ["LOCAL_RANK"] = "0"                      # both ranks share the single GPU of the test box
rank, world, _ = dist.init_process_group_from_env("gloo")
dev = torch.device("cuda:0")
g = Graph(dev, loss_scale=1024.0, seed=3)            # same init on every rank
rng = np.random.default_rng(50 + rank)               # different data per rank
batch = [torch.from_numpy(a).to(dev) for a in synthetic.make_batch(rng, 2, 64)]
def fl(gr, im, px, lk, mk):
    a, b = M.model_vgg(im, graph=gr)
    return M.loss(px, a, lk, b, mk, graph=gr)
step = TrainStep(g, fl, lambda gr: AdamOptimizer(gr, learning_rate=1e-3), world_size=world, bucket_bytes=8 << 20)
for i in range(6):                                   # 2 eager + 1 recorded + 3 replayed
    step(*batch)
torch.cuda.synchronize()
assert step.plan is not None and len(step.reducer.buckets) >= 2
flat = g.store.flat.detach().cpu()
ref = flat.clone()
td.broadcast(ref, src=0)
assert torch.equal(flat, ref), "ranks diverged: max diff %%g" %% (flat - ref).abs().max().item()
assert torch.isfinite(flat).all()
td.barrier(); td.destroy_process_group()
print("rank", rank, "ok")
"""


def test_data_parallel_two_ranks_stay_in_sync(device, tmp_path):
    """multigpu_train.py:70-85 semantics: after every step all towers hold identical parameters
    (mean of tower gradients).  Two ranks share the GPU through gloo — a functional check of
    TrainStep's bucketed all-reduce hooks, side streams and step replay; RCCL itself is exercised by
    the driver's multi-GPU bench."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp.py"
    script.write_text(_DP_WORKER % root)
    import socket

    def launch():
        with socket.socket() as sk:                  # a port nobody holds right now
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        ps = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            ps.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                       stderr=subprocess.STDOUT))
        try:
            return ps, [p.communicate(timeout=180)[0].decode() for p in ps]
        except subprocess.TimeoutExpired:
            for p in ps:                             # a rendezvous that never formed: do not sit on it
                p.kill()
            return ps, None
    procs, outs = launch()
    if outs is None:
        procs, outs = launch()                       # one more try on a fresh port
    assert outs is not None, "two-rank rendezvous timed out twice"
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
