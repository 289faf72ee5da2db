"""GPU: every surviving measurement switch, flipped (VERDICT r5 weak point 8: "every surviving non-default switch gets one
GPU test or goes").

The host layers keep alternative formulations of the same arithmetic behind module-level flags (the pass-by-pass forms the
fused kernels replaced: they are what the fused forms are A/B-ed and debugged against) and the library keeps the general
kernel families behind the special ones (OCR_CONV_W4=0 ... : the instantiations other shapes take anyway).  Each flag is
flipped here on a small training run of the net it belongs to; the run must agree with the default configuration — same
loss trajectory within what the 16-bit rounding of one differently-ordered pass grows to over three optimiser steps
of a net that learns its batch by heart (2e-2; measured 0 .. 3e-3, and 6e-2 by the fifth step: most alternates are bit-identical or differ in the
last f16 digit of a few activations) and the same accumulated parameter update (cosine > 0.9, measured 0.926 .. 1: Adam's
first steps are sign-like, a last-digit difference flips the tiniest gradients).  The exact equivalences are held kernel
by kernel in test_gpu_conv_abi.py, test_gpu_layers.py, test_gpu_resnet.py; this file keeps the alternates ALIVE.
The library-side family selectors are read once per process, so they run the convolution ABI sweep in a child interpreter."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the reduced ResNet block list of tests/test_gpu_resnet.py: (depth, bottleneck depth, stride) per unit
SMALL = [("block1", [(128, 64, 1), (128, 64, 2)]), ("block2", [(256, 64, 1), (256, 64, 2)]),
         ("block3", [(256, 128, 1), (256, 128, 2)]), ("block4", [(512, 128, 1)])]


def _run_vgg(device, steps=3, replay=True):
    from tensorflow_ocr_amd import synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    g = Graph(device, seed=3, loss_scale=1024.0)
    batch = [torch.from_numpy(a).to(device) for a in synthetic.make_batch(np.random.default_rng(1), 2, 128)]

    def fl(gr, im, px, lk, mk):
        a, b = M.model_vgg(im, graph=gr)
        return M.loss(px, a, lk, b, mk, graph=gr)
    st = TrainStep(g, fl, lambda gr: AdamOptimizer(gr, learning_rate=2e-4), replay=replay)
    st.build(*batch)
    p0 = g.store.flat.clone()
    losses = [st(*batch).item() for _ in range(steps)]
    return losses, g.store.flat - p0


def _run_pixellink(device, steps=3):
    from tensorflow_ocr_amd import synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import pixellink
    from tensorflow_ocr_amd.train import MomentumOptimizer, TrainStep
    g = Graph(device, seed=4)
    im, px, lk, _ = synthetic.make_batch(np.random.default_rng(7), 2, 128)
    batch = [torch.from_numpy(a).to(device) for a in (im, np.ascontiguousarray(px[..., 0]), lk)]

    def fl(gr, im_, px_, lk_):
        return pixellink.PixelLinkNet(im_, graph=gr).build_loss(px_, lk_)
    st = TrainStep(g, fl, lambda gr: MomentumOptimizer(gr, base_lr=1e-3), replay=True)
    st.build(*batch)
    p0 = g.store.flat.clone()
    losses = [st(*batch).item() for _ in range(steps)]
    return losses, g.store.flat - p0


def _run_east(device, steps=3):
    from tensorflow_ocr_amd import synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    g = Graph(device, seed=5, loss_scale=1024.0)
    batch = [torch.from_numpy(a).to(device) for a in synthetic.make_batch(np.random.default_rng(2), 2, 128)]

    def fl(gr, im, px, lk, mk):
        fs, geo = M.model(im, graph=gr, blocks=SMALL)
        return M.loss(px, fs, lk, geo, mk, graph=gr)
    st = TrainStep(g, fl, lambda gr: AdamOptimizer(gr, learning_rate=2e-4), replay=True)
    st.build(*batch)
    p0 = g.store.flat.clone()
    losses = [st(*batch).item() for _ in range(steps)]
    return losses, g.store.flat - p0


_BASE = {}


def _baseline(device, name, fn):
    if name not in _BASE:
        _BASE[name] = fn(device)
    return _BASE[name]


def _agree(got, ref, what):
    lg, pg = got
    lr, pr = ref
    assert all(np.isfinite(lg)) and bool(torch.isfinite(pg).all()), what
    for a, b in zip(lg, lr):
        assert abs(a - b) <= 2e-2 * max(1.0, abs(b)), (what, lg, lr)
    # the parameters moved, and in the same direction (the accumulated update of the run, per run from the same start)
    assert float(pg.abs().max()) > 0
    cos = float((pg.double() @ pr.double()) / (pg.double().norm() * pr.double().norm()))
    assert cos > 0.9, (what, cos)


VGG_FLAGS = [("layers", "FUSE_BN_REDUCE", False), ("layers", "FUSE_BN_POOL_REDUCE", False), ("layers", "FIRST_RECOMPUTE", False),
             ("layers", "FIRST_DROP_Y", False), ("layers", "FIRST_MOMENTS", False), ("layers", "FIRST_WGRAD_RECOMPUTE", True),
             ("layers", "GUEST_REDUCE", False), ("layers", "FUSE_FIRST_WGRAD", False), ("layers", "FIRST_WGRAD_SUMS", False),
             ("layers", "FUSE_BN_W4_MAXHW", 0), ("ops", "GUEST_BN", False), ("train", "USE_GUESTS", False),
             ("train", "GUEST_MIN_US", 0.0), ("train", "GUEST_COVER", 1.0), ("train", "GUEST_PAIRED_GRID", 64),
             ("train", "GUEST_BALANCE", False), ("train", "XCHG_AT_FORK", False)]


@pytest.mark.parametrize("mod,flag,value", VGG_FLAGS)
def test_vgg_step_with_one_switch_flipped(device, monkeypatch, mod, flag, value):
    import importlib
    ref = _baseline(device, "vgg", _run_vgg)
    m = importlib.import_module("tensorflow_ocr_amd." + mod)
    assert getattr(m, flag) != value
    monkeypatch.setattr(m, flag, value)
    _agree(_run_vgg(device), ref, "%s.%s=%r" % (mod, flag, value))


@pytest.mark.parametrize("flag", ["FUSE_BIAS_POOL", "FUSE_BIAS_RELU"])
def test_pixellink_step_with_one_switch_flipped(device, monkeypatch, flag):
    from tensorflow_ocr_amd import layers
    ref = _baseline(device, "pixellink", _run_pixellink)
    monkeypatch.setattr(layers, flag, not getattr(layers, flag))
    _agree(_run_pixellink(device), ref, flag)


RESNET_FLAGS = ["USE_S2D", "FUSE_TAIL", "FUSE_SUB", "FUSE_FWD", "FUSE_BWD", "FUSE_FWD_ACT", "FUSE_BWD_WIDE", "FUSE_ROOT_POOL",
                "FUSE_ROOT_WGRAD", "FUSE_ROOT_GATHER", "MASK_BITS", "MERGE_HEADS", "MERGE_REORDER"]


@pytest.mark.parametrize("flag", RESNET_FLAGS)
def test_resnet_east_step_with_one_switch_flipped(device, monkeypatch, flag):
    from tensorflow_ocr_amd import resnet_layers as R
    ref = _baseline(device, "east", _run_east)
    monkeypatch.setattr(R, flag, not getattr(R, flag))
    _agree(_run_east(device), ref, flag)


@pytest.mark.parametrize("env", ["OCR_CONV_W4=0", "OCR_CONV_W4S=0", "OCR_CONV_PERSIST=0", "OCR_CONV_PW=0", "OCR_WGRAD3=0"])
def test_kernel_family_selectors_in_a_child_interpreter(device, env):
    """The library's family selectors (read once per process): the convolution ABI sweep of tests/test_gpu_conv_abi.py —
    forward, input gradient, weight gradient of 29 shapes against the oracle — with one special family switched off, so the
    general kernels take its shapes."""
    k, v = env.split("=")
    e = dict(os.environ)
    e[k] = v
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_conv_abi.py"), "-x", "-q", "-k",
                        "test_conv_fwd_dgrad_wgrad", "-p", "no:cacheprovider"], env=e, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
