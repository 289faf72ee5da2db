"""GPU: batch-norm finalisations CHAINED to the launch that produces their partial rows (round 6; csrc/bn_reduce.h,
include/ocr_hip.h: ocr_bn_finalize_arm, ocr_bn_bwd_coefficients_arm).

The closing workgroups run the very code of the stand-alone launch, in the same order, so every output must be
bit-identical to `convolution ; ocr_bn_finalize` / `input-gradient convolution ; ocr_bn_bwd_coefficients` — on the kernels
that carry the finalisation in their own grid (conv3x3_w4, conv3x3_w4s, conv_igemm, conv_pw) and on those that fall back
to the separate launch behind their kernel (16-row tiles, the persistent 64-channel kernel).  The arrival counters reset
themselves (every case runs three times on one stream, the slots rotate through all sixteen), misuse is refused, and a
whole training step is bit-identical with the chain on and off."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu


def _h(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(O.STORAGE).float().numpy()


CASES = [
    # n, h, w, cin, cout, k, dil — variant expected to carry the chain (None: falls back to the separate launch)
    (2, 37, 70, 256, 256, 3, 1, "conv3x3_w4_kernel"),          # 1 cout tile, T = 30 rows, R = 1
    (2, 32, 64, 128, 512, 3, 1, "conv3x3_w4_kernel"),          # 2 cout tiles, 8 channel groups
    (9, 64, 96, 64, 256, 3, 1, "conv3x3_w4_kernel"),           # T = 216
    (6, 128, 160, 64, 256, 3, 1, "conv3x3_w4_kernel"),         # T = 480
    (5, 136, 256, 64, 256, 3, 1, "conv3x3_w4_kernel"),         # T = 680 rows
    (18, 128, 256, 64, 256, 3, 1, "conv3x3_w4_kernel"),        # T = 2304 rows: R > 1 (stage rows, two ticket levels)
    (2, 50, 70, 128, 128, 3, 1, "conv3x3_w4s_kernel<128>"),
    (2, 16, 40, 128, 64, 3, 1, "conv3x3_w4s_kernel<64>"),
    (2, 12, 20, 256, 512, 3, 6, None),                         # dilated: conv_igemm (chained once that kernel carries it)
    (2, 24, 40, 256, 256, 1, 1, None),                         # pointwise GEMM kernel
    (2, 100, 130, 64, 64, 3, 1, None),                         # persistent 64-channel kernel: always the separate launch
]


def _conv_setup(device, n, h, w, cin, cout, k, dil, seed):
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(_h(rng.standard_normal((n, h, w, cin)))).to(O.STORAGE).to(device)
    wt = torch.from_numpy(_h(rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin)))).to(device)
    w_kc = torch.empty((k * k, cout, cin), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((k * k, cin, cout), dtype=O.STORAGE, device=device)
    ops.pack_weights(wt, w_kc, w_ck)
    return rng, x, w_kc


def _fin_buffers(device, rng, c):
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(device)
    return dict(gamma=f32(rng.uniform(0.5, 1.5, c)), beta=f32(rng.normal(0, 0.3, c)),
                mm=f32(rng.normal(0, 0.1, c)), mv=f32(rng.uniform(0.5, 1.5, c)),
                scale=torch.full((c,), 7.0, device=device), shift=torch.full((c,), 7.0, device=device),
                mean=torch.full((c,), 7.0, device=device), invstd=torch.full((c,), 7.0, device=device))


@pytest.mark.parametrize("n,h,w,cin,cout,k,dil,variant", CASES)
def test_forward_statistics_chained_equal_the_separate_launch_bitwise(device, n, h, w, cin, cout, k, dil, variant):
    from tensorflow_ocr_amd import _lib, ops
    lib = _lib.load()
    rng, x, w_kc = _conv_setup(device, n, h, w, cin, cout, k, dil, cin + cout + h)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, 1, dil)
    if variant is not None:
        assert ops.conv2d_variant(d) == variant
    d.flags = ops.CONV_STATS
    T = ops.conv2d_num_mtiles(d)
    count = float(n * h * w)
    stage = torch.empty((ops.bn_reduce_workspace(T, cout),), dtype=torch.uint8, device=device)
    init = _fin_buffers(device, rng, cout)
    for rep in range(3):                                    # (the counters must come back to zero by themselves)
        ref = {k_: v.clone() for k_, v in init.items()}
        got = {k_: v.clone() for k_, v in init.items()}
        y_r = torch.empty((n, h, w, cout), dtype=O.STORAGE, device=device)
        y_g = torch.empty_like(y_r)
        p_r = torch.zeros((T, 2, cout), dtype=torch.float32, device=device)
        p_g = torch.zeros_like(p_r)
        ops.conv2d(d, x, w_kc, y_r, None, p_r)
        ops.bn_finalize(p_r, T, cout, count, ref["gamma"], ref["beta"], 1e-5, 0.997, ref["mm"], ref["mv"], ref["scale"],
                        ref["shift"], ref["mean"], ref["invstd"], stage)
        torch.cuda.synchronize()
        ops.bn_finalize_arm(p_g, T, cout, count, got["gamma"], got["beta"], 1e-5, 0.997, got["mm"], got["mv"], got["scale"],
                            got["shift"], got["mean"], got["invstd"], stage)
        assert lib.ocr_bn_armed() == 1
        ops.conv2d(d, x, w_kc, y_g, None, p_g)
        assert lib.ocr_bn_armed() == 0
        torch.cuda.synchronize()
        assert torch.equal(y_g, y_r) and torch.equal(p_g, p_r)
        for k_ in ("scale", "shift", "mean", "invstd", "mm", "mv"):
            assert torch.equal(got[k_], ref[k_]), (k_, rep)
        assert float(got["scale"].abs().max()) != 7.0


@pytest.mark.parametrize("n,h,w,c,variant", [(2, 37, 70, 256, "conv3x3_w4_kernel"), (3, 64, 64, 512, "conv3x3_w4_kernel"),
                                             (2, 50, 70, 128, "conv3x3_w4s_kernel<128>"),
                                             (2, 16, 40, 64, None)])
def test_backward_coefficients_chained_equal_the_separate_launch_bitwise(device, n, h, w, c, variant):
    """The input-gradient convolution whose epilogue sums the BN-backward terms of the layer below (ocr_conv2d_bnred_f16),
    with that layer's dgamma / dbeta / apply coefficients as its closing workgroups."""
    from tensorflow_ocr_amd import ops
    rng, x, w_kc = _conv_setup(device, n, h, w, c, c, 3, 1, c + h)
    d = ops.conv_desc((n, h, w, c), c, 3, 3, 1, 1)
    if variant is not None:
        assert ops.conv2d_variant(d) == variant
    d.flags = 0
    T = ops.conv2d_num_mtiles(d)
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(device)
    by = torch.from_numpy(_h(rng.standard_normal((n, h, w, c)))).to(O.STORAGE).to(device)
    ctx = (by, f32(rng.uniform(0.5, 1.5, c)), f32(rng.normal(0, 0.3, c)), f32(rng.normal(0, 0.2, c)),
           f32(rng.uniform(0.7, 1.3, c)), True)
    count = float(n * h * w)
    ws = ops.Workspace(device, 8 << 20)
    stage = torch.empty((ops.bn_reduce_workspace(T, c),), dtype=torch.uint8, device=device)
    for rep in range(3):
        outs = []
        for chained in (False, True):
            dx = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
            part = torch.zeros((T, 2, c), dtype=torch.float32, device=device)
            dgamma, dbeta = torch.full((c,), 7.0, device=device), torch.full((c,), 7.0, device=device)
            coef = tuple(torch.full((c,), 7.0, device=device) for _ in range(3))
            if chained:
                ops.bn_bwd_coefficients_arm(part, T, c, count, ctx[1], ctx[3], ctx[4], dgamma, dbeta, coef, stage)
            ops.conv2d_bnred(d, x, w_kc, dx, part, ctx)
            if not chained:
                ops.bn_bwd_coefficients(part, T, c, count, ctx[1], ctx[3], ctx[4], dgamma, dbeta, coef, ws)
            torch.cuda.synchronize()
            outs.append((dx, part, dgamma, dbeta) + coef)
        for a, b in zip(*outs):
            assert torch.equal(a, b), rep
        assert float(outs[1][2].abs().max()) != 7.0


def test_chain_misuse_is_refused(device):
    """Arming twice, or arming for rows the next convolution does not produce, is an error — and disarms."""
    from tensorflow_ocr_amd import _lib, ops
    lib = _lib.load()
    rng, x, w_kc = _conv_setup(device, 1, 16, 32, 64, 256, 3, 1, 3)
    d = ops.conv_desc((1, 16, 32, 64), 256, 3, 3, 1, 1)
    d.flags = ops.CONV_STATS
    T = ops.conv2d_num_mtiles(d)
    b = _fin_buffers(device, rng, 256)
    stage = torch.empty((ops.bn_reduce_workspace(T, 256),), dtype=torch.uint8, device=device)
    p1 = torch.zeros((T, 2, 256), dtype=torch.float32, device=device)
    p2 = torch.zeros_like(p1)
    y = torch.empty((1, 16, 32, 256), dtype=O.STORAGE, device=device)
    arm = lambda p: ops.bn_finalize_arm(p, T, 256, 512.0, b["gamma"], b["beta"], 1e-5, 0.997, b["mm"], b["mv"], b["scale"],
                                        b["shift"], b["mean"], b["invstd"], stage)
    arm(p1)
    with pytest.raises(_lib.OcrHipError):
        arm(p1)
    assert lib.ocr_bn_armed() == 0
    arm(p1)
    with pytest.raises(_lib.OcrHipError):
        ops.conv2d(d, x, w_kc, y, None, p2)                # other rows than the armed ones
    assert lib.ocr_bn_armed() == 0
    torch.cuda.synchronize()
    ops.conv2d(d, x, w_kc, y, None, p2)                    # nothing armed: plain launch
    torch.cuda.synchronize()


def test_training_step_is_bit_identical_with_and_without_the_chain(device, monkeypatch):
    """model_vgg + dice loss + backward + Adam, 4 steps: OCR_CHAIN_BN on / off leave the same parameters, statistics and
    losses, eagerly and replayed; the recorded plan holds no stand-alone finalisation behind a chained convolution."""
    from tensorflow_ocr_amd import layers, synthetic
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep

    def run(chain, replay):
        monkeypatch.setattr(layers, "CHAIN_BN", chain)
        g = Graph(device, seed=3, loss_scale=1024.0)
        rng = np.random.default_rng(1)
        batch = [torch.from_numpy(a).to(device) for a in synthetic.make_batch(rng, 2, 128)]

        def fl(gr, im, px, lk, mk):
            a, b = M.model_vgg(im, graph=gr)
            return M.loss(px, a, lk, b, mk, graph=gr)
        st = TrainStep(g, fl, lambda gr: AdamOptimizer(gr, learning_rate=1e-3), replay=replay)
        losses = [st(*batch).item() for _ in range(5)]
        return g, st, losses
    g0, s0, l0 = run(False, False)
    g1, s1, l1 = run(True, False)
    g2, s2, l2 = run(True, True)
    assert l0 == l1 == l2, (l0, l1, l2)
    assert torch.equal(g0.store.flat, g1.store.flat) and torch.equal(g0.store.flat, g2.store.flat)
    assert torch.equal(g0.store.flat_aux, g1.store.flat_aux) and torch.equal(g0.store.flat_aux, g2.store.flat_aux)
    names = [e[3] for e in s2.plan if e[0] == "c"]
    assert names.count("ocr_bn_finalize_arm") >= 12 and names.count("ocr_bn_bwd_coefficients_arm") >= 8, (
        names.count("ocr_bn_finalize_arm"), names.count("ocr_bn_bwd_coefficients_arm"))
    for i, nm in enumerate(names):
        if nm in ("ocr_bn_finalize_arm", "ocr_bn_bwd_coefficients_arm"):
            assert names[i + 1].startswith("ocr_conv2d"), (nm, names[i + 1])      # armed for the very next launch
