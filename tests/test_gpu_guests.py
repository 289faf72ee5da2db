"""The GUEST forms of the batch-norm backward apply passes (csrc/guest_bn.hip; reference: the gradient of slim.batch_norm
+ ReLU under nets/model_vgg_16.py:144, nets/vgg.py:14-39) against the general kernels they stand in for and against a
float64 restatement, and the recorded step's pairing of each with the weight gradient it runs beside."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu
BF16 = O.STORAGE == torch.bfloat16
ULP = 2.0 ** -7 if BF16 else 2.0 ** -10          # one rounding of the 16-bit result


def _inputs(rng, n, h, w, c, device):
    from tensorflow_ocr_amd.graph import F16
    y = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(F16).to(device)
    scale = torch.from_numpy((rng.uniform(0.5, 1.5, c) * rng.choice([-1.0, 1.0], c, p=[0.1, 0.9])).astype(np.float32)).to(device)
    shift = torch.from_numpy(rng.normal(0, 0.3, c).astype(np.float32)).to(device)
    mean = torch.from_numpy(rng.normal(0, 0.2, c).astype(np.float32)).to(device)
    invstd = torch.from_numpy(rng.uniform(0.7, 1.3, c).astype(np.float32)).to(device)
    T = 37
    partial = torch.from_numpy((rng.standard_normal((T, 2, c)) * 3).astype(np.float32)).to(device)
    return y, scale, shift, mean, invstd, partial, T


@pytest.mark.parametrize("n,h,w,c,relu", [(2, 16, 24, 64, True), (1, 15, 9, 128, True), (3, 7, 5, 256, True),
                                          (2, 8, 8, 512, True), (1, 4, 6, 1024, True), (2, 9, 11, 32, False),
                                          (1, 33, 17, 8, True)])
def test_apply_affine_guest_equals_the_general_apply_pass(device, n, h, w, c, relu):
    """ocr_bn_bwd_coefficients + ocr_bn_relu_bwd_apply_affine_f16 against ocr_bn_relu_bwd_apply_f16 on the same partial
    sums: dgamma / dbeta bit-identical (the same finalisation), dy within one rounding of the 16-bit result (the affine
    form A*dz + B*y + C and sc*(dz - k_dz - xhat*k_dzx) are two f32 evaluations of one expression), the ReLU mask
    identical — asserted through a float64 evaluation with the mask taken from the stored activation."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    rng = np.random.default_rng(c * 7 + h)
    y, scale, shift, mean, invstd, partial, T = _inputs(rng, n, h, w, c, device)
    da = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(F16).to(device)
    ws = ops.Workspace(device, 8 << 20)
    assert ops.guest_apply_ok(y.shape)
    dg0, db0, dy0 = torch.zeros(c, device=device), torch.zeros(c, device=device), torch.empty_like(y)
    ops.bn_relu_bwd_apply(y, scale, shift, mean, invstd, da, relu, partial, T, dg0, db0, dy0, ws)
    dg1, db1 = torch.zeros(c, device=device), torch.zeros(c, device=device)
    dy1 = torch.full_like(y, float("nan"))
    coef = tuple(torch.empty(c, device=device) for _ in range(3))
    ops.bn_bwd_coefficients_pre(partial, T, c, float(n * h * w), scale, mean, invstd, dg1, db1, coef, ws)
    ops.bn_relu_bwd_apply_affine(y, da, scale, shift, coef[1], coef[2], relu, dy1)
    torch.cuda.synchronize()
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert not torch.isnan(dy1.float()).any()
    # float64 from the same inputs, the mask from the STORED activation (16-bit rounding of fma(y, scale, shift))
    yf = y.double().cpu().numpy()
    sc, sh = scale.double().cpu().numpy(), shift.double().cpu().numpy()
    act = (y.float() * scale + shift).to(F16)                       # (torch evaluates mul + add: ties may differ from the fma ...)
    act_fma = torch.from_numpy((yf * sc + sh).astype(np.float32)).to(F16)   # ... the exact product rounded once IS the fma
    mask = (act_fma.float().numpy() > 0) if relu else np.ones_like(yf, bool)
    dz = da.double().cpu().numpy() * mask
    N = float(n * h * w)
    k_dz, k_dzx = db0.double().cpu().numpy() / N, dg0.double().cpu().numpy() / N
    xh = (yf - mean.double().cpu().numpy()) * invstd.double().cpu().numpy()
    ref = sc * (dz - k_dz - xh * k_dzx)
    tol = ULP * np.abs(ref) + 1e-6 + 1e-6 * np.abs(ref).max()
    for name, dy in (("general", dy0), ("guest", dy1)):
        err = np.abs(dy.double().cpu().numpy() - ref)
        assert (err <= tol).all(), (name, float((err / tol).max()))
    assert float((dy0.float() - dy1.float()).abs().max()) <= 2 * ULP * float(dy0.float().abs().max())
    assert act.shape == act_fma.shape


@pytest.mark.parametrize("n,h,w,c,relu", [(2, 16, 24, 64, True), (1, 14, 10, 128, True), (3, 8, 8, 32, False),
                                          (1, 6, 4, 512, True)])
def test_pooled_apply_affine_guest_equals_the_general_pass(device, n, h, w, c, relu):
    """ocr_bn_relu_pool_bwd_idx_apply_affine_f16 against ocr_bn_relu_pool_bwd_idx_apply_f16 (stored first-max index, even
    maps): same routing, dy within one rounding; odd maps are refused (the host keeps the general kernel)."""
    from tensorflow_ocr_amd import _lib as L, ops
    from tensorflow_ocr_amd.graph import F16
    rng = np.random.default_rng(c + h)
    y, scale, shift, mean, invstd, partial, T = _inputs(rng, n, h, w, c, device)
    y[0, :2, :2, :8] = 0.5                               # ties: the FIRST maximum takes the gradient
    oh, ow = h // 2, w // 2
    da = torch.from_numpy(rng.standard_normal((n, oh, ow, c)).astype(np.float32)).to(F16).to(device)
    ws = ops.Workspace(device, 8 << 20)
    pooled = torch.empty((n, oh, ow, c), dtype=F16, device=device)
    am = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=device)
    ops.bn_relu_pool_idx(y, scale, shift, relu, None, pooled, am)
    dg0, db0, dy0 = torch.zeros(c, device=device), torch.zeros(c, device=device), torch.empty_like(y)
    ops.bn_relu_pool_bwd_idx_apply(y, scale, mean, invstd, am, da, relu, partial, T, dg0, db0, dy0, ws)
    dg1, db1 = torch.zeros(c, device=device), torch.zeros(c, device=device)
    dy1 = torch.full_like(y, float("nan"))
    coef = tuple(torch.empty(c, device=device) for _ in range(3))
    ops.bn_bwd_coefficients_pre(partial, T, c, float(n * h * w), scale, mean, invstd, dg1, db1, coef, ws)
    ops.bn_relu_pool_bwd_idx_apply_affine(y, am, da, coef, relu, dy1)
    torch.cuda.synchronize()
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert not torch.isnan(dy1.float()).any()
    d0, d1 = dy0.float(), dy1.float()
    assert float((d0 - d1).abs().max()) <= 2 * ULP * float(d0.abs().max())
    # where the pooled gradient was routed the two agree on WHERE: dz enters with weight A (|A| >= 0.5 here)
    routed0 = (d0 - (coef[1] * y.float() + coef[2])).abs() > 0.2 * da.float().abs().max()
    routed1 = (d1 - (coef[1] * y.float() + coef[2])).abs() > 0.2 * da.float().abs().max()
    assert torch.equal(routed0, routed1)
    odd = torch.empty((1, 5, 6, 64), dtype=F16, device=device)
    with pytest.raises(L.OcrHipError):
        ops.bn_relu_pool_bwd_idx_apply_affine(odd, am, da, coef, relu, torch.empty_like(odd))


@pytest.mark.parametrize("n,h,w,c,relu", [(2, 16, 24, 64, True), (1, 14, 10, 256, True), (2, 8, 8, 512, True), (3, 6, 4, 32, True)])
def test_pooled_end_point_guest_equals_the_general_backward(device, n, h, w, c, relu):
    """A pooled layer that is also an end point (conv3_3 / conv4_3): ocr_bn_relu_bwd_reduce_f16(da_pool) + the guest
    ocr_bn_relu_poolfull_bwd_apply_affine_f16 (first-max index stored by the forward) against ocr_bn_relu_bwd_f16(pool = 2,
    da_full): dgamma / dbeta bit-identical (the same reduction kernel and finalisation), dy within one rounding, and the
    forward with the index gives the activations of the plain pooled forward bit for bit."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    rng = np.random.default_rng(c + h)
    y, scale, shift, mean, invstd, _, _ = _inputs(rng, n, h, w, c, device)
    y[0, :2, :2, :8] = 0.5                               # ties: the FIRST maximum takes the pooled gradient
    oh, ow = h // 2, w // 2
    da_full = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(F16).to(device)
    da_pool = torch.from_numpy(rng.standard_normal((n, oh, ow, c)).astype(np.float32)).to(F16).to(device)
    ws = ops.Workspace(device, 16 << 20)
    f0, p0 = torch.empty_like(y), torch.empty((n, oh, ow, c), dtype=F16, device=device)
    f1, p1 = torch.empty_like(y), torch.empty_like(p0)
    am = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=device)
    ops.bn_relu(y, scale, shift, relu, 2, f0, p0)
    ops.bn_relu_pool_idx(y, scale, shift, relu, f1, p1, am, None)
    assert torch.equal(f0, f1) and torch.equal(p0, p1)
    dg0, db0, dy0 = torch.zeros(c, device=device), torch.zeros(c, device=device), torch.empty_like(y)
    ops.bn_relu_bwd(y, scale, shift, mean, invstd, da_full, da_pool, relu, 2, dg0, db0, dy0, ws)
    dg1, db1 = torch.zeros(c, device=device), torch.zeros(c, device=device)
    dy1 = torch.full_like(y, float("nan"))
    coef = tuple(torch.empty(c, device=device) for _ in range(3))
    ops.bn_relu_bwd_reduce(y, scale, shift, mean, invstd, da_full, relu, dg1, db1, coef, ws, da_pool=da_pool)
    ops.bn_relu_poolfull_bwd_apply_affine(y, da_full, da_pool, am, scale, shift, coef, relu, dy1)
    torch.cuda.synchronize()
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert not torch.isnan(dy1.float()).any()
    d0, d1 = dy0.float(), dy1.float()
    assert float((d0 - d1).abs().max()) <= 2 * ULP * float(d0.abs().max())
    # float64 from the inputs: routing by the first maximum of the STORED activation, mask by the position's own activation
    yf = y.double().cpu().numpy()
    sc, sh = scale.double().cpu().numpy(), shift.double().cpu().numpy()
    act = torch.from_numpy((yf * sc + sh).astype(np.float32)).to(F16).float().numpy()
    if relu:
        act = np.maximum(act, 0)
    win = act.reshape(n, oh, 2, ow, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, oh, ow, 4, c)
    first = win.argmax(3)                                   # numpy: the first maximum, as TF's gradient routing
    routed = np.zeros((n, oh, ow, 4, c))
    np.put_along_axis(routed, first[:, :, :, None, :], da_pool.double().cpu().numpy()[:, :, :, None, :], 3)
    routed = routed.reshape(n, oh, ow, 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h, w, c)
    dz = (da_full.double().cpu().numpy() + routed) * ((act > 0) if relu else 1.0)
    N = float(n * h * w)
    xh = (yf - mean.double().cpu().numpy()) * invstd.double().cpu().numpy()
    ref = sc * (dz - db0.double().cpu().numpy() / N - xh * dg0.double().cpu().numpy() / N)
    tol = ULP * np.abs(ref) + 1e-6 + 1e-6 * np.abs(ref).max()
    for name, dy in (("general", dy0), ("guest", dy1)):
        err = np.abs(dy.double().cpu().numpy() - ref)
        assert (err <= tol).all(), (name, float((err / tol).max()))
    # dgamma / dbeta themselves
    assert np.allclose(db0.double().cpu().numpy(), dz.sum((0, 1, 2)), rtol=2e-4, atol=2e-3)
    assert np.allclose(dg0.double().cpu().numpy(), (dz * xh).sum((0, 1, 2)), rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("n,h,w,c,relu,pooled", [(2, 16, 24, 64, True, False), (1, 15, 9, 128, True, False),
                                                 (3, 7, 5, 256, False, False), (1, 4, 6, 1024, True, False),
                                                 (32, 32, 32, 512, True, False), (2, 16, 24, 64, True, True),
                                                 (1, 14, 10, 256, True, True), (8, 64, 64, 512, True, True)])
def test_reduce_rows_guest_equals_the_general_reduction(device, n, h, w, c, relu, pooled):
    """ocr_bn_relu_bwd_reduce_rows_f16 (the end-point layers' reduction pass as a guest: per-thread partial rows, no LDS) +
    ocr_bn_bwd_coefficients against ocr_bn_relu_bwd_reduce_f16 on the same tensors: dgamma, dbeta and the three
    coefficient rows agree to f32 summation accuracy (another order of the same sums) and with float64 on the host."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.graph import F16
    rng = np.random.default_rng(c + h + n)
    y, scale, shift, mean, invstd, _, _ = _inputs(rng, n, h, w, c, device)
    da_full = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(F16).to(device)
    ws = ops.Workspace(device, 64 << 20)
    da_pool = am = None
    if pooled:
        oh, ow = h // 2, w // 2
        y[0, :2, :2, :8] = 0.5
        da_pool = torch.from_numpy(rng.standard_normal((n, oh, ow, c)).astype(np.float32)).to(F16).to(device)
        am = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=device)
        ops.bn_relu_pool_idx(y, scale, shift, relu, torch.empty_like(y), torch.empty((n, oh, ow, c), dtype=F16, device=device),
                             am, None)
    dg0, db0 = torch.zeros(c, device=device), torch.zeros(c, device=device)
    coef0 = tuple(torch.empty(c, device=device) for _ in range(3))
    ops.bn_relu_bwd_reduce(y, scale, shift, mean, invstd, da_full, relu, dg0, db0, coef0, ws, da_pool=da_pool)
    dg1, db1 = torch.full((c,), float("nan"), device=device), torch.full((c,), float("nan"), device=device)
    coef1 = tuple(torch.full((c,), float("nan"), device=device) for _ in range(3))
    part, T = ops.bn_relu_bwd_reduce_rows(y, da_full, scale, shift, mean, invstd, relu,
                                          lambda shp, dt: torch.full(shp, float("nan"), dtype=dt, device=device),
                                          da_pool=da_pool, argmax=am)
    assert T <= 256 * (256 // (c // 4)) and not torch.isnan(part).any()
    ops.bn_bwd_coefficients_pre(part, T, c, float(n * h * w), scale, mean, invstd, dg1, db1, coef1, ws)
    torch.cuda.synchronize()
    # float64 on the host
    yf = y.double().cpu().numpy()
    sc, sh = scale.double().cpu().numpy(), shift.double().cpu().numpy()
    act = torch.from_numpy((yf * sc + sh).astype(np.float32)).to(F16).float().numpy()
    g = da_full.double().cpu().numpy()
    if pooled:
        a2 = np.maximum(act, 0) if relu else act
        oh, ow = h // 2, w // 2
        win = a2.reshape(n, oh, 2, ow, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, oh, ow, 4, c)
        first = win.argmax(3)
        routed = np.zeros((n, oh, ow, 4, c))
        np.put_along_axis(routed, first[:, :, :, None, :], da_pool.double().cpu().numpy()[:, :, :, None, :], 3)
        g = g + routed.reshape(n, oh, ow, 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h, w, c)
    dz = g * ((act > 0) if relu else 1.0)
    xh = (yf - mean.double().cpu().numpy()) * invstd.double().cpu().numpy()
    ref_b, ref_g = dz.sum((0, 1, 2)), (dz * xh).sum((0, 1, 2))
    scale_b = np.abs(dz).sum((0, 1, 2)).max()
    for name, db, dg in (("general", db0, dg0), ("guest", db1, dg1)):
        assert np.abs(db.double().cpu().numpy() - ref_b).max() <= 2e-6 * scale_b + 1e-4, name
        assert np.abs(dg.double().cpu().numpy() - ref_g).max() <= 4e-6 * scale_b + 1e-4, name
    for a0, a1 in zip(coef0, coef1):
        d = (a0 - a1).abs().max().item()
        assert d <= 1e-5 * max(1.0, a0.abs().max().item()), d


def test_recorded_step_pairs_every_guest_with_a_weight_gradient(device, monkeypatch):
    """model_vgg's recorded step (train.schedule_guests): each guest apply pass sits between a FORK in front of the
    weight gradient it runs beside and a JOIN behind it, its coefficient call in front of the fork; the replayed step
    (two streams) gives the parameters of the eager step (one stream) bit for bit."""
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.nets import model_vgg_16 as M
    from tensorflow_ocr_amd import train
    from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
    monkeypatch.setattr(train, "GUEST_MIN_US", 0.0)          # (at this size every pass is shorter than the pairing threshold)
    rng = np.random.default_rng(0)
    images, pixel, link, mask = O.synthetic_batch(rng, 2, 128)
    batch = [torch.from_numpy(a).to(device) for a in (images, pixel, link, mask)]

    def fl(g, im, px, lk, mk):
        p, l = M.model_vgg(im, graph=g)
        return M.loss(px, p, lk, l, mk, graph=g)
    outs = []
    for replay in (True, False):
        g = Graph(device, loss_scale=1024.0, seed=3)
        step = TrainStep(g, fl, lambda gg: AdamOptimizer(gg), replay=replay)
        for _ in range(5):
            step(*batch)
        torch.cuda.synchronize()
        outs.append(g.store.flat.clone())
        if replay:
            plan = step.plan
            kinds = [e[0] if e[0] != "c" else (e[4][0] if e[4] is not None else "c") for e in plan]
            forks = [i for i, k in enumerate(kinds) if k == "fork"]
            assert len(forks) >= 8, kinds                        # conv1_2 ... conv5_2 and fc6, minus guests left without a host
            for i in forks:
                assert plan[i + 1][4][0] == "guest" and plan[i + 1][4][-1] == "paired" and kinds[i + 2] == "side"
                j = kinds.index("join", i)
                assert all(k == "side" for k in kinds[i + 2:j]) and kinds[j + 1] in ("reduce", "side")
            assert kinds.count("fork") == kinds.count("join")
            # every weight gradient is issued, and before the optimiser
            opt = max(i for i, e in enumerate(plan) if e[0] == "py")
            sides = [i for i, k in enumerate(kinds) if k == "side"]
            assert len(sides) >= 14 and max(sides) < opt               # fc7, fc6, conv5_3 ... conv1_2
    assert torch.equal(outs[0], outs[1])
