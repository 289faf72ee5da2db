"""GPU: the three MFMA convolution entry points called directly through the C ABI over a sweep of
shapes that exercises every tile variant (256/128/64/32-cout tiles, 16-row tiles, 16x16x32 and
32x32x16 MFMA, 32- and 64-channel chunks, LDS-DMA weight ring), the tap-sweeping / GEMM-tiled /
per-tap-dilated weight-gradient kernels, ragged edges and partial blocks — against the oracle's
convolution (torch CPU f32 on the same f16-rounded operands).

Tolerance: operands are exact f16 values, products accumulate in f32 on both sides; the forward /
dgrad outputs are rounded to f16 once (<= 2^-11 relative = 4.9e-4 of the tensor's max here 1e-3),
weight gradients stay f32 (measured 1e-7..5e-7, bar 5e-6: accumulation order only).  Under
OCR_STORAGE=bf16 (libocr_hip_bf16.so, run by tests/test_gpu_bf16.py) the operands are exact bf16
values and the single output rounding is 2^-8: bar 8e-3."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu

SHAPES = [
    # n, h,  w,  cin, cout, k, dil
    (2, 16, 40, 64, 64, 3, 1),
    (1, 24, 33, 64, 128, 3, 1),      # 16-row tile variant, ragged width
    (2, 9, 11, 128, 64, 3, 1),
    (1, 20, 32, 128, 256, 3, 1),     # 256-cout tile, 2 chunks
    (1, 12, 12, 256, 512, 3, 1),
    (1, 12, 12, 64, 128, 3, 6),      # fc6-style dilation (32-channel chunks)
    (2, 12, 20, 256, 512, 3, 6),     # ... on 256 x 256 weight-gradient tiles: nine pointwise GEMMs by LDS-DMA, ragged rows
    (3, 9, 37, 512, 256, 1, 1),      # 256 x 256 tiles, odd height and width, several images per split
    (2, 8, 8, 128, 128, 1, 1),       # 1x1: 32x32x16 MFMA path
    (1, 10, 34, 256, 64, 1, 1),
    (1, 8, 8, 512, 1024, 1, 1),
    (3, 7, 5, 32, 32, 3, 1),         # 32-channel partial blocks
    (1, 17, 19, 96, 160, 3, 1),      # cin/cout not multiples of 64: 32-wide tiles
    (2, 8, 32, 32, 64, 3, 1),        # 64-cout tile with 32-channel chunks
    (1, 9, 20, 96, 64, 3, 1),
    (1, 8, 16, 96, 128, 1, 1),
    (2, 24, 40, 256, 256, 1, 1),     # pointwise GEMM kernel: 7.5 flat tiles, 256-cout tile
    (1, 16, 32, 64, 64, 1, 1),       # ... 64-cout tile, one K stage
    (1, 32, 32, 1024, 128, 1, 1),    # ... 128-cout tile, 16 K stages
    (2, 100, 130, 64, 64, 3, 1),     # persistent weight-stationary 64-channel kernel: 130 ragged pixel tiles
    (1, 128, 256, 64, 128, 3, 1),    # small-tile 4-wave kernel (conv3x3_w4s<128>), one chunk pair
    (9, 64, 64, 64, 64, 3, 1),       # persistent kernel: more images than tiles per image
    (3, 203, 230, 64, 64, 3, 1),     # ... 624 ragged tiles, 2 or 3 per workgroup
    (2, 16, 40, 128, 64, 3, 1),      # conv3x3_w4s<64>, two chunk pairs
    (2, 40, 64, 256, 128, 3, 1),     # ... <128>, four chunk pairs
    (1, 24, 40, 192, 64, 3, 1),      # ... <64>, three chunk pairs (odd)
    (2, 33, 70, 128, 256, 3, 1),     # 4-wave 3x3 kernel (conv3x3_w4): ragged rows and columns, 2 chunks
    (1, 16, 64, 512, 512, 3, 1),     # ... 8 chunks x 18 steps, two cout tiles
    (3, 8, 32, 64, 256, 3, 1),       # ... one chunk (prologue + tail only), several images
]


def _h(x):
    """round to the library's 16-bit storage type"""
    return torch.from_numpy(np.asarray(x, np.float32)).to(O.STORAGE).float().numpy()


@pytest.mark.parametrize("n,h,w,cin,cout,k,dil", SHAPES)
def test_conv_fwd_dgrad_wgrad(device, n, h, w, cin, cout, k, dil):
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.ops import Workspace
    rng = np.random.default_rng(cin + cout + k + dil)
    x = _h(rng.standard_normal((n, h, w, cin)))
    wt = _h(rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin)))
    dy = _h(rng.standard_normal((n, h, w, cout)) * 0.25)
    # oracle (f32 on f16-exact operands)
    xt = torch.from_numpy(x).requires_grad_(True)
    wtt = torch.from_numpy(wt).requires_grad_(True)
    yo = O.conv2d(xt, wtt, 1, dil)
    yo.backward(torch.from_numpy(dy))
    # device
    xd = torch.from_numpy(x).to(O.STORAGE).to(device)
    dyd = torch.from_numpy(dy).to(O.STORAGE).to(device)
    wm = torch.from_numpy(wt).to(device)                      # HWIO f32 master
    w_kc = torch.empty((k * k, cout, cin), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((k * k, cin, cout), dtype=O.STORAGE, device=device)
    ops.pack_weights(wm, w_kc, w_ck)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, 1, dil)
    d.flags = 0
    y = torch.empty((n, h, w, cout), dtype=O.STORAGE, device=device)
    ops.conv2d(d, xd, w_kc, y)
    pt = dil * (k - 1) - d.pad_top
    pl = dil * (k - 1) - d.pad_left
    dg = ops.ConvDesc(n, h, w, cout, h, w, cin, k, k, 1, dil, pt, pl, 1, 0)
    dx = torch.empty((n, h, w, cin), dtype=O.STORAGE, device=device)
    ops.conv2d(dg, dyd, w_ck, dx)
    dw = torch.zeros((k, k, cin, cout), dtype=torch.float32, device=device)
    ws = Workspace(device, 256 << 20)
    dd = ops.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, dil, d.pad_top, d.pad_left, 0, 0)
    ops.conv2d_wgrad(dd, xd, dyd, dw, ws)
    torch.cuda.synchronize()

    def rel(a, b):
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-20))
    e_y = rel(y.float().cpu().numpy(), yo.detach().numpy())
    e_dx = rel(dx.float().cpu().numpy(), xt.grad.numpy())
    e_dw = rel(dw.cpu().numpy(), wtt.grad.numpy())
    print("%s variant %s: y %.2e dx %.2e dw %.2e" % ((n, h, w, cin, cout, k, dil), ops.conv2d_variant(d), e_y, e_dx, e_dw))
    tol = 8e-3 if O.STORAGE == torch.bfloat16 else 1e-3
    assert e_y < tol and e_dx < tol and e_dw < 5e-6


@pytest.mark.parametrize("n,h,w,c,variant", [(2, 100, 130, 64, "conv_c64_persist_kernel<64>"),
                                             (2, 50, 70, 128, "conv3x3_w4s_kernel<128>"),
                                             (2, 37, 70, 256, "conv3x3_w4_kernel")])
def test_special_kernels_are_selected_and_emit_stats(device, n, h, w, c, variant):
    """The 3x3 layers run the 4-wave kernels (64- / 128-cout tiles: two small workgroups per CU; 256-cout
    tiles: one wave per SIMD with the whole register file); their per-tile batch-norm partials (sum, sum of
    squares of the STORED 16-bit values) add up to the tensor's own sums, and the fused BN-backward
    variant's partials to (sum dz, sum dz*xhat)."""
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(5)
    x = _h(rng.standard_normal((n, h, w, c)))
    wt = _h(rng.standard_normal((3, 3, c, c)) * np.sqrt(2.0 / (9 * c)))
    xd = torch.from_numpy(x).to(O.STORAGE).to(device)
    wm = torch.from_numpy(wt).to(device)
    w_kc = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    ops.pack_weights(wm, w_kc, w_ck)
    d = ops.conv_desc((n, h, w, c), c, 3, 3, 1, 1)
    assert ops.conv2d_variant(d) == variant
    d.flags = ops.CONV_STATS
    T = ops.conv2d_num_mtiles(d)
    y = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    part = torch.zeros((T, 2, c), dtype=torch.float32, device=device)
    ops.conv2d(d, xd, w_kc, y, None, part)
    yf = y.float().cpu().numpy().astype(np.float64)
    got = part.cpu().numpy().astype(np.float64).sum(0)
    assert np.allclose(got[0], yf.sum((0, 1, 2)), rtol=1e-4, atol=1e-2)
    assert np.allclose(got[1], (yf * yf).sum((0, 1, 2)), rtol=1e-4)
    # fused BN-backward reduction: dz = out * [relu(bn(by)) > 0], xhat = (by - mean) * invstd
    by = _h(rng.standard_normal((n, h, w, c)))
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.normal(0, 0.3, c).astype(np.float32)
    mean = rng.normal(0, 0.2, c).astype(np.float32)
    invstd = rng.uniform(0.7, 1.3, c).astype(np.float32)
    d2 = ops.conv_desc((n, h, w, c), c, 3, 3, 1, 1)
    d2.flags = 0
    y2 = torch.empty_like(y)
    part2 = torch.zeros((T, 2, c), dtype=torch.float32, device=device)
    ctx = tuple(torch.from_numpy(a).to(device) for a in (scale, shift, mean, invstd))
    ops.conv2d_bnred(d2, xd, w_kc, y2, part2, (torch.from_numpy(by).to(O.STORAGE).to(device),) + ctx + (True,))
    assert torch.equal(y2, y)
    z = torch.from_numpy(by * scale + shift).to(O.STORAGE).float().numpy()
    dz = yf * (z > 0)
    xh = (by.astype(np.float64) - mean) * invstd
    got2 = part2.cpu().numpy().astype(np.float64).sum(0)
    assert np.allclose(got2[0], dz.sum((0, 1, 2)), rtol=1e-3, atol=5e-2)
    assert np.allclose(got2[1], (dz * xh).sum((0, 1, 2)), rtol=1e-3, atol=5e-2)


@pytest.mark.parametrize("n,h,w", [(2, 100, 130), (9, 64, 64), (2, 129, 97)])
def test_fused_reduction_with_conv1_1_recomputed_equals_reading_it(device, n, h, w):
    """ocr_conv2d_bnred_first_f16 (the persistent 64-channel kernel's epilogue mode 6): conv1_2's input gradient with
    conv1_1's BN-backward sums, conv1_1's y evaluated again from the image in the epilogue instead of read — the
    forward's own MFMA sequence, so output and partial rows are bit-identical to ocr_conv2d_bnred_f16 on the stored y;
    ragged tiles and image borders included.  A shape another kernel would run is refused."""
    from tensorflow_ocr_amd import layers, ops
    from tensorflow_ocr_amd.graph import Graph
    rng = np.random.default_rng(h)
    c = 64
    g = Graph(device, loss_scale=1.0)
    img = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    x4 = layers.prep_images(g, torch.from_numpy(img).to(device)).data
    wt1 = torch.from_numpy((rng.standard_normal((3, 3, 3, c)) * 0.05).astype(np.float32)).to(device)
    wf = torch.empty((3, c, 16), dtype=O.STORAGE, device=device)
    ops.pack_weights_first(wt1, wf)
    y1 = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    ops.conv2d_first(x4, wf, y1)
    # statistics-only launch: same partial rows, nothing stored
    T1 = ops.conv2d_first_num_mtiles(n, h, w)
    pa = torch.zeros((T1, 2, c), dtype=torch.float32, device=device)
    pb = torch.zeros_like(pa)
    y1b = torch.empty_like(y1)
    ops.conv2d_first(x4, wf, y1b, ops.CONV_STATS, None, pa)
    ops.conv2d_first(x4, wf, None, ops.CONV_STATS, None, pb, cout=c)
    assert torch.equal(y1, y1b) and torch.equal(pa, pb) and float(pa.abs().sum()) > 0
    # conv1_2's input gradient
    dy = torch.from_numpy(_h(rng.standard_normal((n, h, w, c)) * 0.1)).to(O.STORAGE).to(device)
    w2 = torch.from_numpy(_h(rng.standard_normal((3, 3, c, c)) * np.sqrt(2.0 / (9 * c)))).to(device)
    w_kc = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    ops.pack_weights(w2, w_kc, w_ck)
    dg = ops.ConvDesc(n, h, w, c, h, w, c, 3, 3, 1, 1, 1, 1, 1, 0)
    assert ops.conv2d_variant(dg) == "conv_c64_persist_kernel<64>"
    T = ops.conv2d_num_mtiles(dg)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).to(device)
    scale[5] = -0.8
    ctx = (scale,) + tuple(torch.from_numpy(a.astype(np.float32)).to(device) for a in (
        rng.normal(0, 0.3, c), rng.normal(0, 0.2, c), rng.uniform(0.7, 1.3, c)))
    dx_r, dx_f = torch.empty_like(y1), torch.empty_like(y1)
    part_r = torch.zeros((T, 2, c), dtype=torch.float32, device=device)
    part_f = torch.zeros_like(part_r)
    ops.conv2d_bnred(dg, dy, w_ck, dx_r, part_r, (y1,) + ctx + (True,))
    ops.conv2d_bnred_first(dg, dy, w_ck, dx_f, part_f, (x4, wf) + ctx + (True,))
    torch.cuda.synchronize()
    assert torch.equal(dx_r, dx_f)
    assert float(part_r.abs().sum()) > 0 and torch.equal(part_r, part_f), float((part_r - part_f).abs().max())
    d128 = ops.ConvDesc(n, h, w, 128, h, w, 64, 3, 3, 1, 1, 1, 1, 1, 0)
    from tensorflow_ocr_amd._lib import OcrHipError
    with pytest.raises(OcrHipError):
        ops.conv2d_bnred_first(d128, dy, w_ck, dx_f, part_f, (x4, wf) + ctx + (True,))


@pytest.mark.parametrize("n,h,w", [(2, 100, 130), (9, 64, 64), (2, 129, 97), (4, 256, 256)])
def test_first_layer_weight_gradient_from_sums_equals_the_pass_over_the_gradient(device, n, h, w):
    """ocr_conv2d_bnred_first_wgrad_f16 (epilogue mode 7: conv1_2's input gradient is NOT stored, the launch leaves
    S1 = V^T dz) + ocr_conv2d_first_moments_keep_f16 + ocr_conv2d_first_wgrad_sums_f32 against the pass-by-pass form
    (ocr_conv2d_bnred_first_f16 stores the gradient, ocr_conv2d_first_wgrad_bn_f16 reads it): the partial rows are
    bit-identical; dW agrees within the 16-bit rounding of dy that only the pass-by-pass form applies, and with float64
    on the host from the stored gradient.  Ragged tiles, image borders and a negative scale included."""
    from tensorflow_ocr_amd import layers, ops
    from tensorflow_ocr_amd.graph import Graph
    rng = np.random.default_rng(h + n)
    c = 64
    g = Graph(device, loss_scale=1.0)
    img = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    x4 = layers.prep_images(g, torch.from_numpy(img).to(device)).data
    wt1 = torch.from_numpy((rng.standard_normal((3, 3, 3, c)) * 0.05).astype(np.float32)).to(device)
    wf = torch.empty((3, c, 16), dtype=O.STORAGE, device=device)
    ops.pack_weights_first(wt1, wf)
    dy = torch.from_numpy(_h(rng.standard_normal((n, h, w, c)) * 0.1)).to(O.STORAGE).to(device)
    w2 = torch.from_numpy(_h(rng.standard_normal((3, 3, c, c)) * np.sqrt(2.0 / (9 * c)))).to(device)
    w_kc = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    ops.pack_weights(w2, w_kc, w_ck)
    dg = ops.ConvDesc(n, h, w, c, h, w, c, 3, 3, 1, 1, 1, 1, 1, 0)
    T = ops.conv2d_num_mtiles(dg)
    # conv1_1's batch norm: its real statistics (so that the mask cuts about half) with a negative gamma in one channel
    ws = ops.Workspace(device, 16 << 20)
    row = torch.zeros((1, 2, c), dtype=torch.float32, device=device)
    moments = torch.zeros((32 * 32,), dtype=torch.float64, device=device)
    ops.conv2d_first_moments(x4, wf, row, c, ws, moments=moments)
    cnt = float(n * h * w)
    r = row.cpu().numpy().astype(np.float64)[0]
    mean = r[0] / cnt
    var = np.maximum(r[1] / cnt - mean * mean, 0.0)
    invstd = 1.0 / np.sqrt(var + 1e-5)
    gamma = rng.uniform(0.5, 1.5, c)
    gamma[5] = -0.8
    beta = rng.normal(0, 0.2, c)
    f = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(device)
    scale, shift, mu, istd = f(gamma * invstd), f(beta - mean * gamma * invstd), f(mean), f(invstd)
    ctx = (scale, shift, mu, istd)
    dx = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    part_r = torch.zeros((T, 2, c), dtype=torch.float32, device=device)
    part_s = torch.zeros_like(part_r)
    ops.conv2d_bnred_first(dg, dy, w_ck, dx, part_r, (x4, wf) + ctx + (True,))
    blocks = ops.conv2d_bnred_first_wgrad_blocks(dg)
    assert blocks > 0
    s1 = torch.full((blocks, 32, 64), float("nan"), dtype=torch.float32, device=device)
    ops.conv2d_bnred_first_wgrad(dg, dy, w_ck, part_s, (x4, wf) + ctx + (True,), s1)
    torch.cuda.synchronize()
    assert float(part_r.abs().sum()) > 0 and torch.equal(part_r, part_s)
    # coefficients (any values: the identity is linear in them; A = the layer's scale, which the mask reads)
    coef = (scale, f(rng.normal(0, 0.05, c)), f(rng.normal(0, 0.05, c)))
    dw_p = torch.zeros((3, 3, 3, c), dtype=torch.float32, device=device)
    dw_s = torch.full_like(dw_p, float("nan"))
    ops.conv2d_first_wgrad_bn(x4, dx, None, shift, coef, True, dw_p, ws, w_first=wf)
    ops.conv2d_first_wgrad_sums(s1, moments, wf, coef, dw_s)
    torch.cuda.synchronize()
    # float64 on the host from the stored gradient and the stored-precision y
    y1 = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    ops.conv2d_first(x4, wf, y1)
    yv = y1.float().cpu().numpy().astype(np.float64)
    dxv = dx.float().cpu().numpy().astype(np.float64)
    A, B, C = (t.cpu().numpy().astype(np.float64) for t in coef)
    sh = shift.cpu().numpy().astype(np.float64)
    dzv = np.where(yv * A + sh > 0, dxv, 0.0)       # (elements within a rounding of the threshold carry no weight here)
    dyv = A * dzv + B * yv + C
    xi = x4.float().cpu().numpy().astype(np.float64)[..., :3]
    xp = np.pad(xi, ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = np.zeros((3, 3, 3, c))
    for ky in range(3):
        for kx in range(3):
            ref[ky, kx] = np.einsum("nhwc,nhwd->cd", xp[:, ky:ky + h, kx:kx + w], dyv)
    got_s = dw_s.cpu().numpy().astype(np.float64)
    got_p = dw_p.cpu().numpy().astype(np.float64)
    assert np.isfinite(got_s).all()
    tol = 2e-3 * np.abs(ref).max()
    assert np.abs(got_s - ref).max() <= tol, (np.abs(got_s - ref).max(), np.abs(ref).max())
    assert np.abs(got_p - ref).max() <= tol, (np.abs(got_p - ref).max(), np.abs(ref).max())
    # a shape another kernel would run is refused
    d128 = ops.ConvDesc(n, h, w, 128, h, w, 64, 3, 3, 1, 1, 1, 1, 1, 0)
    assert ops.conv2d_bnred_first_wgrad_blocks(d128) < 0
    from tensorflow_ocr_amd._lib import OcrHipError
    with pytest.raises(OcrHipError):
        ops.conv2d_bnred_first_wgrad(d128, dy, w_ck, part_s, (x4, wf) + ctx + (True,), s1)


def test_batched_weight_repack_equals_per_layer(device):
    """ocr_pack_weights_batch_f16 (every conv layer's two operand layouts in one launch, after the optimiser step)
    writes exactly what ocr_pack_weights_f16 writes layer by layer — including ragged 32-tiles and a NULL layout."""
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(7)
    shapes = [(3, 3, 64, 64), (1, 1, 96, 160), (3, 3, 40, 24), (2, 2, 256, 64), (1, 1, 1024, 512)]
    ws = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(device) for s in shapes]
    ref, ents = [], []
    for i, w in enumerate(ws):
        kh, kw, cin, cout = w.shape
        a = torch.empty((kh * kw, cout, cin), dtype=O.STORAGE, device=device)
        b = torch.empty((kh * kw, cin, cout), dtype=O.STORAGE, device=device)
        ops.pack_weights(w, a, b)
        ref.append((a, b))
        a2 = torch.zeros_like(a)
        b2 = None if i == 2 else torch.zeros_like(b)          # one layer without the [tap][cin][cout] layout
        ents.append((w, a2, b2))
    pb = ops.PackBatch(ents, torch.device(device))
    pb.run()
    torch.cuda.synchronize()
    for (a, b), (_, a2, b2) in zip(ref, ents):
        assert torch.equal(a, a2)
        if b2 is not None:
            assert torch.equal(b, b2)


@pytest.mark.parametrize("n,h,w,cin,cout,with_sub", [(2, 12, 16, 64, 128, False), (2, 12, 16, 64, 256, True),
                                                     (1, 9, 9, 64, 64, True), (3, 10, 32, 128, 512, False)])
def test_bottleneck_tail_conv_vs_numpy(device, n, h, w, cin, cout, with_sub):
    """ocr_conv2d_bnred_tail_f16 (1x1 input-gradient conv completing a bottleneck output's gradient): stored value =
    [tail_out > 0] * (conv + what was there [+ gradient of the stride-2 subsample at the even positions]), partial
    rows summing to (sum dz, sum dz * xhat(bn_y)) — against a float64 restatement on the same 16-bit operands.  Flat
    256-pixel tiles (pixel count a multiple of 32: the batched epilogue) and the generic tile kernel (9 x 9)."""
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(n * 100 + cout)
    x = _h(rng.standard_normal((n, h, w, cin)) * 0.5)
    wt = _h(rng.standard_normal((1, 1, cin, cout)) * np.sqrt(2.0 / cin))
    old = _h(rng.standard_normal((n, h, w, cout)) * 0.3)
    out = _h(rng.standard_normal((n, h, w, cout)))                    # the bottleneck output (mask = out > 0)
    by = _h(rng.standard_normal((n, h, w, cout)))
    mean = rng.normal(0, 0.2, cout).astype(np.float32)
    invstd = rng.uniform(0.7, 1.3, cout).astype(np.float32)
    sh, sw = (h + 1) // 2, (w + 1) // 2
    sub = _h(rng.standard_normal((n, sh, sw, cout)) * 0.3)
    dev = lambda a: torch.from_numpy(a).to(O.STORAGE).to(device)
    wm = torch.from_numpy(wt).to(device)
    w_kc = torch.empty((1, cout, cin), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((1, cin, cout), dtype=O.STORAGE, device=device)
    ops.pack_weights(wm, w_kc, w_ck)
    d = ops.conv_desc((n, h, w, cin), cout, 1, 1, 1, 1)
    d.flags = ops.CONV_ACCUM_F16
    T = ops.conv2d_num_mtiles(d)
    y = dev(old).clone()
    part = torch.zeros((T, 2, cout), dtype=torch.float32, device=device)
    ctx = (dev(by), torch.from_numpy(mean).to(device), torch.from_numpy(invstd).to(device), dev(out))
    ops.conv2d_bnred_tail(d, dev(x), w_kc, y, part, ctx, dev(sub) if with_sub else None)
    torch.cuda.synchronize()
    # the same launch with the ReLU mask as BITS (one byte per 8 channels, bit e = out[.. + e] > 0; OCR_RESNET_MASK_BITS=1
    # path: measured slower and off by default, kept correct): identical stores and partial sums
    bits = torch.from_numpy(np.packbits((out > 0).reshape(-1, 8), axis=1, bitorder="little").reshape(-1)).to(device)
    y_b = dev(old).clone()
    part_b = torch.zeros_like(part)
    ops.conv2d_bnred_tail(d, dev(x), w_kc, y_b, part_b, ctx + (bits,), dev(sub) if with_sub else None)
    torch.cuda.synchronize()
    assert torch.equal(y_b, y) and torch.equal(part_b, part)
    # restatement: 16-bit roundings where the kernel rounds (conv result, + old, + sub)
    conv = _h(x.reshape(-1, cin).astype(np.float64) @ wt.reshape(cin, cout).astype(np.float64)).reshape(n, h, w, cout)
    tot = _h(conv + old)
    if with_sub:
        z = np.zeros_like(tot)
        z[:, ::2, ::2, :] = sub
        tot = _h(tot + z)
    dz = np.where(out > 0, tot, 0.0).astype(np.float32)
    got = y.float().cpu().numpy()
    tol = 8e-3 if O.STORAGE == torch.bfloat16 else 1.5e-3
    assert np.abs(got - dz).max() <= tol * max(1.0, np.abs(dz).max())
    assert ((got == 0) == (dz == 0))[out <= 0].all()                  # the mask is exact
    xh = (by.astype(np.float64) - mean) * invstd
    sums = part.cpu().numpy().astype(np.float64).sum(0)
    gd = got.astype(np.float64)                                        # sums are of the STORED values
    assert np.allclose(sums[0], gd.sum((0, 1, 2)), rtol=1e-4, atol=2e-2)
    assert np.allclose(sums[1], (gd * xh).sum((0, 1, 2)), rtol=1e-3, atol=5e-2)


@pytest.mark.parametrize("n,h,w,c,k", [(2, 16, 40, 128, 3), (2, 100, 130, 64, 3), (2, 37, 70, 256, 3), (2, 12, 16, 128, 1)])
def test_fused_reduction_relu_mask_at_the_rounding_boundary(device, n, h, w, c, k):
    """The fused BN-backward reductions decide the ReLU mask of the STORED activation by `z > half the smallest
    subnormal` instead of rounding z to 16 bits (common.h: OCR_RELU_TIE).  With bn_y = 0, scale = 1 the pre-activation
    is the shift itself: channels whose shift is tie/2 or exactly the tie round to 0 (masked), a hair above the tie
    or twice it round to the smallest subnormal (pass) — on every kernel family that carries the reduction."""
    from tensorflow_ocr_amd import ops
    tie = 2.0 ** -134 if O.STORAGE == torch.bfloat16 else 2.0 ** -25
    rng = np.random.default_rng(3)
    x = _h(rng.standard_normal((n, h, w, c)))
    wt = _h(rng.standard_normal((k, k, c, c)) * np.sqrt(2.0 / (k * k * c)))
    xd = torch.from_numpy(x).to(O.STORAGE).to(device)
    w_kc = torch.empty((k * k, c, c), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((k * k, c, c), dtype=O.STORAGE, device=device)
    ops.pack_weights(torch.from_numpy(wt).to(device), w_kc, w_ck)
    d = ops.conv_desc((n, h, w, c), c, k, k, 1, 1)
    d.flags = 0
    T = ops.conv2d_num_mtiles(d)
    y = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    part = torch.zeros((T, 2, c), dtype=torch.float32, device=device)
    shift = np.array([tie / 2, tie, tie * (1 + 2.0 ** -10), 2 * tie] * (c // 4), np.float32)
    expect_pass = np.array([False, False, True, True] * (c // 4))
    # the 16-bit conversion this build performs agrees with the rule
    conv16 = torch.from_numpy(shift).to(device).to(O.STORAGE).float().cpu().numpy() > 0
    assert (conv16 == expect_pass).all()
    ctx = (torch.zeros((n, h, w, c), dtype=O.STORAGE, device=device), torch.ones(c, device=device),
           torch.from_numpy(shift).to(device), torch.zeros(c, device=device), torch.ones(c, device=device), True)
    ops.conv2d_bnred(d, xd, w_kc, y, part, ctx)
    torch.cuda.synchronize()
    s0 = part.cpu().numpy().astype(np.float64).sum(0)[0]
    full = y.float().cpu().numpy().astype(np.float64).sum((0, 1, 2))
    assert np.allclose(s0[expect_pass], full[expect_pass], rtol=1e-4, atol=2e-2)
    assert (s0[~expect_pass] == 0).all()


# ------------------------------------------------------------------ round 3: operand transforms inside the 1x1 kernels
@pytest.mark.parametrize("n,h,w,cin,cout,proj", [
    (2, 16, 24, 256, 64, False),      # conv1 of a stage-1 unit: 64-cout tile, four K stages, three flat tiles
    (1, 20, 32, 512, 128, True),      # 128-cout tile, projection shortcut's BN folded in
    (2, 8, 20, 1024, 256, False),     # 256-cout tile, 1.25 flat tiles (ragged last tile)
    (1, 8, 8, 2048, 512, False),      # two cout tiles: only the first writes x_out / bits
    (3, 5, 32, 128, 64, True),        # two K stages
])
def test_pw_bnaddrelu_equals_the_two_pass_form(device, n, h, w, cin, cout, proj):
    """ocr_conv2d_pw_bnaddrelu_f16 (the previous unit's relu(bn(y3) + shortcut) computed while the 1x1 convolution
    loads its operand) against ocr_bn_add_relu_f16 followed by ocr_conv2d_f16: the written-back operand and its mask
    bits equal the element-wise kernel's bit for bit (same f32 expression, same roundings), and so do the convolution
    output and its BN partial sums (same MFMA order on the same operand); the element-wise kernel itself against numpy."""
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(cin + cout)
    y3 = _h(rng.standard_normal((n, h, w, cin)))
    short = _h(rng.standard_normal((n, h, w, cin)))
    scale = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    shift = rng.normal(0, 0.3, cin).astype(np.float32)
    ssc = rng.uniform(0.5, 1.5, cin).astype(np.float32) if proj else None
    ssh = rng.normal(0, 0.3, cin).astype(np.float32) if proj else None
    wt = _h(rng.standard_normal((1, 1, cin, cout)) * np.sqrt(2.0 / cin))
    dev = lambda a: torch.from_numpy(a).to(O.STORAGE).to(device)
    f32 = lambda a: None if a is None else torch.from_numpy(a).to(device)
    w_kc = torch.empty((1, cout, cin), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((1, cin, cout), dtype=O.STORAGE, device=device)
    ops.pack_weights(torch.from_numpy(wt).to(device), w_kc, w_ck)
    d = ops.conv_desc((n, h, w, cin), cout, 1, 1, 1, 1)
    d.flags = ops.CONV_STATS
    T = ops.conv2d_num_mtiles(d)
    # two passes
    x_ref = torch.empty((n, h, w, cin), dtype=O.STORAGE, device=device)
    bits_ref = torch.zeros((n * h * w * cin // 8,), dtype=torch.uint8, device=device)
    ops.bn_add_relu(dev(y3), f32(scale), f32(shift), dev(short), x_ref, f32(ssc), f32(ssh), bits_ref)
    y_ref = torch.empty((n, h, w, cout), dtype=O.STORAGE, device=device)
    part_ref = torch.zeros((T, 2, cout), dtype=torch.float32, device=device)
    ops.conv2d(d, x_ref, w_kc, y_ref, None, part_ref)
    # one
    x_out = torch.full((n, h, w, cin), float("nan"), dtype=O.STORAGE, device=device)
    bits = torch.full((n * h * w * cin // 8,), 0xAA, dtype=torch.uint8, device=device)
    y = torch.empty((n, h, w, cout), dtype=O.STORAGE, device=device)
    part = torch.zeros((T, 2, cout), dtype=torch.float32, device=device)
    ops.conv2d_pw_bnaddrelu(d, dev(y3), f32(scale), f32(shift), dev(short), f32(ssc), f32(ssh), x_out, bits, w_kc, y, part)
    torch.cuda.synchronize()
    dxo = (x_out.float() - x_ref.float())
    assert torch.equal(x_out, x_ref), ("x_out", int(torch.isnan(x_out.float()).sum()), int((dxo != 0).sum()), float(dxo.abs().nan_to_num().max()),
                                        (dxo != 0).nonzero()[:6].tolist())
    assert torch.equal(bits, bits_ref), ("bits", int((bits != bits_ref).sum()))
    assert torch.equal(y, y_ref), ("y", float((y.float() - y_ref.float()).abs().max()))
    assert torch.equal(part, part_ref)
    # the element-wise definition (nets/resnet_v1.py:107), roundings where the kernels round
    z = _h(y3 * scale + shift)
    r = _h(short * ssc + ssh) if proj else short
    want = _h(np.maximum(z + r, 0.0))
    got = x_ref.float().cpu().numpy()
    tol = 8e-3 if O.STORAGE == torch.bfloat16 else 1e-3
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
    b = np.unpackbits(bits_ref.cpu().numpy(), bitorder="little").reshape(got.shape)
    assert ((got > 0) == (b == 1)).all()


@pytest.mark.parametrize("n,h,w,c4,c", [(2, 16, 24, 256, 64), (1, 20, 32, 512, 128), (2, 8, 20, 1024, 256), (1, 8, 8, 2048, 512)])
def test_pw_bnbwd_bnred_equals_the_two_pass_form(device, n, h, w, c4, c):
    """ocr_bn_bwd_coefficients + ocr_conv2d_pw_bnbwd_bnred_f16 (conv3's batch-norm backward apply computed while its
    input-gradient convolution loads dy) against ocr_bn_relu_bwd_apply_f16 + ocr_conv2d_bnred_f16 and against float64:
    dgamma / dbeta identical (same reduction), dy within one 16-bit rounding (the affine form A*dz + B*y + C groups the
    f32 operations differently), dx and the fused reduction's partial sums accordingly."""
    from tensorflow_ocr_amd import ops
    from tensorflow_ocr_amd.ops import Workspace
    rng = np.random.default_rng(c4 + c)
    npix = n * h * w
    dz = _h(rng.standard_normal((n, h, w, c4)) * 0.3)
    y3 = _h(rng.standard_normal((n, h, w, c4)))
    gamma = rng.uniform(0.5, 1.5, c4).astype(np.float32)
    mean = rng.normal(0, 0.2, c4).astype(np.float32)
    invstd = rng.uniform(0.7, 1.3, c4).astype(np.float32)
    scale = (gamma * invstd).astype(np.float32)
    shift = rng.normal(0, 0.3, c4).astype(np.float32)
    xh = (y3.astype(np.float64) - mean) * invstd
    s1 = dz.astype(np.float64).sum((0, 1, 2))
    s2 = (dz.astype(np.float64) * xh).sum((0, 1, 2))
    partial = torch.from_numpy(np.stack([s1, s2]).astype(np.float32)[None]).to(device)       # T = 1
    wt = _h(rng.standard_normal((1, 1, c, c4)) * np.sqrt(2.0 / c))                             # conv3: c -> c4
    y2 = _h(rng.standard_normal((n, h, w, c)))                                                  # conv2's raw output (below)
    bsc = rng.uniform(0.5, 1.5, c).astype(np.float32)
    bsh = rng.normal(0, 0.3, c).astype(np.float32)
    bmu = rng.normal(0, 0.2, c).astype(np.float32)
    bis = rng.uniform(0.7, 1.3, c).astype(np.float32)
    dev = lambda a: torch.from_numpy(a).to(O.STORAGE).to(device)
    f32 = lambda a: torch.from_numpy(a).to(device)
    w_kc = torch.empty((1, c4, c), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((1, c, c4), dtype=O.STORAGE, device=device)
    ops.pack_weights(torch.from_numpy(wt).to(device), w_kc, w_ck)
    ws = Workspace(device, 64 << 20)
    dg = ops.ConvDesc(n, h, w, c4, h, w, c, 1, 1, 1, 1, 0, 0, 1, 0)
    Tm = ops.conv2d_num_mtiles(dg)
    ctx = (dev(y2), f32(bsc), f32(bsh), f32(bmu), f32(bis), True)
    # two passes
    dgam_r, dbet_r = torch.zeros(c4, device=device), torch.zeros(c4, device=device)
    dy_r = torch.empty((n, h, w, c4), dtype=O.STORAGE, device=device)
    ops.bn_relu_bwd_apply(dev(y3), f32(scale), f32(shift), f32(mean), f32(invstd), dev(dz), False, partial, 1, dgam_r, dbet_r, dy_r, ws)
    dx_r = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    part_r = torch.zeros((Tm, 2, c), dtype=torch.float32, device=device)
    ops.conv2d_bnred(dg, dy_r, w_ck, dx_r, part_r, ctx)
    # one
    dgam, dbet = torch.zeros(c4, device=device), torch.zeros(c4, device=device)
    coef = tuple(torch.empty(c4, device=device) for _ in range(3))
    ops.bn_bwd_coefficients(partial, 1, c4, float(npix), f32(scale), f32(mean), f32(invstd), dgam, dbet, coef, ws)
    dy = torch.full((n, h, w, c4), float("nan"), dtype=O.STORAGE, device=device)
    dx = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    part = torch.zeros((Tm, 2, c), dtype=torch.float32, device=device)
    ops.conv2d_pw_bnbwd_bnred(dg, dev(dz), dev(y3), coef, dy, w_ck, dx, part, ctx)
    torch.cuda.synchronize()
    assert torch.equal(dgam, dgam_r) and torch.equal(dbet, dbet_r)
    # float64 definition of the apply step
    want = scale * (dz - s1 / npix - xh * (s2 / npix))
    tol = 8e-3 if O.STORAGE == torch.bfloat16 else 1e-3
    m = np.abs(want).max()
    got, got_r = dy.float().cpu().numpy(), dy_r.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= tol * m and np.abs(got_r - want).max() <= tol * m
    # the convolution of the SAME operand is the same convolution: run the plain kernel on the fused kernel's dy
    dx_c = torch.empty_like(dx)
    part_c = torch.zeros_like(part)
    ops.conv2d_bnred(dg, dy, w_ck, dx_c, part_c, ctx)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_c) and torch.equal(part, part_c)
    dxf, dxr = dx.float().cpu().numpy(), dx_r.float().cpu().numpy()
    assert np.abs(dxf - dxr).max() <= 2 * tol * np.abs(dxr).max()


@pytest.mark.parametrize("n,h,w", [(2, 100, 130), (3, 203, 230), (9, 64, 64), (2, 129, 97)])
def test_conv_relu_pool_equals_the_two_pass_form(device, n, h, w):
    """ocr_conv2d_relu_pool_f16 (64 -> 64 channel 3x3 conv + bias + ReLU + 2x2/2 max-pool in the persistent kernel's
    epilogue, only the pooled tile written) against ocr_conv2d_f16 followed by ocr_maxpool_f16: pooled activations and
    first-max positions bit for bit — even and odd map sizes (SAME pooling ignores the missing row / column), ragged
    tiles, more images than tiles per image."""
    from tensorflow_ocr_amd import ops
    rng = np.random.default_rng(h + w)
    c = 64
    x = _h(rng.standard_normal((n, h, w, c)))
    wt = _h(rng.standard_normal((3, 3, c, c)) * np.sqrt(2.0 / (9 * c)))
    bias = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32)).to(device)
    xd = torch.from_numpy(x).to(O.STORAGE).to(device)
    w_kc = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    w_ck = torch.empty((9, c, c), dtype=O.STORAGE, device=device)
    ops.pack_weights(torch.from_numpy(wt).to(device), w_kc, w_ck)
    d = ops.conv_desc((n, h, w, c), c, 3, 3, 1, 1)
    assert ops.conv2d_variant(d) == "conv_c64_persist_kernel<64>"
    d.flags = ops.CONV_BIAS | ops.CONV_RELU
    a = torch.empty((n, h, w, c), dtype=O.STORAGE, device=device)
    ops.conv2d(d, xd, w_kc, a, bias)
    ph, pw = (h + 1) // 2, (w + 1) // 2
    p_ref = torch.empty((n, ph, pw, c), dtype=O.STORAGE, device=device)
    i_ref = torch.zeros((n, ph, pw, c), dtype=torch.uint8, device=device)
    ops.maxpool(a, 2, 2, (0, 0), p_ref, i_ref)
    p = torch.full((n, ph, pw, c), float("nan"), dtype=O.STORAGE, device=device)
    i = torch.full((n, ph, pw, c), 0xEE, dtype=torch.uint8, device=device)
    ops.conv2d_relu_pool(d, xd, w_kc, bias, p, i)
    torch.cuda.synchronize()
    assert torch.equal(p, p_ref), (int(torch.isnan(p.float()).sum()), int((p != p_ref).sum()))
    assert torch.equal(i, i_ref), int((i != i_ref).sum())
    # and without the index tensor (inference)
    p2 = torch.empty_like(p)
    ops.conv2d_relu_pool(d, xd, w_kc, bias, p2, None)
    torch.cuda.synchronize()
    assert torch.equal(p2, p_ref)
