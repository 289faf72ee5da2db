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
