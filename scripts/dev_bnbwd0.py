"""Dev: the BN-backward REDUCTION pass (bn_relu_bwd_kernel<0>) = full backward minus (finalize + apply)."""
import sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import ops
from tensorflow_ocr_amd.ops import Workspace
dev = 'cuda'
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best
for n, hw, c in [(64, 160, 256), (64, 80, 512), (64, 160, 64), (32, 128, 256), (32, 32, 1024)]:
    y = torch.randn(n, hw, hw, c, device=dev).half(); da = torch.randn(n, hw, hw, c, device=dev).half()
    dy = torch.empty_like(y)
    v = [torch.rand(c, device=dev) + 0.5 for _ in range(4)]
    T = 64
    part = torch.randn(T, 2, c, device=dev); dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
    ws = Workspace(torch.device(dev))
    full = t(lambda: ops.bn_relu_bwd(y, v[0], v[1], v[2], v[3], da, None, False, 0, dg, db, dy, ws))
    app = t(lambda: ops.bn_relu_bwd_apply(y, v[0], v[1], v[2], v[3], da, False, part, T, dg, db, dy, ws))
    red = full - app
    print('%dx%dx%dx%d  reduction pass %.1f us = %4.0f GB/s | apply (+finalize) %.1f us' % (n, hw, hw, c, red * 1e3, 2 * y.numel() * 2 / red / 1e6, app * 1e3), flush=True)
