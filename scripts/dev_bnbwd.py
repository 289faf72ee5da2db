"""Dev: HBM rate of the BN-backward apply pass (bn_relu_bwd_kernel<1>) and of the forward apply (bn_relu_kernel)."""
import sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import ops
from tensorflow_ocr_amd.ops import Workspace
dev = 'cuda'
for n, hw, c in [(32, 128, 256), (32, 64, 512), (32, 256, 128), (32, 512, 64)]:
    y = torch.randn(n, hw, hw, c, device=dev).half(); da = torch.randn(n, hw, hw, c, device=dev).half()
    dy = torch.empty_like(y); a = torch.empty_like(y)
    v = [torch.rand(c, device=dev) + 0.5 for _ in range(4)]
    T = 64
    part = torch.randn(T, 2, c, device=dev); dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
    ws = Workspace(torch.device(dev))
    fb = lambda: ops.bn_relu_bwd_apply(y, v[0], v[1], v[2], v[3], da, True, part, T, dg, db, dy, ws)
    ff = lambda: ops.bn_relu(y, v[0], v[1], True, 0, a, None)
    res = []
    for f, nb in ((fb, 3), (ff, 2)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        best = 1e9
        for r in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        res.append('%.1f us %4.0f GB/s' % (best * 1e3, nb * y.numel() * 2 / best / 1e6))
    print('%dx%dx%dx%d  bwd apply (+finalize) %s | fwd apply %s' % (n, hw, hw, c, res[0], res[1]), flush=True)
