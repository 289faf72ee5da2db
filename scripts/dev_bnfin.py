"""Dev: time ocr_bn_finalize over (partial rows, channels); OCR_BN_ROWS forces the rows-per-block."""
import ctypes, os, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L, ops
dev = 'cuda'
out = []
for T, C in [(6400, 64), (6400, 256), (1600, 512), (2048, 256), (8192, 128), (2048, 64), (400, 1024), (512, 512)]:
    part = torch.randn(T, 2, C, device=dev)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
    o = [torch.empty(C, device=dev) for _ in range(4)]
    stage = torch.empty(ops.bn_reduce_workspace(T, C), dtype=torch.uint8, device=dev)
    f = lambda: ops.bn_finalize(part, T, C, 1e6, g, b, 1e-5, 0.997, mm, mv, o[0], o[1], o[2], o[3], stage)
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    out.append('%dx%d %.1f' % (T, C, best * 1e3))
print('rows=%s | ' % os.environ.get('OCR_BN_ROWS', 'auto') + ' | '.join(out) + '  (us, back-to-back launches)')
