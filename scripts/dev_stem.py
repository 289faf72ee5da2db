"""Dev-only: time the ResNet stem conv (7x7/2, 3->64) forward and weight gradient at 64 x 640^2."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_ocr_amd import _lib as L, ops
dev = "cuda"
n, h, w, cout = 64, 640, 640, 64
x4 = torch.randn(n, h, w, 4, device=dev).half()
oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
dy = torch.randn(n, oh, ow, cout, device=dev).half()
dw = torch.empty(7, 7, 3, cout, device=dev)
ws = ops.Workspace(torch.device(dev), 256 << 20)
def t(f, it=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print(os.environ.get("OCR_STEM_BLOCKS", "256"), "stem wgrad %.3f ms" % t(lambda: ops.conv2d_stem_wgrad(x4, dy, dw, ws)))
n, h, w = 32, 512, 512
x4 = torch.randn(n, h, w, 4, device=dev).half(); dy = torch.randn(n, h, w, 64, device=dev).half(); dw = torch.empty(3, 3, 3, 64, device=dev)
print(os.environ.get("OCR_FIRST_BLOCKS", "256"), "first wgrad %.3f ms" % t(lambda: ops.conv2d_first_wgrad(x4, dy, dw, ws)))
