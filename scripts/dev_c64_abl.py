"""Dev: ablation builds of conv_c64_persist_kernel (devlibs/libabl_<mask>.so, -DC64_ABL=mask) in one process."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
libs = {}
for f in sorted(glob.glob('devlibs/libabl_*.so'), key=lambda s: int(s.split('_')[-1].split('.')[0])):
    libs[os.path.basename(f)[7:-3]] = ctypes.CDLL(os.path.abspath(f))
B, hw, c = 32, 512, 64
dev = 'cuda'
x = torch.randn(B, hw, hw, c, device=dev).half(); w = (torch.randn(9, c, c, device=dev) * 0.05).half()
y = torch.empty(B, hw, hw, c, dtype=torch.half, device=dev)
d = L.ConvDesc(B, hw, hw, c, hw, hw, c, 3, 3, 1, 1, 1, 1, 0, 0)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def f(lib):
    assert lib.ocr_conv2d_f16(ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(None), L.ptr(y), L.ptr(None), sp) == 0
res = {k: [] for k in libs}
for k, lib in libs.items():
    for _ in range(3): f(lib)
torch.cuda.synchronize()
for r in range(4):
    for k, lib in libs.items():
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f(lib)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 10)
for k in libs:
    print('abl=%-2s %.3f ms' % (k, min(res[k])), flush=True)
