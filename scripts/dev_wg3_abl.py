"""Dev: ablation variants of wgrad3_kernel (devlibs/libabl_<mask>.so, -DWG3_ABL=mask -DOCR_DIAG_CLOCK) in one process."""
import ctypes, glob, os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
libs = {}
for f in sorted(glob.glob('devlibs/libabl_*.so'), key=lambda s: int(s.split('_')[-1].split('.')[0])):
    libs[os.path.basename(f)[7:-3]] = ctypes.CDLL(os.path.abspath(f))
B = 32
def case(hw, cin, cout):
    dev = 'cuda'
    x = torch.randn(B, hw, hw, cin, device=dev).half(); dy = (torch.randn(B, hw, hw, cout, device=dev) * 0.1).half()
    d = L.ConvDesc(B, hw, hw, cin, hw, hw, cout, 3, 3, 1, 1, 1, 1, 0, 0)
    l0 = next(iter(libs.values()))
    l0.ocr_conv2d_wgrad_workspace.restype = ctypes.c_size_t
    nbytes = l0.ocr_conv2d_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev); dw = torch.empty(3, 3, cin, cout, device=dev)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def f(lib):
        assert lib.ocr_conv2d_wgrad_f16(ctypes.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), ctypes.c_size_t(nbytes), sp) == 0
    res = {k: [] for k in libs}
    for k, lib in libs.items():
        for _ in range(5): f(lib)
    torch.cuda.synchronize()
    for r in range(4):
        for k, lib in libs.items():
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f(lib)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * B * hw * hw * cout * cin * 9
    for k, lib in libs.items():
        buf = (ctypes.c_ulonglong * (2 * 256))()
        lib.ocr_diag_read_wgrad(buf, ctypes.c_int(256))
        a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2).astype(np.float64)
        ok = a[:, 1] > 0
        ms = float(np.median(res[k]))
        print("%d:%d>%d abl=%-3s %.3f ms %5.0f TF | main loop %8.0f cyc clock %.2f GHz" % (
            hw, cin, cout, k, ms, fl / ms / 1e9, np.median(a[ok, 0]), np.median(a[ok, 0] / a[ok, 1] * 0.1)), flush=True)
for a in sys.argv[1:] or ["64,512,512"]:
    case(*[int(v) for v in a.split(",")])
