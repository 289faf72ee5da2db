"""Dev: ablation variants of conv3x3_w4_kernel (devlibs/libabl_<mask>.so, -DW4_ABL=mask -DOCR_DIAG_CLOCK),
all loaded in ONE process, interleaved rounds; prints wall ms, TFLOP/s-equivalent, main-loop cycles and
in-kernel clock per variant.  Ablated variants compute garbage: only their timing means anything."""
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
    x = torch.randn(B, hw, hw, cin, device=dev).half(); w = (torch.randn(9, cout, cin, device=dev) * 0.05).half()
    d = L.ConvDesc(B, hw, hw, cin, hw, hw, cout, 3, 3, 1, 1, 1, 1, 0, L.CONV_STATS)
    y = torch.empty(B, hw, hw, cout, dtype=torch.half, device=dev)
    st = torch.zeros(4 * B * (hw // 8) * (hw // 32), 2, cout, device=dev)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {k: [] for k in libs}
    def f(lib):
        rc = lib.ocr_conv2d_f16(ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(None), L.ptr(y), L.ptr(st), sp)
        assert rc == 0
    for k, lib in libs.items():
        for _ in range(10): f(lib)
    torch.cuda.synchronize()
    for r in range(5):
        for k, lib in libs.items():
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f(lib)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * B * hw * hw * cout * cin * 9
    ideal = (cin // 64) * 18 * 1024 * min(cout, 256) // 256
    for k, lib in libs.items():
        buf = (ctypes.c_ulonglong * (2 * 1024))()
        lib.ocr_diag_read_conv(buf, ctypes.c_int(1024))
        a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 2).astype(np.float64)
        ok = a[:, 1] > 0
        ms = float(np.median(res[k]))
        print("%d:%d>%d abl=%-3s %.3f ms %5.0f TF | main loop %7.0f cyc (ideal %d: %.0f%%) clock %.2f GHz" % (
            hw, cin, cout, k, ms, fl / ms / 1e9, np.median(a[ok, 0]), ideal, 100 * ideal / np.median(a[ok, 0]),
            np.median(a[ok, 0] / a[ok, 1] * 0.1)), flush=True)
for hw, cin, cout in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(64, 512, 512), (128, 256, 256)]:
    case(hw, cin, cout)
