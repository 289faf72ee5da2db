"""Dev: the dilated fc6 convolution (3x3, rate 6, 512 -> 1024 at 32x32, batch 32): forward and input gradient."""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
B, hw = 32, 32
for cin, cout, flip in ((512, 1024, 0), (1024, 512, 1)):
    x = torch.randn(B, hw, hw, cin, device='cuda').half(); w = (torch.randn(9, cout, cin, device='cuda') * 0.05).half()
    d = L.ConvDesc(B, hw, hw, cin, hw, hw, cout, 3, 3, 1, 6, 6, 6, flip, L.CONV_STATS)
    y = torch.empty(B, hw, hw, cout, dtype=torch.half, device='cuda')
    mt = L.call_int('ocr_conv2d_num_mtiles', ctypes.byref(d)); st = torch.zeros(mt, 2, cout, device='cuda')
    f = lambda: L.call('ocr_conv2d_f16', ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(None), L.ptr(y), L.ptr(st), L.stream_ptr())
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    name = ctypes.create_string_buffer(128); L.load().ocr_conv2d_variant(ctypes.byref(d), name, ctypes.c_size_t(128))
    print('%d>%d flip%d %s %.3f ms %.0f TF' % (cin, cout, flip, name.value.decode(), best, 2.0 * B * hw * hw * cin * cout * 9 / best / 1e9), flush=True)
