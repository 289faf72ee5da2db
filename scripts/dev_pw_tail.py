"""Dev: the 1x1 input-gradient convolutions of ResNet-50 (cin -> 4*cin expansions) in their three epilogue forms:
plain, fused BN-backward reduction (bnred), bottleneck tail (ACCUM + mask + sums), against their HBM streams."""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
B = 64
SH = [(160, 64, 256), (80, 128, 512), (40, 256, 1024), (20, 512, 2048), (160, 256, 64), (80, 512, 128)]
def run(hw, cin, cout, mode, iters=10):
    dev = 'cuda'
    x = torch.randn(B, hw, hw, cin, device=dev).half(); w = (torch.randn(1, cout, cin, device=dev) * 0.05).half()
    flags = L.CONV_ACCUM_F16 if mode == 'tail' else 0
    d = L.ConvDesc(B, hw, hw, cin, hw, hw, cout, 1, 1, 1, 1, 0, 0, 1, flags)
    y = torch.zeros(B, hw, hw, cout, dtype=torch.half, device=dev)
    by = torch.randn(B, hw, hw, cout, device=dev).half(); out = torch.randn(B, hw, hw, cout, device=dev).half()
    v = [torch.rand(cout, device=dev) + 0.5 for _ in range(4)]
    mt = L.call_int('ocr_conv2d_num_mtiles', ctypes.byref(d)); st = torch.zeros(mt, 2, cout, device=dev)
    if mode == 'tail':
        f = lambda: L.call('ocr_conv2d_bnred_tail_f16', ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), L.ptr(by), L.ptr(v[2]), L.ptr(v[3]), L.ptr(out), L.ptr(None), L.stream_ptr())
        nb = B * hw * hw * (cin + 4 * cout) * 2
    elif mode == 'bnred':
        f = lambda: L.call('ocr_conv2d_bnred_f16', ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(st), L.ptr(by), L.ptr(v[0]), L.ptr(v[1]), L.ptr(v[2]), L.ptr(v[3]), ctypes.c_int(1), L.stream_ptr())
        nb = B * hw * hw * (cin + 2 * cout) * 2
    else:
        f = lambda: L.call('ocr_conv2d_f16', ctypes.byref(d), L.ptr(x), L.ptr(w), L.ptr(None), L.ptr(y), L.ptr(None), L.stream_ptr())
        nb = B * hw * hw * (cin + cout) * 2
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best, nb
for hw, cin, cout in SH:
    r = {m: run(hw, cin, cout, m) for m in ('plain', 'bnred', 'tail')}
    print('%3d: %4d>%4d ' % (hw, cin, cout) + ' | '.join('%s %.3f ms %4.0f GB/s' % (m, r[m][0], r[m][1] / r[m][0] / 1e6) for m in r), flush=True)
