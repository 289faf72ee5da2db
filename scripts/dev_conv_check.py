"""Dev-only: check ocr_conv2d_f16 against torch's GPU conv on the same f16-rounded data and time it."""
import ctypes, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L

def run(n, h, w, cin, cout, k=3, dil=1, check=True, iters=5):
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    x = (torch.randn(n, h, w, cin, generator=g) ).half().to(dev)
    wt = (torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k*k*cin))**0.5).half().to(dev)  # HWIO
    w_kc = wt.permute(0, 1, 3, 2).contiguous().view(k*k, cout, cin)
    pad = dil * (k - 1) // 2
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, dil, pad, pad, 0, L.CONV_STATS)
    y = torch.empty(n, h, w, cout, dtype=torch.half, device=dev)
    mt = L.call_int('ocr_conv2d_num_mtiles', ctypes.byref(d))
    stats = torch.zeros(mt, 2, cout, dtype=torch.float32, device=dev)
    L.call('ocr_conv2d_f16', ctypes.byref(d), L.ptr(x), L.ptr(w_kc), L.ptr(None), L.ptr(y), L.ptr(stats), L.stream_ptr())
    torch.cuda.synchronize()
    if check:
        ref = F.conv2d(x.float().permute(0,3,1,2), wt.float().permute(3,2,0,1), padding=pad, dilation=dil).permute(0,2,3,1)
        err = (y.float() - ref).abs().max().item()
        s_ref = y.float().sum(dim=(0,1,2)); q_ref = (y.float()**2).sum(dim=(0,1,2))
        s = stats[:,0].sum(0); q = stats[:,1].sum(0)
        print(f'  max|y-ref|={err:.3e} (ref max {ref.abs().max().item():.2f}) stats err {((s-s_ref).abs().max()/s_ref.abs().max()).item():.2e} {((q-q_ref).abs().max()/q_ref.abs().max()).item():.2e}')
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.call('ocr_conv2d_f16', ctypes.byref(d), L.ptr(x), L.ptr(w_kc), L.ptr(None), L.ptr(y), L.ptr(stats), L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * n * h * w * cout * cin * k * k
    print(f'conv n{n} {h}x{w} {cin}->{cout} k{k} d{dil}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s')

if __name__ == '__main__':
    print(torch.cuda.get_device_name(0))
    run(2, 40, 72, 64, 64)          # ragged tile edges
    run(1, 32, 32, 512, 1024, dil=6)
    run(2, 32, 32, 1024, 1024, k=1)
    run(2, 64, 64, 32, 128)
    B = 32
    run(B, 512, 512, 64, 64, check=False)
    run(B, 256, 256, 64, 128, check=False)
    run(B, 256, 256, 128, 128, check=False)
    run(B, 128, 128, 128, 256, check=False)
    run(B, 128, 128, 256, 256, check=False)
    run(B, 64, 64, 256, 512, check=False)
    run(B, 64, 64, 512, 512, check=False)
    run(B, 32, 32, 512, 512, check=False)
    run(B, 32, 32, 512, 1024, dil=6, check=False)
    run(B, 32, 32, 1024, 1024, k=1, check=False)
