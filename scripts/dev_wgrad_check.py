"""Dev-only: check ocr_conv2d_wgrad_f16 against torch GPU autograd and time it."""
import ctypes, sys
import torch, torch.nn.functional as F
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L

def run(n, h, w, cin, cout, k=3, dil=1, check=True, iters=3, stride=1):
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn(n, h, w, cin, generator=g).half().to(dev)
    oh, ow = (h + stride - 1) // stride, (w + stride - 1) // stride
    dy = (torch.randn(n, oh, ow, cout, generator=g) * 0.1).half().to(dev)
    pad = dil * (k - 1) // 2
    d = L.ConvDesc(n, h, w, cin, oh, ow, cout, k, k, stride, dil, pad, pad, 0, 0)
    ws_bytes = L.call_size('ocr_conv2d_wgrad_workspace', ctypes.byref(d))
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    dw = torch.zeros(k, k, cin, cout, dtype=torch.float32, device=dev)
    def go():
        L.call('ocr_conv2d_wgrad_f16', ctypes.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), ctypes.c_size_t(ws_bytes), L.stream_ptr())
    go(); torch.cuda.synchronize()
    if check:
        wt = torch.zeros(cout, cin, k, k, device=dev, requires_grad=True)
        out = F.conv2d(x.float().permute(0,3,1,2), wt, padding=pad, dilation=dil, stride=stride)
        out.backward(dy.float().permute(0,3,1,2))
        ref = wt.grad.permute(2,3,1,0)
        err = (dw - ref).abs().max().item()
        print(f'  max|dw-ref|={err:.3e} (ref max {ref.abs().max().item():.2f})')
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * n * oh * ow * cout * cin * k * k
    print(f'wgrad n{n} {h}x{w} {cin}->{cout} k{k} d{dil} s{stride}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s  ws {ws_bytes/1e6:.0f} MB')

if __name__ == '__main__':
    run(2, 40, 72, 64, 64)
    run(1, 32, 32, 128, 64, dil=6)
    run(2, 32, 32, 128, 128, k=1)
    for ci, co in [(64,64),(64,128),(64,256),(128,64),(128,256),(256,64),(256,128),(256,256),(512,1024)]:
        run(2, 33, 45, ci, co, k=1)
    run(3, 31, 47, 256, 512, k=1, stride=2)
    if 'pw' in sys.argv:
        for ci, co, hw in [(1024,1024,32),(64,256,160),(256,64,160),(256,128,160),(128,512,80),(512,128,80),(512,256,80),(256,1024,40),(1024,256,40),(1024,512,40),(512,2048,20),(2048,512,20)]:
            run(64 if hw != 32 else 32, hw, hw, ci, co, k=1, check=False)
        sys.exit()
    B = 32
    run(B, 512, 512, 64, 64, check=False)
    run(B, 256, 256, 64, 128, check=False)
    run(B, 256, 256, 128, 128, check=False)
    run(B, 128, 128, 256, 256, check=False)
    run(B, 64, 64, 512, 512, check=False)
    run(B, 32, 32, 512, 512, check=False)
    run(B, 32, 32, 512, 1024, dil=6, check=False)
    run(B, 32, 32, 1024, 1024, k=1, check=False)
