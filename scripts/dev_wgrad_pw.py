"""Dev: time the pointwise weight-gradient kernel on ResNet-50 / VGG shapes."""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
# (B, hw, cin, cout, k, dil)
SH = [(64,160,64,256,1,1),(64,160,256,64,1,1),(64,80,128,512,1,1),(64,80,512,128,1,1),(64,40,256,1024,1,1),(64,40,1024,256,1,1),
      (64,20,512,2048,1,1),(64,20,2048,512,1,1),(32,32,1024,1024,1,1),(32,32,512,1024,3,6)]
def run(B,hw,cin,cout,k,dil,iters=10):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); dy=(torch.randn(B,hw,hw,cout,device=dev)*0.1).half()
    pad = dil*(k-1)//2
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,k,k,1,dil,pad,pad,0,0)
    nbytes=L.call_size("ocr_conv2d_wgrad_workspace", ctypes.byref(d))
    ws=torch.empty(nbytes,dtype=torch.uint8,device=dev); dw=torch.empty(k,k,cin,cout,device=dev)
    f=lambda: L.call("ocr_conv2d_wgrad_f16",ctypes.byref(d),L.ptr(x),L.ptr(dy),L.ptr(dw),L.ptr(ws),ctypes.c_size_t(nbytes),L.stream_ptr())
    for _ in range(3): f()
    torch.cuda.synchronize()
    best=1e9
    for r in range(3):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/iters)
    return best
tot=0
for s in SH:
    ms=run(*s); tot+=ms
    B,hw,cin,cout,k,dil=s
    fl=2.0*B*hw*hw*cin*cout*k*k; gb=B*hw*hw*(cin+cout)*2/1e9
    print('%3d: %4d>%4d k%d  %.3f ms (incl. slab reduce) %5.0f TF  %4.0f GB/s'%(hw,cin,cout,k,ms,fl/ms/1e9,gb/ms*1e3),flush=True)
print('total %.3f ms'%tot)
