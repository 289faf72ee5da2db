"""Dev: time the 3x3 weight-gradient kernel on the VGG layer shapes for the env-selected arm (OCR_WGRAD3=0/1)."""
import ctypes, os, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
SH = [(512,64,64),(256,64,128),(256,128,128),(128,128,256),(128,256,256),(64,256,512),(64,512,512),(32,512,512)]
WT = [1,1,1,1,2,1,2,3]
def run(hw,cin,cout,iters=10,B=32):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); dy=(torch.randn(B,hw,hw,cout,device=dev)*0.1).half()
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,3,3,1,1,1,1,0,0)
    nbytes=L.call_size("ocr_conv2d_wgrad_workspace", ctypes.byref(d))
    ws=torch.empty(nbytes,dtype=torch.uint8,device=dev); dw=torch.empty(3,3,cin,cout,device=dev)
    f=lambda: L.call("ocr_conv2d_wgrad_f16", ctypes.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), ctypes.c_size_t(nbytes), L.stream_ptr())
    for _ in range(3): f()
    torch.cuda.synchronize()
    best=1e9
    for r in range(3):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/iters)
    return best, 2.0*B*hw*hw*cout*cin*9/best/1e9, dw
tot=0; out=[]
for (hw,cin,cout),wt in zip(SH,WT):
    ms,tf,_=run(hw,cin,cout); tot+=ms*wt; out.append('%d:%d>%d %.3fms %.0fTF'%(hw,cin,cout,ms,tf))
print('WGRAD3=%s step-total %.3f ms | '%(os.environ.get('OCR_WGRAD3','1'),tot)+' | '.join(out), flush=True)
