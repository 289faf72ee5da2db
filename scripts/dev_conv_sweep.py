"""Dev-only: time the VGG conv layer shapes (B=32) for the currently selected lib/env."""
import ctypes, sys, os
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
SH = [(512,64,64,3,1),(256,64,128,3,1),(256,128,128,3,1),(128,128,256,3,1),(128,256,256,3,1),(64,256,512,3,1),(64,512,512,3,1),(32,512,512,3,1),(32,512,1024,3,6),(32,1024,1024,1,1)]
W = [1,1,1,1,2,1,2,3,1,1]
def run(hw,cin,cout,k,dil,iters=5,B=32):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); w=(torch.randn(k*k,cout,cin,device=dev)*0.05).half()
    if os.environ.get('ZERO'): x.zero_(); w.zero_()
    pad=dil*(k-1)//2
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,k,k,1,dil,pad,pad,0,L.CONV_STATS)
    y=torch.empty(B,hw,hw,cout,dtype=torch.half,device=dev)
    mt=L.call_int('ocr_conv2d_num_mtiles',ctypes.byref(d)); st=torch.zeros(mt,2,cout,device=dev)
    f=lambda: L.call('ocr_conv2d_f16',ctypes.byref(d),L.ptr(x),L.ptr(w),L.ptr(None),L.ptr(y),L.ptr(st),L.stream_ptr())
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/iters
    return ms, 2.0*B*hw*hw*cout*cin*k*k/ms/1e9
tot=0
out=[]
for (hw,cin,cout,k,dil),wt in zip(SH,W):
    ms,tf=run(hw,cin,cout,k,dil); tot+=ms*wt; out.append('%d:%d>%d %.2fms %.0fTF'%(hw,cin,cout,ms,tf))
print(os.environ.get('TAG',''), 'fwd-total %.2f ms | '%tot + ' | '.join(out))
