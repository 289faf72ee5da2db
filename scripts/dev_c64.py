"""Dev-only: time the 64-channel 3x3 layers (conv1_2 class) with / without stats."""
import ctypes, sys, os
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
def run(hw,cin,cout,flags,iters=10,B=32):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); w=(torch.randn(9,cout,cin,device=dev)*0.05).half()
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,3,3,1,1,1,1,0,flags)
    y=torch.empty(B,hw,hw,cout,dtype=torch.half,device=dev)
    mt=L.call_int('ocr_conv2d_num_mtiles',ctypes.byref(d)); st=torch.zeros(mt,2,cout,device=dev)
    f=lambda: L.call('ocr_conv2d_f16',ctypes.byref(d),L.ptr(x),L.ptr(w),L.ptr(None),L.ptr(y),L.ptr(st if flags else None),L.stream_ptr())
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/iters
    return '%d:%d>%d f%d %.3fms %.0fTF'%(hw,cin,cout,flags,ms,2.0*B*hw*hw*cout*cin*9/ms/1e9)
print(os.environ.get('TAG',''), ' | '.join([run(512,64,64,4),run(512,64,64,0),run(256,64,128,4)]))
