"""Dev-only: run one conv shape a few times (for rocprofv3 --pmc)."""
import ctypes, sys, os
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
hw,cin,cout,k,dil = [int(a) for a in sys.argv[1:6]]
B=32; dev='cuda'
x=torch.randn(B,hw,hw,cin,device=dev).half(); w=(torch.randn(k*k,cout,cin,device=dev)*0.05).half()
pad=dil*(k-1)//2
d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,k,k,1,dil,pad,pad,0,L.CONV_STATS)
y=torch.empty(B,hw,hw,cout,dtype=torch.half,device=dev)
mt=L.call_int('ocr_conv2d_num_mtiles',ctypes.byref(d)); st=torch.zeros(mt,2,cout,device=dev)
for _ in range(3):
    L.call('ocr_conv2d_f16',ctypes.byref(d),L.ptr(x),L.ptr(w),L.ptr(None),L.ptr(y),L.ptr(st),L.stream_ptr())
torch.cuda.synchronize()
