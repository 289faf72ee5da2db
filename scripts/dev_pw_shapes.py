"""Dev: time the 1x1 convolutions of ResNet-50 EAST at 640x640, batch 64 (forward and input-gradient shapes),
against the HBM stream of their operands."""
import ctypes, os, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
B = 64
# (hw, cin, cout)
SH = [(160,64,64),(160,64,256),(160,256,64),(80,256,128),(80,128,512),(80,512,128),(40,512,256),(40,256,1024),(40,1024,256),
      (20,1024,512),(20,512,2048),(20,2048,512)]
def run(hw,cin,cout,flags,iters=10):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); w=(torch.randn(1,cout,cin,device=dev)*0.05).half()
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,1,1,1,1,0,0,0,flags)
    y=torch.empty(B,hw,hw,cout,dtype=torch.half,device=dev)
    mt=L.call_int('ocr_conv2d_num_mtiles',ctypes.byref(d)); st=torch.zeros(mt,2,cout,device=dev)
    f=lambda: L.call('ocr_conv2d_f16',ctypes.byref(d),L.ptr(x),L.ptr(w),L.ptr(None),L.ptr(y),L.ptr(st),L.stream_ptr())
    for _ in range(3): f()
    torch.cuda.synchronize()
    best=1e9
    for r in range(3):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/iters)
    name=ctypes.create_string_buffer(128); L.load().ocr_conv2d_variant(ctypes.byref(d),name,ctypes.c_size_t(128))
    return best, name.value.decode()
tot=0; floor=0
for hw,cin,cout in SH:
    for flags,tag in ((L.CONV_STATS,'stats'),(0,'plain')):
        ms,nm=run(hw,cin,cout,flags)
        gb=B*hw*hw*(cin+cout)*2/1e9
        print('%3d: %4d>%4d %-5s %-24s %.3f ms  %5.0f TF  %.2f GB  %4.0f GB/s  (5 TB/s floor %.3f ms)'%(hw,cin,cout,tag,nm,ms,2.0*B*hw*hw*cin*cout/ms/1e9,gb,gb/ms*1e3,gb/5.0), flush=True)
        if tag=='stats': tot+=ms; floor+=gb/5.0
print('total (stats) %.3f ms, floor %.3f ms'%(tot,floor))
