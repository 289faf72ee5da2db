"""Dev: time the 64-/128-cout 3x3 layers (forward shapes and the input-gradient shapes) for the env-selected arm."""
import ctypes, os, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
# (hw, cin, cout, flip)  conv1_2 fwd/dgrad, conv2_1 fwd, conv2_1 dgrad, conv2_2 fwd/dgrad, conv3_1 dgrad
SH = [(512,64,64,0),(512,64,64,1),(256,64,128,0),(256,128,64,1),(256,128,128,0),(256,128,128,1),(128,256,128,1)]
def run(hw,cin,cout,flip,iters=10,B=32):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); w=(torch.randn(9,cout,cin,device=dev)*0.05).half()
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,3,3,1,1,1,1,flip,L.CONV_STATS)
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
    return best, 2.0*B*hw*hw*cout*cin*9/best/1e9, name.value.decode()
tot=0
for hw,cin,cout,flip in SH:
    ms,tf,nm=run(hw,cin,cout,flip); tot+=ms
    print('W4S=%s %d:%d>%d flip%d %-34s %.3f ms %5.0f TF  %.0f GB/s in+out'%(os.environ.get('OCR_CONV_W4S','1'),hw,cin,cout,flip,nm,ms,tf, 32*hw*hw*(cin+cout)*2/ms/1e6), flush=True)
print('total %.3f ms'%tot)
