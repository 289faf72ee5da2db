"""Dev: A/B of the 4-wave 3x3 kernel against the 8-wave one on the VGG 256-cout layers (interleaved rounds,
one process; OCR_CONV_W4 is read once per process, so the two arms are two libraries... simplest: two
child processes would break rule 24 — instead this script times the env-selected arm only and is run twice
in ONE gpurun call, back to back on the same device)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
SH = [(128,128,256),(128,256,256),(64,256,512),(64,512,512),(32,512,512)]
def run(hw,cin,cout,iters=20,B=32,flip=0,stats=True):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); w=(torch.randn(9,cout,cin,device=dev)*0.05).half()
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,3,3,1,1,1,1,flip,L.CONV_STATS if stats else 0)
    y=torch.empty(B,hw,hw,cout,dtype=torch.half,device=dev)
    mt=L.call_int('ocr_conv2d_num_mtiles',ctypes.byref(d)); st=torch.zeros(mt,2,cout,device=dev)
    f=lambda: L.call('ocr_conv2d_f16',ctypes.byref(d),L.ptr(x),L.ptr(w),L.ptr(None),L.ptr(y),L.ptr(st),L.stream_ptr())
    for _ in range(5): f()
    torch.cuda.synchronize()
    best=1e9
    for r in range(3):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/iters)
    return best, 2.0*B*hw*hw*cout*cin*9/best/1e9
name=ctypes.create_string_buffer(128)
out=[]
for hw,cin,cout in SH:
    ms,tf=run(hw,cin,cout)
    d=L.ConvDesc(32,hw,hw,cin,hw,hw,cout,3,3,1,1,1,1,0,0); L.load().ocr_conv2d_variant(ctypes.byref(d),name,ctypes.c_size_t(128))
    out.append('%d:%d>%d %.3fms %.0fTF'%(hw,cin,cout,ms,tf))
print('W4=%s %s | '%(os.environ.get('OCR_CONV_W4','1'), name.value.decode()) + ' | '.join(out), flush=True)
