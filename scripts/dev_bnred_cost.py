"""Dev: cost of the fused BN-backward reduction in the input-gradient kernels (bnred vs plain STATS vs no stats)."""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from tensorflow_ocr_amd import _lib as L
B = 32
SH = [(512,64,64),(256,128,128),(256,128,64),(128,256,256),(64,512,512),(32,512,512)]
def run(hw,cin,cout,mode,iters=10):
    dev='cuda'
    x=torch.randn(B,hw,hw,cin,device=dev).half(); w=(torch.randn(9,cout,cin,device=dev)*0.05).half()
    flags = L.CONV_STATS if mode == 'stats' else 0
    d=L.ConvDesc(B,hw,hw,cin,hw,hw,cout,3,3,1,1,1,1,1,flags)
    y=torch.empty(B,hw,hw,cout,dtype=torch.half,device=dev)
    by=torch.randn(B,hw,hw,cout,device=dev).half()
    v=[torch.rand(cout,device=dev)+0.5 for _ in range(4)]
    mt=L.call_int('ocr_conv2d_num_mtiles',ctypes.byref(d)); st=torch.zeros(mt,2,cout,device=dev)
    if mode == 'bnred':
        f=lambda: L.call('ocr_conv2d_bnred_f16',ctypes.byref(d),L.ptr(x),L.ptr(w),L.ptr(y),L.ptr(st),L.ptr(by),L.ptr(v[0]),L.ptr(v[1]),L.ptr(v[2]),L.ptr(v[3]),ctypes.c_int(1),L.stream_ptr())
    else:
        f=lambda: L.call('ocr_conv2d_f16',ctypes.byref(d),L.ptr(x),L.ptr(w),L.ptr(None),L.ptr(y),L.ptr(st if flags else None),L.stream_ptr())
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
for hw,cin,cout in SH:
    r={m:run(hw,cin,cout,m) for m in ('none','stats','bnred')}
    print('%3d: %3d>%3d %-28s none %.3f  stats %.3f  bnred %.3f ms'%(hw,cin,cout,r['none'][1],r['none'][0],r['stats'][0],r['bnred'][0]),flush=True)
