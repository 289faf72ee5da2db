import sys, numpy as np, torch
sys.path.insert(0,'.')
from oracle import ocr_oracle as O
from tensorflow_ocr_amd import ops
dev='cuda'
def run(n,h,w,cin,cout,flags=0):
    rng=np.random.default_rng(1)
    x=torch.from_numpy(rng.standard_normal((n,h,w,cin)).astype(np.float32)).half().float().numpy()
    wt=torch.from_numpy((rng.standard_normal((3,3,cin,cout))*np.sqrt(2.0/(9*cin))).astype(np.float32)).half().float().numpy()
    yo=O.conv2d(torch.from_numpy(x),torch.from_numpy(wt),1,1).numpy()
    xd=torch.from_numpy(x).half().to(dev); wm=torch.from_numpy(wt).to(dev)
    w_kc=torch.empty((9,cout,cin),dtype=torch.half,device=dev); w_ck=torch.empty((9,cin,cout),dtype=torch.half,device=dev)
    ops.pack_weights(wm,w_kc,w_ck)
    d=ops.conv_desc((n,h,w,cin),cout,3,3,1,1); d.flags=flags
    errs=[]
    for it in range(6):
        y=torch.full((n,h,w,cout),float('nan'),dtype=torch.half,device=dev)
        part=torch.zeros((ops.conv2d_num_mtiles(d),2,cout),device=dev)
        ops.conv2d(d,xd,w_kc,y,None,part if flags else None)
        torch.cuda.synchronize()
        yy=y.float().cpu().numpy()
        e=np.abs(yy-yo); errs.append(float(np.nanmax(e)/np.abs(yo).max()))
        if errs[-1]>1e-2 and it==0:
            bad=np.argwhere(~(e<1e-2*np.abs(yo).max()))
            print("  bad count",len(bad),"first",bad[:5].tolist(),"last",bad[-3:].tolist(), "nan count", int(np.isnan(yy).sum()))
    print((n,h,w,cin,cout,flags), ops.conv2d_variant(d), ["%.1e"%e for e in errs])
for sh in [(1,20,32,128,256),(1,24,32,128,256),(1,16,32,128,256),(1,20,32,64,256),(2,33,70,128,256),(1,20,64,128,256)]:
    run(*sh)
    run(*sh,flags=4)
