import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import ocr_oracle as O
from tensorflow_ocr_amd import checkpoint
from tensorflow_ocr_amd.graph import Graph
from tensorflow_ocr_amd.nets import model_vgg_16 as M
S=1024.0
size=int(sys.argv[1]) if len(sys.argv)>1 else 64; n=int(sys.argv[2]) if len(sys.argv)>2 else 2
rng=np.random.default_rng(0)
p=O.init_model_vgg_params(rng)
images,pixel,link,mask=O.synthetic_batch(rng,n,size)
g=Graph('cuda:0',loss_scale=S)
M.model_vgg(images,graph=g); g.reset_tape()
g.store.load_state_dict(checkpoint.tf_to_internal(g.store.order,p))
px,lk=M.model_vgg(images,graph=g)
ep_d={k:(v.data.float().cpu().numpy() if v.data is not None else None) for k,v in g.end_points.items()}
L=M.loss(pixel,px,link,lk,mask,graph=g); g.backward(); torch.cuda.synchronize()
for mixed in (True,False):
    tp=O.to_torch_params(p)
    a,b,ep=O.model_vgg(torch.from_numpy(images),tp,True,mixed=mixed)
    Lo=O.dice_loss(torch.from_numpy(pixel),a,torch.from_numpy(link),b,torch.from_numpy(mask)); (Lo*S).backward()
    print('mixed' if mixed else 'f32', 'loss', L.item(), float(Lo))
    for k in ['conv1_2','conv2_2','conv3_3','conv4_3','conv5_3','fc6','fc7']:
        if ep_d[k] is None: continue
        e=ep[k].detach().numpy(); d=ep_d[k]
        print('  %-8s Linf %.3e  mean|d| %.3e  max|ref| %.2f  frac>1e-2: %.4f'%(k,np.abs(d-e).max(),np.abs(d-e).mean(),np.abs(e).max(),(np.abs(d-e)>1e-2).mean()))
    print('  pixel Linf %.3e link Linf %.3e'%(np.abs(px.data.cpu().numpy()-a.detach().numpy()).max(),np.abs(lk.data.cpu().numpy()-b.detach().numpy()).max()))
    gr=checkpoint.internal_to_tf({nm:(v.grad/S).cpu().numpy() for nm,v in g.store.vars.items() if v.trainable})
    for k in sorted(gr):
        og=(tp[k].grad/S).numpy()
        print('  grad %-45s rel %.3e |g|max %.3e'%(k,np.abs(gr[k]-og).max()/(np.abs(og).max()+1e-20),np.abs(og).max()))
