import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tensorflow_ocr_amd import synthetic
from tensorflow_ocr_amd.graph import Graph
from tensorflow_ocr_amd.nets import model_vgg_16 as M
from tensorflow_ocr_amd.train import AdamOptimizer, TrainStep
g=Graph('cuda:0',loss_scale=1024.0,seed=1)
rng=np.random.default_rng(100)
B=int(sys.argv[1]) if len(sys.argv)>1 else 8; S=int(sys.argv[2]) if len(sys.argv)>2 else 256
batch=[torch.from_numpy(a).cuda() for a in synthetic.make_batch(rng,B,S)]
def fl(gr,im,px,lk,mk):
    a,b=M.model_vgg(im,graph=gr); return M.loss(px,a,lk,b,mk,graph=gr)
step=TrainStep(g,fl,lambda gr:AdamOptimizer(gr,learning_rate=float(sys.argv[3]) if len(sys.argv)>3 else 1e-4))
for i in range(12):
    L=step(*batch); print(i, round(L.item(),5), [round(float(t),3) for t in L.terms()[:3]])
