"""CPU-only: candidates per query of the kNN prefilter when its first pass samples every SUB-th stage of the interleaved
reference tiles (knn.hip), on benchmark-shaped features from the oracle net, in f64.  `python tools/knn_sample_sim.py`"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from deepglobalregistration_amd import synth
from oracle import pipeline as opipe, resunet as oresunet
torch.set_num_threads(8)
voxel=0.05
ck = synth.synth_checkpoint(seed=0, voxel_size=voxel, feat_conv1_kernel_size=7)
x0,x1,T=synth.synth_pair(0,n_raw=50000)
t=time.time()
p0,c0,f0=opipe.preprocess(x0,voxel); p1,c1,f1=opipe.preprocess(x1,voxel)
F0=oresunet.resunet_forward(ck['state_dict'],c0,f0,3,7,True); F1=oresunet.resunet_forward(ck['state_dict'],c1,f1,3,7,True)
print('features', F0.shape, F1.shape, time.time()-t)

F0=torch.from_numpy(F0).double(); F1=torch.from_numpy(F1).double()
N0,N1=len(F0),len(F1)
nrt=(N1+31)//32
r=np.arange(N1)
tile=r % nrt            # interleave: row r sits in tile r % nrt
stage=tile//4
for SUB in (1,2,4):
    sampled = torch.from_numpy((stage % SUB)==0)
    na=(F0*F0).sum(1); nb=(F1*F1).sum(1); nmax=nb.max()
    cnts=[]
    for s in range(0,N0,2048):
        a=F0[s:s+2048]
        d=(na[s:s+2048,None]+nb[None,:]-2*a@F1.T)
        dm=d[:,sampled].min(1).values
        thr=dm+8e-5*(na[s:s+2048]+nmax)
        cnts.append((d<=thr[:,None]).sum(1))
    c=torch.cat(cnts).numpy()
    print('SUB',SUB,'mean cand',c.mean(),'p50',np.percentile(c,50),'p90',np.percentile(c,90),'p99',np.percentile(c,99),'max',c.max(),'frac>8',(c>8).mean(),'frac>16',(c>16).mean(),'frac>32',(c>32).mean())
# how concentrated: gap between nn and 2nd nn
d=(na[:2048,None]+nb[None,:]-2*F0[:2048]@F1.T)
srt=torch.sort(d,1).values
print('d1 median',srt[:,0].median().item(),'d2-d1 median',(srt[:,1]-srt[:,0]).median().item(),'d16-d1 median',(srt[:,15]-srt[:,0]).median().item(), 'd100-d1', (srt[:,99]-srt[:,0]).median().item())
