"""CPU-only: how many (16- or 32-row group, offset) tiles of the dense-tile 3-D kernel are EMPTY, by row order?  (Round 5:
would skipping empty tiles pay once the rows are in Morton order?  No: 99 % are non-empty.)  `python tools/tile_fill_stats.py`"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from deepglobalregistration_amd import synth
from oracle import pipeline as opipe, me_semantics as me
def morton3(c):
    c = (c - c.min(0)).astype(np.uint64); key = np.zeros(len(c), np.uint64)
    for b in range(16):
        for d in range(3):
            key |= ((c[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3*b+d)
    return key
def stats(coords, ts, name):
    n=len(coords)
    k,i,o = me.kernel_map(coords, coords, 3, 3, ts)
    nb = -np.ones((27,n),np.int64); nb[k,o]=i
    xyz=coords[:,1:]
    print(name,'n',n,'fill', (nb>=0).mean())
    for oname in ('first','morton'):
        order=np.arange(n) if oname=='first' else np.argsort(morton3(xyz//ts),kind='stable')
        for T in (16,32):
            m=(n//T)*T
            f=(nb[:,order[:m]]>=0).reshape(27,m//T,T)
            nonempty=f.any(2)   # [27, groups]
            print('   ',oname,'T',T,'non-empty (group, offset) tiles %.3f'%nonempty.mean(), ' fill inside non-empty %.3f'%(f.sum()/ (nonempty.sum()*T)))
a,b,_=synth.synth_pair(1,n_raw=50000)
p0,c0,_=opipe.preprocess(a,0.05)
stats(c0,1,'level0')
stats(me.stride_coords(c0,2),2,'level1')
