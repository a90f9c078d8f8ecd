"""CPU-only: how many DISTINCT neighbour rows do the 128 (64) output rows of a workgroup have, by row order?  The design
input of the LDS-resident halo tiles planned in DESIGN.md section 8 (item 3): first-occurrence order (today), rows
grouped by their stride-8 parent (parents in first-occurrence or Morton order), full Morton order.  Uses the oracle's
kernel maps on synthetic BASELINE-shaped clouds; `python tools/halo_stats.py`."""
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
    res={}
    for oname in ('first_occurrence','block8_first','block8_morton','morton'):
        if oname=='first_occurrence': order=np.arange(n)
        elif oname=='morton': order=np.argsort(morton3(xyz//ts),kind='stable')
        else:
            blk = xyz//8   # stride-8 parent
            _, first, inv = np.unique(blk,axis=0,return_index=True,return_inverse=True)
            if oname=='block8_first':
                rank = np.argsort(np.argsort(first))   # blocks in order of first occurrence
                order=np.lexsort((np.arange(n), rank[inv.reshape(-1)]))
            else:
                bm = morton3(blk)
                order=np.lexsort((np.arange(n), bm))
        for T in (128,64):
            sizes=[]
            for s in range(0,n,T):
                rows=order[s:s+T]
                u=np.unique(nb[:,rows]); u=u[u>=0]
                sizes.append(len(u))
            sizes=np.array(sizes)
            res[(oname,T)]=(sizes.mean(), np.percentile(sizes,99), sizes.max())
    print(name, 'n=',n, 'pairs/row', len(k)/n)
    for kk,v in res.items(): print('   ',kk, 'halo mean %.0f p99 %.0f max %d'%v)

a,b,_=synth.synth_pair(1,n_raw=50000)
p0,c0,_=opipe.preprocess(a,0.05)
stats(c0,1,'level0 (stride 1)')
c2=me.stride_coords(c0,2)
stats(c2,2,'level1 (stride 2)')
a,b,_=synth.synth_pair(1,n_raw=120000,kind='outdoor')
p0,c0,_=opipe.preprocess(a,0.3)
stats(c0,1,'outdoor level0')
