"""CPU-only (round 6, verdict task 3-ii): could the 256 -> 256 layers of the 6-D net (block4: 121 k rows, 5.4 M pairs per
6-pair batch, rule-major tiles of 64 pairs) reduce product rows INSIDE the kernel?  Measured on the oracle's stride-8
kernel map of one BASELINE configs[1] pair (teacher-forced matches as in bench.py):
  * how the pairs spread over the 729 offsets (a skewed map would allow offset-major tiles over row BLOCKS);
  * for output-row blocks of R consecutive rows (bucket order = the library's numbering of the coarse maps): pairs per
    (block, offset) = the fill of a 64-pair tile that stays inside one row block, and the share of a tile's output rows that
    the NEXT offset's tile of the same block touches again (what an LDS stage could add up before the row leaves).
    python tools/recurrence_stats.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepglobalregistration_amd import synth
from oracle import pipeline as opipe, me_semantics as me

VOX = 0.05
a, b, Tg = synth.synth_pair(0, n_raw=50000)
p0, c0, _ = opipe.preprocess(a, VOX)
p1, c1, _ = opipe.preprocess(b, VOX)
gt = synth.gt_correspondences(p0, p1, Tg, VOX, seed=0)
rng = np.random.default_rng(0)
idx1 = np.where(gt >= 0, gt, rng.integers(0, len(p1), len(p0)))     # 20 % GT matches, the rest as good as random (untrained net)
c6, _ = opipe.inlier_inputs(p0, p1, c0, c1, np.arange(len(p0)), idx1)
for ts in (2, 4, 8):
    c6 = me.stride_coords(c6 if ts == 2 else c6, ts)
# rows in bucket order: sorted by the first half (x0, y0, z0), like the library numbers its coarse 6-D maps
order = np.lexsort((c6[:, 6], c6[:, 5], c6[:, 4], c6[:, 3], c6[:, 2], c6[:, 1]))
c6 = c6[order]
k, i, o = me.kernel_map(c6, c6, 6, 3, 8)
N, P = len(c6), len(k)
cnt = np.bincount(k, minlength=729)
print(f'stride-8 map of one pair: {N} rows, {P} pairs = {P / N:.1f} per row; offsets used {np.count_nonzero(cnt)} / 729')
print(f'pairs per offset: centre {cnt[364]}, others min {np.delete(cnt, 364).min()} median {int(np.median(np.delete(cnt, 364)))} '
      f'max {np.delete(cnt, 364).max()}  (uniform would be {(P - N) / 728:.0f}); density of an offset = {np.median(cnt) / N:.3f} of the rows')
same_first = (k // 27 == 13) if False else None
for R in (64, 256, 1024, 4096):
    blk = o // R
    nb = (N + R - 1) // R
    fill = np.zeros((nb, 729), np.int32)
    np.add.at(fill, (blk, k), 1)
    nz = fill[fill > 0]
    # consecutive offsets of one block: output rows of offset kk that offset kk + 1 touches again
    rec_num = rec_den = 0
    for bsel in rng.choice(nb, min(nb, 12), replace=False):
        m = blk == bsel
        kb, ob = k[m], o[m]
        rows_of = [set(ob[kb == kk].tolist()) for kk in range(729)]
        for kk in range(728):
            if rows_of[kk] and rows_of[kk + 1]:
                rec_num += len(rows_of[kk] & rows_of[kk + 1]); rec_den += len(rows_of[kk + 1])
    print(f'R = {R:5d} rows per block ({R} KB of f32 accumulators at 256 channels): pairs per (block, offset) mean {fill.mean():6.1f} '
          f'(non-empty {nz.mean():6.1f}, share of non-empty cells {np.count_nonzero(fill) / fill.size:.2f}) -> fill of a 64-pair tile '
          f'{min(1.0, nz.mean() / 64):.2f}; rows of a tile that the next offset\'s tile of the block touches again: {rec_num / max(1, rec_den):.3f}')
print('(a 160-KB LDS holds the accumulators of R <= 128 rows: tiles 6 % full; an LDS stage that adds recurring rows of CONSECUTIVE tiles '
      'saves the recurrence share above of the product rows)')
