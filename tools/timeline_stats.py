#!/usr/bin/env python
"""What a multi-stream run does with the GPU's time (from tools/timeline_dump.py's CSV): how much of a steady-state
window has 0 / 1 / 2 / 3+ kernels in flight, per-stream busy time, and per kernel name the mean duration next to a
one-stream run of the same workload.

    python tools/timeline_stats.py timeline_s3_b6.csv.gz [timeline_s1_b6.csv.gz]
"""
import csv
import gzip
import sys
from collections import defaultdict

import numpy as np


def load(path):
    with gzip.open(path, 'rt', newline='') as f:
        r = csv.reader(f)
        names = next(r)[1:]
        cols = next(r)
        rows = [row for row in r]
    ix = {c: i for i, c in enumerate(cols)}
    nm = np.array([int(x[0]) for x in rows])
    st = np.array([int(x[ix['start']]) for x in rows], np.int64)
    en = np.array([int(x[ix['end']]) for x in rows], np.int64)
    q = np.array([int(x[ix['stream_id']]) for x in rows])
    gx = np.array([int(x[ix['grid_x']]) for x in rows])
    return names, nm, st, en, q, gx


def window(st, en, frac=(0.45, 0.95)):
    """The steady-state part of the run: the given fraction of the span between the first and the last launch (the timed
    steps are the tail of the process; the head is start-up and warm-up)."""
    t0, t1 = st.min(), en.max()
    return t0 + int((t1 - t0) * frac[0]), t0 + int((t1 - t0) * frac[1])


def short(n):
    n = n.replace('void ', '')
    return n[:n.index('(')] if '(' in n else n


def main():
    names, nm, st, en, q, gx = load(sys.argv[1])
    # the timed region: the densest part -- take the last 8 steps' worth by looking at the big kernel's launches
    big = [i for i, n in enumerate(names) if 'sparse_conv_wide_f16x2<256, 2, 1>' in n]
    sel = np.isin(nm, big)
    bs = np.sort(st[sel])
    # every batch has 2 launches of it; the steady state = the last 2/3 of them
    lo, hi = bs[len(bs) // 3], bs[-1]
    cnt = {s_: int((sel & (q == s_)).sum()) for s_ in sorted(set(q.tolist()))}
    streams = [s_ for s_, c in cnt.items() if c > 4 and 2 * c >= max(cnt.values())]   # (not the stream of a short extra leg)
    if len(streams) > 1:
        # the multi-stream region: where EVERY stream that runs the big kernel is active; its first 30 % is warm-up
        # (each stream also has earlier launches of its own: start-up, its single-stream warm-up) -- the longest run of
        # 100-ms bins in which EVERY one of these streams launches the big kernel
        t0 = st.min()
        nb = int((st.max() - t0) // 100_000_000) + 1
        every = np.ones(nb, bool)
        for s_ in streams:
            h = np.zeros(nb, bool)
            h[((st[sel & (q == s_)] - t0) // 100_000_000).astype(int)] = True
            every &= h
        best, cur, best_end = 0, 0, 0
        for i, v in enumerate(every):
            cur = cur + 1 if v else 0
            if cur > best:
                best, best_end = cur, i
        a = t0 + (best_end - best + 1) * 100_000_000
        b = t0 + (best_end + 1) * 100_000_000
        lo, hi = a + int(0.3 * (b - a)), b
    m = (st >= lo) & (en <= hi)
    print(f'{sys.argv[1]}: window {1e-6 * (hi - lo):.1f} ms, {m.sum()} launches, streams {sorted(set(q[m].tolist()))}')
    ev = np.concatenate([np.stack([st[m], np.ones(m.sum(), np.int64)], 1), np.stack([en[m], -np.ones(m.sum(), np.int64)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    depth, t_prev, hist = 0, lo, defaultdict(int)
    for t, d in ev:
        hist[depth] += t - t_prev
        t_prev, depth = t, depth + d
    tot = sum(hist.values())
    print('  kernels in flight -> share of the window:', {k: round(v / tot, 3) for k, v in sorted(hist.items())})
    for s in sorted(set(q[m].tolist())):
        ms = m & (q == s)
        print(f'  stream {s}: {ms.sum()} launches, sum of durations {1e-6 * (en[ms] - st[ms]).sum():.1f} ms '
              f'({(en[ms] - st[ms]).sum() / (hi - lo):.2f} of the window)')
    n_big = int(((st[sel] >= lo) & (en[sel] <= hi)).sum())
    batches = n_big / 2
    print(f'  {batches:.1f} batches in the window: {1e-6 * (hi - lo) / batches:.2f} ms per batch; kernel time per batch '
          f'{1e-6 * (en[m] - st[m]).sum() / batches:.2f} ms')
    per = defaultdict(list)
    for i in np.nonzero(m)[0]:
        per[nm[i]].append(en[i] - st[i])
    ref = {}
    if len(sys.argv) > 2:
        n1, nm1, st1, en1, q1, gx1 = load(sys.argv[2])
        big1 = [i for i, n in enumerate(n1) if 'sparse_conv_wide_f16x2<256, 2, 1>' in n]
        b1 = np.sort(st1[np.isin(nm1, big1)])
        lo1, hi1 = b1[len(b1) // 3], b1[-1]
        m1 = (st1 >= lo1) & (en1 <= hi1)
        nb1 = int((np.isin(nm1, big1) & m1).sum()) / 2
        d1 = defaultdict(list)
        for i in np.nonzero(m1)[0]:
            d1[n1[nm1[i]]].append(en1[i] - st1[i])
        ref = {k: (np.mean(v), len(v) / nb1) for k, v in d1.items()}
        ev1 = (en1[m1] - st1[m1]).sum()
        print(f'  one stream: {1e-6 * (hi1 - lo1) / nb1:.2f} ms per batch, kernel time per batch {1e-6 * ev1 / nb1:.2f} ms')
    print(f'  {"kernel":58s} {"n/batch":>7s} {"ms/batch":>8s} {"mean us":>8s} {"alone us":>8s} {"stretch":>7s}')
    rows = sorted(per.items(), key=lambda kv: -sum(kv[1]))
    for k, v in rows[:32]:
        a = ref.get(names[k])
        print(f'  {short(names[k])[:58]:58s} {len(v) / batches:7.1f} {1e-6 * sum(v) / batches:8.3f} {1e-3 * np.mean(v):8.1f} '
              + (f'{1e-3 * a[0]:8.1f} {np.mean(v) / a[0]:7.2f}' if a else ''))


if __name__ == '__main__':
    main()
