#!/bin/bash
# round 6, run 42: same-box A/B, three times each: the shipped default (3 unmasked streams x 6 pairs) against four streams
# on their own quarter of the compute units (64 CUs each; scratch build of the run37/38 code)
R=$PWD; O=$R/gpurun_out/run42; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
F="--no-parity --no-cpu-baseline --no-exact-leg --steps 40"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'dominant us %.0f' % r['avg_launch_us'])
P
}
for i in 1 2 3; do
  timeout 300 python $R/bench.py $F > $O/base_$i.json 2> $O/base_$i.err; show $O/base_$i.json
  (cd $R/scratch_ab/w77 && DGR_BENCH_CU_SPLIT=se DGR_WIDE_CUS=64 timeout 300 python bench.py $F --streams 4 > $O/m4se_$i.json 2> $O/m4se_$i.err); show $O/m4se_$i.json
  (cd $R/scratch_ab/w77 && DGR_BENCH_CU_SPLIT=slots DGR_WIDE_CUS=64 timeout 300 python bench.py $F --streams 4 > $O/m4sl_$i.json 2> $O/m4sl_$i.err); show $O/m4sl_$i.json
done
(cd $R/scratch_ab/w77 && timeout 300 python bench.py $F --streams 4 > $O/u4.json 2> $O/u4.err); show $O/u4.json
