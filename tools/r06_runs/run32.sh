#!/bin/bash
# round 6, run 32 (= run 28 again, on a tree that is not being edited): the whole GPU suite and the driver's command at the round's last commit
R=$PWD; O=$R/gpurun_out/run32; mkdir -p $O; rm -rf $O/*
COMMIT=$(tr -d '\n' < $R/tools/COMMIT 2>/dev/null || echo unknown)
cd $R && (timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6; echo "# commit $COMMIT") | tee $O/pytest_gpu.log
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6; echo "# commit $COMMIT") > $O/smoke.log; head -1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
DGR_EVIDENCE_COMMIT=$COMMIT timeout 900 python $R/bench.py > $O/bench_c1_default.json 2> $O/bench_c1_default.err
python - <<P
import json
d=json.loads([l for l in open('$O/bench_c1_default.json') if l.startswith('{')][-1]); r=d['roofline']
print('pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), d['stage_ms_per_batch'], d['config'].get('parity_ok'), d['config'].get('parity_within_1e-4'), d.get('commit'))
P
