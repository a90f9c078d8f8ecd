#!/bin/bash
# round 6, run 25: bench lines after the FCGF kernel work (default 3 x 6, one stream x 6, one pair)
R=$PWD; O=$R/gpurun_out/run25; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), d['stage_ms_per_batch'])
print('   ', {k: round(r[k], 4) if isinstance(r[k], float) else r[k] for k in ('frac','frac_one_stream','c_le_64_hbm_frac','c_le_64_frac_own_pipe','c_le_64_ms_per_batch')}, d['config'].get('parity_ok'))
P
}
timeout 600 $B --no-parity --steps 60 > $O/b_default.json 2> $O/b_default.err; show $O/b_default.json
timeout 300 $B --no-parity --streams 1 --steps 40 > $O/b_s1.json 2> $O/b_s1.err; show $O/b_s1.json
timeout 300 $B --no-parity --streams 1 --pairs-per-step 1 --steps 80 > $O/b_s1b1.json 2> $O/b_s1b1.err; show $O/b_s1b1.json
