#!/bin/bash
# round 6, run 43: the shipped CU partition (dgr_ctx_create_partition_stream; bench default 4 contexts x 6 pairs, each on
# its own quarter of the compute units) -- its test, the whole default line with every leg, and the old default next to it
R=$PWD; O=$R/gpurun_out/run43; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'frac %.3f (peak %.0f, %s CUs) one-stream %.3f; C<=64 hbm %.3f own %.3f; exact %s; parity %s' % (r['frac'], r['peak'], r.get('cus_of_a_launch'), r['frac_one_stream'], r['c_le_64_hbm_frac'], r['c_le_64_frac_own_pipe'], r.get('exact_f32_pairs_per_s'), d['config'].get('parity_ok')), d['stage_ms_per_batch'])
P
}
(cd $R && timeout 900 python -m pytest tests/test_gpu_partition.py -x -q -m gpu 2>&1 | tail -5)
timeout 900 python $R/bench.py --steps 40 > $O/b_default.json 2> $O/b_default.err || tail -20 $O/b_default.err; show $O/b_default.json
timeout 600 python $R/bench.py --steps 40 --no-cu-partition --streams 3 --no-parity --no-cpu-baseline > $O/b_s3_plain.json 2> $O/b_s3_plain.err; show $O/b_s3_plain.json
timeout 600 python $R/bench.py --steps 40 --pairs-per-step 8 --no-parity --no-cpu-baseline --no-exact-leg > $O/b_p4_b8.json 2> $O/b_p4_b8.err; show $O/b_p4_b8.json
timeout 600 python $R/bench.py --steps 40 --pairs-per-step 4 --no-parity --no-cpu-baseline --no-exact-leg > $O/b_p4_b4.json 2> $O/b_p4_b4.err; show $O/b_p4_b4.json
timeout 600 python $R/bench.py --steps 40 --streams 2 --no-parity --no-cpu-baseline --no-exact-leg > $O/b_p2_b6.json 2> $O/b_p2_b6.err; show $O/b_p2_b6.json
