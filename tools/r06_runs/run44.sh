#!/bin/bash
# round 6, run 44: after the bundle at 08bc4d4 -- the default line again (naps in the last stretch of a long wait: host CPU),
# the exact-f32 line (roofline.frac priced against the whole GPU where the kernel does not stamp its span), and the
# configurations the CU partition does NOT help, on plain streams next to the bundle's partitioned lines
R=$PWD; O=$R/gpurun_out/run44; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
export DGR_EVIDENCE_COMMIT=$(tr -d '\n' < $R/tools/COMMIT)
B="python $R/bench.py"
show() { python - <<P
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
d['commit']='$DGR_EVIDENCE_COMMIT'; json.dump(d, open('$1','w'))
print('$1'.split('/')[-1], 'pairs/s %.1f ms/step %.2f' % (d['value'], d['ms_per_step']), 'frac %.3f (%s CUs) host cpu %.4f parity %s' % (r['frac'], r.get('cus_of_a_launch'), d['host_cpu_s_per_step_per_rank'], d['config'].get('parity_ok')))
P
}
(cd $R && timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3)
timeout 900 $B > $O/bench_c1_default.json 2> $O/bench_c1_default.err; show $O/bench_c1_default.json
DGR_EXACT_F32=1 timeout 600 $B --steps 15 --warmup 3 --no-parity > $O/bench_c1_exact_f32.json 2> $O/bench_c1_exact_f32.err; show $O/bench_c1_exact_f32.json
P3="--streams 3 --no-cu-partition --no-parity"
timeout 600 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 4 $P3 > $O/bench_c3_plain.json 2> $O/bench_c3_plain.err; show $O/bench_c3_plain.json
timeout 600 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 8 $P3 > $O/bench_c3_b8_plain.json 2> $O/bench_c3_b8_plain.err; show $O/bench_c3_b8_plain.json
timeout 600 $B --n-raw 200000 --voxel 0.025 --pairs-per-step 1 $P3 > $O/bench_c5_plain.json 2> $O/bench_c5_plain.err; show $O/bench_c5_plain.json
timeout 300 $B --streams 1 --pairs-per-step 1 --no-parity > $O/bench_c1_s1_b1.json 2> $O/bench_c1_s1_b1.err; show $O/bench_c1_s1_b1.json
