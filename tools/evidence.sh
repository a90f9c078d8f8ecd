#!/bin/bash
# Round evidence bundle, run on the GPU box from the repo root: PMC passes of the dominant kernel first (so that the bench
# lines that follow can quote them), bench lines (configs[1], [2], [4], the 512-pair configs[3] shape on one GPU),
# rocprofv3 kernel stats, the whole GPU suite.  Everything lands in gpurun_out/final/ (copy what should be judged
# into profiles/).   bash tools/evidence.sh [quick|nopmc]
# (nopmc: everything but the counter passes -- for a refresh after changes that did not touch the profiled kernels)
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
RND=r06
W="BASELINE configs[1], 6 pairs per batch"
# the commit these files belong to: written into tools/COMMIT by the caller before gpurun snapshots the tree (the box has
# no .git): `git rev-parse --short HEAD > tools/COMMIT; git diff --quiet || echo "+dirty" >> tools/COMMIT`
COMMIT=$(tr -d '\n' < $R/tools/COMMIT 2>/dev/null || echo unknown)
export DGR_EVIDENCE_COMMIT=$COMMIT
echo "evidence bundle of commit $COMMIT" > $O/COMMIT.txt
# one stream, the default batch of 6 pairs: names the dominant kernel
timeout 300 $B --streams 1 --no-parity > $O/bench_c1_s1_b6.json 2> $O/bench_c1_s1_b6.err
if [ "$1" != nopmc ]; then
  # PMC: one pass per counter set, never combined with other trace domains
  K=$(python -c "import json;print(json.loads(open('$O/bench_c1_s1_b6.json').read().strip().splitlines()[-1])['roofline']['kernel'])")
  i=0; DBS=""
  for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc -o p$i -- $B --streams 1 --steps 2 --warmup 1 --no-parity > $O/pmc_$i.log 2>&1
    DBS="$DBS $O/pmc/p${i}_results.db"
  done
  python $R/tools/pmc_dominant.py "$K" "$W" $DBS > $O/dominant_pmc.json
  python $R/tools/pmc_dominant.py "sparse_conv_os<64, 64" "$W" $DBS > $O/os_conv_pmc.json
  python $R/tools/pmc_dominant.py "sparse_conv_dense_f16x2<64, 64" "$W" $DBS > $O/dense_conv_pmc.json
  python $R/tools/pmc_dominant.py "sparse_conv_up_f16x2<128" "$W" $DBS > $O/up_conv_pmc.json
  python $R/tools/pmc_dominant.py "reduce_rows_kernel<64" "$W" $DBS > $O/reduce_rows_pmc.json
  python $R/tools/pmc_dominant.py "sparse_conv_wide_f16x2<64, 1, 2>" "$W" $DBS > $O/wide64_pmc.json
  python $R/tools/pmc_dominant.py "conv1_grid_mfma" "$W" $DBS > $O/conv1_pmc.json
  python $R/tools/pmc_dominant.py "knn_mfma_kernel<true>" "$W" $DBS > $O/knn_pmc.json
  python $R/tools/pmc_dominant.py "registration_kernel" "$W" $DBS > $O/registration_pmc.json
  python - <<P
import json
d = json.load(open('$O/dominant_pmc.json')); d['commit'] = '$COMMIT'; json.dump(d, open('$O/dominant_pmc.json', 'w'), indent=1)
P
  rm -rf $O/pmc
  cp $O/dominant_pmc.json $R/profiles/${RND}_dominant_pmc.json    # the lines below quote it (roofline.traffic)
fi
# the driver's command: no flags (4 contexts x 6 pairs, each context on its own quarter of the compute units; 5 warm-up + 100
# timed steps, two input sets taken in turn, parity + CPU baseline)
timeout 900 $B > $O/bench_c1_default.json 2> $O/bench_c1_default.err
# the default of rounds 5-6 until the CU partition: three plain streams x 6 pairs competing for all compute units; and 3 x 4 (rounds 1-4)
timeout 300 $B --streams 3 --no-cu-partition --no-parity > $O/bench_c1_s3_b6_plain.json 2> $O/bench_c1_s3_b6_plain.err
timeout 300 $B --streams 3 --no-cu-partition --pairs-per-step 4 --no-parity > $O/bench_c1_s3_b4.json 2> $O/bench_c1_s3_b4.err
timeout 300 $B --streams 1 --pairs-per-step 4 --no-parity > $O/bench_c1_s1_b4.json 2> $O/bench_c1_s1_b4.err
timeout 300 $B --streams 1 --pairs-per-step 1 --no-parity > $O/bench_c1_s1_b1.json 2> $O/bench_c1_s1_b1.err
# BASELINE configs[2] (KITTI-shaped, 13 k voxels per scan): batches of 4 and 8 pairs per stream
timeout 600 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 4 > $O/bench_c3.json 2> $O/bench_c3.err
# ... and the configurations the CU partition does NOT help (small batches: configs[2]; one big pair per batch: configs[4]) on
# three plain streams next to their partitioned lines
P3="--streams 3 --no-cu-partition --no-parity"
timeout 600 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 4 $P3 > $O/bench_c3_plain.json 2> $O/bench_c3_plain.err
timeout 600 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 8 $P3 > $O/bench_c3_b8_plain.json 2> $O/bench_c3_b8_plain.err
timeout 600 $B --n-raw 200000 --voxel 0.025 --pairs-per-step 1 $P3 > $O/bench_c5_plain.json 2> $O/bench_c5_plain.err
timeout 600 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 8 --no-parity > $O/bench_c3_b8.json 2> $O/bench_c3_b8.err
timeout 600 $B --n-raw 200000 --voxel 0.025 --pairs-per-step 1 --no-parity > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 $B --n-raw 200000 --voxel 0.025 --pairs-per-step 1 --no-parity --no-refine > $O/bench_c5_norefine.json 2> $O/bench_c5_norefine.err
if [ "$1" != quick ]; then
  timeout 900 $B --total-pairs 512 --steps 2 --warmup 1 --no-parity > $O/bench_c4shape_512pairs_1gpu.json 2> $O/bench_c4shape.err
  timeout 300 $B --from-host --no-parity > $O/bench_c1_from_host.json 2> $O/bench_c1_from_host.err
fi
# the reference arithmetic (exact-f32 MFMA kernels) on the same workload, register() as shipped (ICP on), and every pair
# forced through the safeguard
DGR_EXACT_F32=1 timeout 600 $B --steps 15 --warmup 3 --no-parity > $O/bench_c1_exact_f32.json 2> $O/bench_c1_exact_f32.err
timeout 600 $B --full-register --no-parity > $O/bench_c1_full_register.json 2> $O/bench_c1_full_register.err
timeout 600 $B --force-safeguard --no-parity --steps 3 --warmup 1 > $O/bench_c1_force_safeguard.json 2> $O/bench_c1_force_safeguard.err
# kernel stats: the single-stream commands give durations free of time-slicing (comparable with roofline.avg_launch_us)
timeout 300 rocprofv3 --kernel-trace -d $O/kt1 -o kt -- $B --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $O/kt1.log 2>&1
python $R/tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv --trace sparse_conv $O/conv_trace_s1_b4.csv
timeout 300 rocprofv3 --kernel-trace -d $O/kt6 -o kt -- $B --streams 1 --no-parity --steps 5 > $O/kt6.log 2>&1
python $R/tools/rocpd_summary.py $O/kt6/kt_results.db $O/kernel_stats_s1_b6.csv
grep '^{' $O/kt6.log | tail -1 > $O/bench_c1_s1_b6_under_rocprof.json
timeout 300 rocprofv3 --kernel-trace -d $O/kt3 -o kt -- $B --no-parity --steps 5 > $O/kt3.log 2>&1
python $R/tools/rocpd_summary.py $O/kt3/kt_results.db $O/kernel_stats_s4_b6.csv --trace sparse_conv_wide_f16x2 $O/wide_trace_s4_b6.csv
grep '^{' $O/kt3.log | tail -1 > $O/bench_c1_s4_b6_under_rocprof.json
# every launch of that run with its stream, and of the one-stream run: what the contexts do with the GPU's time (tools/timeline_stats.py)
python $R/tools/timeline_dump.py $O/kt3/kt_results.db $O/timeline_s4_b6.csv.gz 2> /dev/null
python $R/tools/timeline_dump.py $O/kt6/kt_results.db $O/timeline_s1_b6.csv.gz 2> /dev/null
python $R/tools/timeline_stats.py $O/timeline_s4_b6.csv.gz $O/timeline_s1_b6.csv.gz > $O/timeline_s4_vs_s1.txt 2>&1
# the line printed INSIDE that profiled run against the profiler's own table (round-5 verdict, task 1b).  The line's
# roofline.avg_launch_us is the mean of the spans the kernel stamped in the multi-stream region AFTER the timed steps -- a few of
# the process's launches, next to whatever the other two streams happened to run (3.2 - 6 ms from region to region) --, so
# it is held to the profiler LAUNCH BY LAUNCH: every stamped span must have its own rocprofv3 record of the same duration
# (order-preserving assignment of the sorted spans to the sorted records); the two plain averages are printed next to it
python - <<P | tee $O/line_vs_rocprof.txt
import csv, json
line = [l for l in open('$O/kt3.log') if l.startswith('{')][-1]
d = json.loads(line); k = d['roofline']['kernel']; us = d['roofline']['avg_launch_us']
rows = [r for r in csv.DictReader(open('$O/kernel_stats_s4_b6.csv')) if k in r['Name']]
avg = float(rows[0]['AverageNs']) / 1e3
S = sorted(d['roofline_detail'].get('dominant_spans_us_timed_config', []))
Rr = sorted(float(r['DurationNs']) / 1e3 for r in csv.DictReader(open('$O/wide_trace_s4_b6.csv')) if k in r['Name'])
msg = f'commit $COMMIT: {k}: bench line (timed stream configuration, under rocprofv3) mean of {len(S)} stamped spans {us:.1f} us; rocprofv3 --kernel-trace average over ALL {rows[0]["Calls"]} calls of the process {avg:.1f} us'
if S and len(Rr) >= len(S):
    INF = float('inf')
    cost = [[INF] * (len(Rr) + 1) for _ in range(len(S) + 1)]
    for j in range(len(Rr) + 1):
        cost[0][j] = 0.0
    for i in range(1, len(S) + 1):
        for j in range(i, len(Rr) + 1):
            cost[i][j] = min(cost[i][j - 1], cost[i - 1][j - 1] + abs(S[i - 1] - Rr[j - 1]) / Rr[j - 1])
    mean_dev = cost[len(S)][len(Rr)] / len(S) * 100
    msg += f'; launch by launch: every stamped span matched to its own profiler record, mean deviation {mean_dev:.2f} % ({"OK" if mean_dev <= 3.0 else "MORE THAN 3 %"})'
print(msg)
P
rm -rf $O/kt1 $O/kt3 $O/kt6
(cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log)
# the whole GPU suite with the parity tables (split operands vs f64; refinement vs the oracle, iteration-matched and free-running)
if [ "$1" != quick ]; then   # PYTEST_ARGS: a subset (default: the whole suite)
  (cd $R && DGR_PARITY_REPORT=$O/parity timeout 1800 python -m pytest ${PYTEST_ARGS:-tests} -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log)
fi
# stamp the commit into every JSON / text artefact of the bundle
python - <<P
import glob, json
for f in glob.glob('$O/*.json'):
    txt = open(f).read().strip()
    if not txt:
        continue
    try:
        d, multi = json.loads(txt), True
    except Exception:
        try:
            d, multi = json.loads(txt.splitlines()[-1]), False
        except Exception:
            continue
    if isinstance(d, dict):
        d['commit'] = '$COMMIT'
        json.dump(d, open(f, 'w'), indent=1 if '\n' in txt and multi else None)
for f in glob.glob('$O/*.csv') + glob.glob('$O/parity/*.txt'):
    s = open(f).read()
    open(f, 'w').write(s + ('' if s.endswith('\n') else '\n') + '# commit $COMMIT\n')
P
ls -la $O; cat $O/pytest_gpu.log; tail -c 600 $O/bench_c1_default.json
