#!/bin/bash
# Round evidence bundle, run on the GPU box from the repo root: PMC passes of the dominant kernel first (so that the bench
# lines that follow can quote them), bench lines (configs[1], [2], [4], the 512-pair configs[3] shape on one GPU),
# rocprofv3 kernel stats, the whole GPU suite.  Everything lands in gpurun_out/final/ (copy what should be judged
# into profiles/).   bash tools/evidence.sh [quick|nopmc]
# (nopmc: everything but the counter passes -- for a refresh after changes that did not touch the profiled kernels)
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
RND=r05
W="BASELINE configs[1], 6 pairs per batch"
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
  python $R/tools/pmc_dominant.py "reduce_rows_kernel<64" "$W" $DBS > $O/reduce_rows_pmc.json
  python $R/tools/pmc_dominant.py "sparse_conv_wide_f16x2<64, 1, 2>" "$W" $DBS > $O/wide64_pmc.json
  python $R/tools/pmc_dominant.py "conv1_grid_mfma" "$W" $DBS > $O/conv1_pmc.json
  python $R/tools/pmc_dominant.py "knn_mfma_kernel<true>" "$W" $DBS > $O/knn_pmc.json
  python $R/tools/pmc_dominant.py "registration_kernel" "$W" $DBS > $O/registration_pmc.json
  rm -rf $O/pmc
  cp $O/dominant_pmc.json $R/profiles/${RND}_dominant_pmc.json    # the lines below quote it (roofline.traffic)
fi
# the driver's command: no flags (3 streams x 6 pairs, 5 warm-up + 100 timed steps, two input sets taken in turn, parity + CPU baseline)
timeout 900 $B > $O/bench_c1_default.json 2> $O/bench_c1_default.err
timeout 300 $B --pairs-per-step 4 --no-parity > $O/bench_c1_s3_b4.json 2> $O/bench_c1_s3_b4.err   # the default of rounds 1-4
timeout 300 $B --streams 1 --pairs-per-step 4 --no-parity > $O/bench_c1_s1_b4.json 2> $O/bench_c1_s1_b4.err
timeout 300 $B --streams 1 --pairs-per-step 1 --no-parity > $O/bench_c1_s1_b1.json 2> $O/bench_c1_s1_b1.err
# BASELINE configs[2] (KITTI-shaped, 13 k voxels per scan): batches of 4 and 8 pairs per stream
timeout 600 $B --kind outdoor --n-raw 120000 --voxel 0.3 --conv1-ks 5 --pairs-per-step 4 > $O/bench_c3.json 2> $O/bench_c3.err
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
timeout 300 rocprofv3 --kernel-trace -d $O/kt3 -o kt -- $B --no-parity --steps 5 > $O/kt3.log 2>&1
python $R/tools/rocpd_summary.py $O/kt3/kt_results.db $O/kernel_stats_s3_b6.csv
rm -rf $O/kt1 $O/kt3 $O/kt6
# the whole GPU suite with the parity tables (split operands vs f64; refinement vs the oracle, iteration-matched and free-running)
if [ "$1" != quick ]; then   # PYTEST_ARGS: a subset (default: the whole suite)
  (cd $R && DGR_PARITY_REPORT=$O/parity timeout 1800 python -m pytest ${PYTEST_ARGS:-tests} -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log)
fi
ls -la $O; cat $O/pytest_gpu.log; tail -c 600 $O/bench_c1_default.json
