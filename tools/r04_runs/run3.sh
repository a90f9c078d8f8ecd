set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_maps.py tests/test_gpu_resunet.py tests/test_gpu_dense_conv.py tests/test_gpu_split_f64.py tests/test_gpu_numeric_modes.py tests/test_gpu_fullsize.py -m gpu -x -q -s 2>&1 | tail -40) > $O/pytest.log 2>&1
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_a.json 2> $O/bench_a.err
DGR_KMAP_REGION8=3456 timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_b_region3456.json 2> $O/bench_b.err
DGR_KMAP_ROWMAJOR=1 timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_c_rowmajor.json 2> $O/bench_c.err
DGR_KMAP_ROWMAJOR=1 DGR_KMAP_REGION8=3456 timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_d_rowmajor_region3456.json 2> $O/bench_d.err
DGR_HIP_LIB=$PWD/deepglobalregistration_amd/lib_r3/libdgr_hip.so timeout 600 python tests/aux/split_f64_dump.py $O/dump_r3.npz > $O/dump_r3.log 2>&1
timeout 600 python tests/aux/split_f64_dump.py $O/dump_new.npz > $O/dump_new.log 2>&1
python - <<PY > $O/ab_wide.txt 2>&1
import numpy as np
a=np.load('$O/dump_r3.npz'); b=np.load('$O/dump_new.npz')
for k in a.files:
    if k=='kinds': continue
    x,y=np.ascontiguousarray(a[k]),np.ascontiguousarray(b[k])
    fin=np.isfinite(x)&np.isfinite(y)
    print(k, 'bit patterns equal', bool((x.view(np.uint32)==y.view(np.uint32)).all()), 'nonfinite', int((~np.isfinite(x)).sum()), int((~np.isfinite(y)).sum()), 'max|d| finite', float(np.abs(x[fin].astype(np.float64)-y[fin]).max()) if fin.any() else None)
PY
rm -f $O/dump_r3.npz $O/dump_new.npz
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt1 -o kt -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $GRAFT_REPO_ROOT/$O/kt1.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
tail -5 $O/pytest.log; cat $O/ab_wide.txt | head -50
