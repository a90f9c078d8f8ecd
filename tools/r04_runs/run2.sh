set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_maps.py tests/test_gpu_resunet.py tests/test_gpu_split_f64.py tests/test_gpu_numeric_modes.py tests/test_gpu_fullsize.py -m gpu -x -q -s 2>&1 | tail -40) > $O/pytest.log 2>&1
timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_s1_b4.json 2> $O/bench_s1_b4.err
DGR_KMAP_GENERIC8=1 timeout 300 python bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 20 > $O/bench_s1_b4_generic8.json 2> $O/bench_s1_b4_generic8.err
DGR_HIP_LIB=$PWD/deepglobalregistration_amd/lib_r3/libdgr_hip.so timeout 600 python tests/aux/split_f64_dump.py $O/dump_r3.npz > $O/dump_r3.log 2>&1
timeout 600 python tests/aux/split_f64_dump.py $O/dump_new.npz > $O/dump_new.log 2>&1
python - <<PY > $O/ab_wide.txt 2>&1
import numpy as np
a=np.load('$O/dump_r3.npz'); b=np.load('$O/dump_new.npz')
for k in a.files:
    if k=='kinds': print('kinds r3', a[k].tolist()); print('kinds new', b[k].tolist()); continue
    print(k, 'bitwise', bool((a[k]==b[k]).all()), 'max|d|/max|a|', float(np.abs(a[k].astype(np.float64)-b[k]).max()/max(1e-30,np.abs(a[k]).max())))
PY
rm -f $O/dump_r3.npz $O/dump_new.npz
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt1 -o kt -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --pairs-per-step 4 --no-parity --steps 5 > $GRAFT_REPO_ROOT/$O/kt1.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py $O/kt1/kt_results.db $O/kernel_stats_s1_b4.csv
rm -rf $O/kt1
tail -5 $O/pytest.log; cat $O/ab_wide.txt | head -40; tail -c 300 $O/bench_s1_b4.err
