#!/bin/bash
# round 6, run 47: the whole GPU suite and smoke() at the final commit
R=$PWD; O=$R/gpurun_out/run47; mkdir -p $O; rm -rf $O/*
cd $R
C=$(tr -d '\n' < tools/COMMIT)
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep "smoke ok" $O/smoke.log
DGR_PARITY_REPORT=$O/parity timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log
echo "# commit $C" >> $O/pytest_gpu.log; cat $O/pytest_gpu.log
