#!/bin/bash
# run 15: the whole GPU suite at the final HEAD (78 tests), parity report kept
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/run15; mkdir -p $O
export TMPDIR=/tmp
(DGR_PARITY_REPORT=$O/parity timeout 840 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log)
cat $O/pytest_gpu.log
