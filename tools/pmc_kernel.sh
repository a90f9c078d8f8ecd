#!/bin/bash
# PMC counters of one kernel (name substring $1) while running command "$2..." -- separate rocprofv3 passes per
# counter set (SQ has 8 slots; never combined with sys/hip traces).  Run on the GPU box from the repo root.
R=$PWD; K="$1"; shift
O=$R/gpurun_out/pmc_k; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o p$i -- "$@" > $O/run_$i.log 2>&1)
  python $R/tools/pmc_summary.py $O/p${i}_results.db "$K"
done
rm -f $O/*.db
