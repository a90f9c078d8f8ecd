R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc3 -o p$i -- python $R/tools/layer_bench.py 10 > $R/gpurun_out/pmc3_$i.log 2>&1
done
for j in 1 2 3 4; do python $R/tools/pmc_summary.py $R/gpurun_out/pmc3/p${j}_results.db "sparse_conv_mfma_v2<256, 1, 4, 2, 2"; done; rm -f $R/gpurun_out/pmc3/*.db
