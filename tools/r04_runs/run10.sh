cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r10; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O -o p$i -- python bench.py --streams 1 --pairs-per-step 4 --steps 2 --warmup 1 --no-parity > $O/run_$i.log 2>&1)
  for K in kmap_place_hits kmap_bits_pruned6 kmap_colmask renumber_rows; do python $R/tools/pmc_summary.py $O/p${i}_results.db "$K"; done > $O/pmc_$i.txt 2>&1
done
rm -f $O/*.db
cat $O/pmc_*.txt
