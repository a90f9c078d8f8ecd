# round 5, run 13: the withdrawn cluster registration kernel with all members of a cluster on ONE XCD, next to a second
# process (tools/contention_reg.sh): does the cross-XCD path make the difference?
set -x
cd $GRAFT_REPO_ROOT
VARIANTS="xcd xcd xcd" NREG=2000 NCOMP=600 bash tools/contention_reg.sh > gpurun_out/r5_13_contention.txt 2>&1
cat gpurun_out/r5_13_contention.txt
