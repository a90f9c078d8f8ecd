# round 5, run 10: the cluster registration kernel next to a second process on the same GPU (tools/contention_reg.sh, base
# variant only: 2000 runs of the kernel alone on fixed inputs -- 11 k rows, a cluster of four -- beside a full-pipeline
# competitor): every run bit for bit the first one, no exchange time-out
set -x
cd $GRAFT_REPO_ROOT
VARIANTS="base base base" NREG=2000 NCOMP=600 bash tools/contention_reg.sh > gpurun_out/r5_10h_contention.txt 2>&1
cat gpurun_out/r5_10h_contention.txt
