#!/bin/bash
# round 6, run 20: would a parity-sorted row order pay for the transposed convs?  (fine rows of one parity class use the
# same 2^(odd dims) of the 27 offsets.)  Emulated at level 0 by sorting the INPUT rows by parity class: only conv2_tr (L18)
# sees class-homogeneous row blocks
R=$PWD; O=$R/gpurun_out/run20; mkdir -p $O; rm -rf $O/*
cd $R
run() { AB_TAG=$1 timeout 300 python tools/ab_fcgf.py > $O/ab_$1.txt 2>&1; echo "== $1"; grep -E "fwd ms|L18|L19|L20|L 1 |L 2 |L21|maps_3d" $O/ab_$1.txt; }
run default
AB_PARITY=1 run parity
