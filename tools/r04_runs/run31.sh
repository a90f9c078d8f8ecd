#!/bin/bash
# run 31 (branch dense-ring): FCGF features of the ring build against the shipped library, bit for bit
O=gpurun_out/run31; mkdir -p $O
python tools/r04_runs/ab_bits.py $O/ring.bin
DGR_HIP_LIB=$PWD/lib_main/libdgr_hip.so python tools/r04_runs/ab_bits.py $O/main.bin
cmp $O/ring.bin $O/main.bin && echo BITWISE_IDENTICAL
rm -f $O/*.bin
