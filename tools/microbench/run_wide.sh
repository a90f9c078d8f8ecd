#!/bin/bash
# accuracy + timing of the wide-layer kernel harness (tools/microbench/wide_check.hip), every shape; on the GPU box
cd $(dirname $0)
for cfg in 256_256 128_128 64_128 128_256 256_128; do
  [ -x ./wide_check_$cfg ] || continue
  P=3570000; [ $cfg = 128_128 ] && P=790000; [ $cfg = 64_128 ] && P=205000; [ $cfg = 128_256 ] && P=279000; [ $cfg = 256_128 ] && P=279000
  timeout -s KILL 120 ./wide_check_$cfg $P 5 2>&1 | tail -8
done
for cfg in 256_256 128_128; do
  P=3570000; [ $cfg = 128_128 ] && P=790000
  [ -x old/old_time_$cfg ] && timeout -s KILL 120 old/old_time_$cfg $P 5 2>&1 | tail -2
done
