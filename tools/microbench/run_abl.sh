#!/bin/bash
cd $(dirname $0)
for f in wide_check_256_256 wide_check_v_*; do echo "== $f"; timeout -s KILL 120 ./$f 3570000 5 2>&1 | grep -E "TIMING|CYCLES|grid=8" | tail -6; done
