#!/bin/bash
# C5: XCD tile blocks (HJ_TB1 x HJ_TB2 positions x the axis-3 tiles) with round 5's kernel (the 4 x 4 default was tuned on round 4's 5x6x34 tiles)
mkdir -p gpurun_out
out=gpurun_out/r37_c5_tb.txt; : > $out
for tb in "4 4" "2 8" "8 2" "4 8" "8 4" "2 4" "4 2" "8 8" "0 0" "16 1" "1 16" "3 5" "5 3"; do
  set -- $tb
  v=$(HJ_TB1=$1 HJ_TB2=$2 C5_STEPS=12 C5_WARMUP=4 timeout -k 10 200 python tools/bench_configs.py c5 2>/dev/null | grep "^C5" | tail -1)
  echo "TB $1 x $2: $v" >> $out
done
cat $out
