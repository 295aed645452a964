#!/bin/bash
# round 4, call 36: cells per thread / occupancy hint of the tiled term kernels (tune builds: R = 1 with launch bound 3 and 2) against the product (R = 2)
out=gpurun_out/r04_run36; mkdir -p $out
D=$PWD/levelsetpy_amd/csrc
for rep in 1 2; do for v in libhj_mi355x.so libhj_vTR1.so libhj_vTR1o2.so; do
  echo "== $v pass $rep" | tee -a $out/ab.txt
  HJ_LIB=$D/$v timeout -k 10 300 python3 tools/term_timing.py 201 2>&1 | grep "term" | cut -c1-60 | tee -a $out/ab.txt
done; done
