#!/bin/bash
# round 5, call 2: C5 with the compile-time-tile kernel (hj_fused4v.h) against the generic pair kernel (HJ_PAIR4=0), same box, alternating
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run2; rm -rf $out; mkdir -p $out
export C5_STEPS=20 C5_WARMUP=40
for rep in 1 2 3; do
  for v in 1 0; do
    echo "== HJ_PAIR4=$v" >> $out/c5.txt
    HJ_PAIR4=$v HJ_DEBUG=1 python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1
  done
done
grep -v "amdgpu.ids" $out/c5.txt
