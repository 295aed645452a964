#!/bin/bash
# round 5, call 10: cache policy of the pair4 kernel's streams on C5 (tile 5x6x66, 512 threads): N0 default; Y2 y0 nt; O2 output nt; YO both;
# W2 own + y0 + output nt; H1 y0/out nt + halo sc0
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run10; rm -rf $out; mkdir -p $out
export C5_STEPS=20 C5_WARMUP=30
for rep in 1 2; do
  for v in N0 Y2 O2 YO W2 H1; do
    echo "== policy $v" >> $out/c5.txt
    HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vX$v.so timeout -k 10 200 python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1
  done
done
grep -v "amdgpu.ids" $out/c5.txt
