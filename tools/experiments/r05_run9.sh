#!/bin/bash
# round 5, call 9: prefetch depth of the own cells / y0 in the pair4 kernel (register sets, loop unrolled by their common multiple):
# libhj_vD<own><y0>.so, tile 5x6x66 in 512 threads; correctness of each by the pair4 tests first
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run9; rm -rf $out; mkdir -p $out
for v in 22 42 44 33; do
  HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vD$v.so timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -x -q -k "every_built_tile and 0-" > $out/t$v.log 2>&1; echo "D$v tests: $(tail -1 $out/t$v.log)"
done
export C5_STEPS=20 C5_WARMUP=30
for rep in 1 2; do
  for v in 22 42 44 33; do
    echo "== depth $v" >> $out/c5.txt
    HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vD$v.so timeout -k 10 200 python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1
  done
done
grep -v "amdgpu.ids" $out/c5.txt
