#!/bin/bash
# round 5, call 7: C5 on higher-occupancy shapes of the pair4 kernel (tune build libhj_vT4.so):
#   0: 768 x 2 pairs, 5x9x66 (3 waves/SIMD, 12 B scratch)   1: 768 x 2, 5x8x66   2: 1024 x 1 pair, 5x6x66 (4 waves/SIMD, 122 VGPRs)
#   3: 384 x 2, 4x5x66 (two workgroups per CU)               4: 512 x 2, 5x6x66 (the round's default so far)
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run7; rm -rf $out; mkdir -p $out
export C5_STEPS=20 C5_WARMUP=40 HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vT4.so
for rep in 1 2; do
  for v in 0 1 2 3 4; do
    echo "== tile $v" >> $out/c5.txt
    HJ_TILE4_SEL=$v HJ_DEBUG=1 timeout -k 10 200 python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1
  done
done
grep -v "amdgpu.ids" $out/c5.txt
