#!/bin/bash
# round 5, call 8: ablation of the pair4 kernel's plane loop on C5, tile 5x6x66 in 512 threads (tune builds libhj_vA<bits>.so; results of
# the ablated builds are wrong by construction).  bits: 1 no stencil / Hamiltonian arithmetic, 2 no LDS stencil reads, 8 no halo loads,
# 16 no halo LDS stores, 32 no barrier, 64 no y0 loads
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run8; rm -rf $out; mkdir -p $out
export C5_STEPS=20 C5_WARMUP=30
for rep in 1 2; do
  for v in 0 1 2 24 32 64 27 91; do
    echo "== ablate $v" >> $out/c5.txt
    HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vA$v.so timeout -k 10 200 python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1
  done
done
grep -v "amdgpu.ids" $out/c5.txt
