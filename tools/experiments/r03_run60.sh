#!/bin/bash
# round 3, run 60: ablation of the C5 launch (pair kernel (256,2,10), tune builds -DHJ_ABLATE=bits; results are WRONG by construction):
# 1 no Hamiltonian / dissipation arithmetic, 2 no stencil LDS reads, 8 no halo loads, 16 no halo LDS stores, NOSYNC no barrier
out=gpurun_out/r03bh; mkdir -p $out; rm -rf $out/*
cat > /tmp/c5only.py <<'PY'
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid
wl = bench.workload(L, _ffi, torch, "C5", None, None, 0)
r = bench.time_single(torch, _ffi, DeviceGrid, wl, 10, 3, 3, 20)
s = bench.summarize(r, 10)
print("   %.4e cell-substeps/s  %.3f ms/launch  frac %.4f  %s" % (s["value"], s["ms_per_step"] / 3, s["value"] * 32 / 3 / 8e12, r["kernel"]))
PY
for v in AB0 AB1 AB2 AB8 AB16 AB24 AB27 ABNS; do
  echo "== $v" >> $out/ab.txt
  HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_v$v.so HJ_AUTOTUNE=0 timeout -k 10 300 python /tmp/c5only.py >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err >> $out/ab.txt; }
done
cat $out/ab.txt
