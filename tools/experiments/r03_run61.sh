#!/bin/bash
# round 3, run 61: 4-D pair kernel with the halo layers of the non-contiguous axes dealt as PAIRS (5 pair + 1 single slots per thread
# instead of 10 single ones): the 4-D tests (bitwise against the one-cell-per-lane and the direct kernel, ghost cells included), C5
out=gpurun_out/r03bi; mkdir -p $out; rm -rf $out/*
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "4d or pendulum or c5" > $out/test.txt 2>&1; rc=$?; echo "rc=$rc" >> $out/test.txt; tail -4 $out/test.txt
[ $rc -eq 0 ] || exit 1
cat > /tmp/c5only.py <<'PY'
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid
wl = bench.workload(L, _ffi, torch, "C5", None, None, 0)
r = bench.time_single(torch, _ffi, DeviceGrid, wl, 10, 3, 5, 30)
s = bench.summarize(r, 10)
print("   %.4e cell-substeps/s  %.3f ms/launch  frac %.4f  %s" % (s["value"], s["ms_per_step"] / 3, s["value"] * 32 / 3 / 8e12, r["kernel"]))
PY
for i in 1 2 3; do HJ_DEBUG=1 timeout -k 10 300 python /tmp/c5only.py 2> $out/last.err | tee -a $out/ab.txt; done
grep "pair tiling" $out/last.err | head -2
