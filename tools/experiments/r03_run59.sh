#!/bin/bash
# round 3, run 59: C5 with the halo slots' ghost handling compiled out (all plane axes periodic: MODE 3 / 4): the default (256,2,10) shape
# with and without it; 512 threads x 1 pair at 4 waves per SIMD (two 8-wave workgroups per CU, 128-VGPR cap) and at 2
out=gpurun_out/r03bg; mkdir -p $out; rm -rf $out/*
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
print(json.dumps({"also": {"C5": {"value": s["value"], "roofline_frac": s["value"] * 32 / 3 / 8e12, "kernel": r["kernel"]}}}))
PY
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 400 python /tmp/c5only.py >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err; exit 1; }; grep -E "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
run HJ_NO_NG=1
run HJ_X=0
run HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=5 HJ_PAIR_OCC=4
run HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=5 HJ_PAIR_OCC=2
run HJ_NO_NG=1
run HJ_X=0
python - <<'PY'
import json
for ln in open("gpurun_out/r03bg/ab.txt"):
    if ln.startswith("=="): print(ln.rstrip()[:120], end="  "); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("  ".join("%s %.4f (%.4e)" % (k.split()[0], v["roofline_frac"], v["value"]) for k, v in d["also"].items()), end="  ")
    elif "tiling" in ln: print(ln.strip()[2:][20:140])
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "4d or pendulum or c5" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
