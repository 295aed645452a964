#!/bin/bash
# round 3, run 63: stage-fused pair kernel as TWO independent 256-thread workgroups per CU (one wave per SIMD each, half-size tiles so that
# two of them fit the LDS) against the default 512-thread shape and the unfused step, 513^3
out=gpurun_out/r03bk; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 HJ_AUTOTUNE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --n 513 --steps 10 --repeats 3 >> $out/ab.txt 2> $out/last.err || { tail -2 $out/last.err >> $out/ab.txt; }; grep -E "fused12" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
run HJ_FUSE12=0
run HJ_FUSE12=1
run HJ_FUSE12=1 HJ_F12_NT=256 HJ_F12_R=4 HJ_F12_KH=4
for e in "13 52" "13 64" "11 52" "15 44" "9 64"; do set -- $e; run HJ_FUSE12=1 HJ_F12_NT=256 HJ_F12_R=4 HJ_F12_KH=4 HJ_F12_E1=$1 HJ_F12_E2=$2; done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r03bk/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-70s %.4e  frac %.4f  ms/step %.3f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"]), end="  ")
        print()
    else: print("     ", ln.strip()[:170])
PY
