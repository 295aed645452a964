#!/bin/bash
# round 3, run 34: C5 through smaller pair workgroups (more independent workgroups per CU); tune build libhj_vC5R.so
out=gpurun_out/r03ah; mkdir -p $out; rm -rf $out/*
export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vC5R.so
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 HJ_AUTOTUNE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 10 --repeats 3 --also C5 >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; tail -1 $out/last.err >> $out/ab.txt; }
run HJ_PAIR=1
run HJ_PAIR_NT=128 HJ_PAIR_R=2 HJ_PAIR_KH=13
run HJ_PAIR_NT=128 HJ_PAIR_R=2 HJ_PAIR_KH=14 HJ_PAIR_OCC=4
run HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_PAIR_KH=7
run HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_PAIR_KH=7 HJ_PAIR_OCC=3
python - <<'PY'
import json
for ln in open("gpurun_out/r03ah/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:220]); continue
    d = json.loads(ln)
    for k, v in (d.get("also") or {}).items(): print("      also", k, {x: v.get(x) for x in ("value", "ms_per_step", "roofline_frac", "kernel", "error")})
PY
