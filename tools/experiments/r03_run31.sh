#!/bin/bash
# round 3, run 31: C5 (129^4 fp32): more cells per thread / smaller workgroups, both tiled kernels (tune build libhj_vC5Q.so)
out=gpurun_out/r03ae; mkdir -p $out; rm -rf $out/*
export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vC5Q.so
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 HJ_AUTOTUNE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 8 --repeats 2 --also C5 >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; tail -1 $out/last.err >> $out/ab.txt; }
run HJ_PAIR=1
run HJ_NT=512 HJ_R=2 HJ_KH=4 HJ_OCC=2 HJ_PD=2
run HJ_NT=512 HJ_R=4 HJ_KH=7 HJ_OCC=2 HJ_PD=2
run HJ_NT=1024 HJ_R=2 HJ_KH=4 HJ_OCC=2 HJ_PD=2
run HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=10
run HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=10 HJ_PAIR_RING=1
run HJ_PAIR_NT=512 HJ_PAIR_R=4 HJ_PAIR_KH=11
python - <<'PY'
import json
for ln in open("gpurun_out/r03ae/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:220]); continue
    d = json.loads(ln)
    for k, v in (d.get("also") or {}).items(): print("      also", k, {x: v.get(x) for x in ("value", "ms_per_step", "roofline_frac", "kernel", "error")})
PY
