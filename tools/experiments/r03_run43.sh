#!/bin/bash
# round 3, run 43: C3 (4096^2 ENO3, NumPy operation order) with two pairs per thread in 512-thread workgroups (fits in 2-D: 215-232 VGPRs)
out=gpurun_out/r03aq; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 20 --repeats 3 --also C3 >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; tail -1 $out/last.err >> $out/ab.txt; }
run HJ_X=0
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=2
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_RING=0
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_RING=1
run HJ_PAIR=0
python - <<'PY'
import json
for ln in open("gpurun_out/r03aq/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:220]); continue
    d = json.loads(ln)
    for k, v in (d.get("also") or {}).items(): print("      also", k, {x: v.get(x) for x in ("value", "ms_per_step", "roofline_frac", "kernel", "error")})
PY
