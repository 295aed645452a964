#!/bin/bash
# round 3, run 47: stage fusion in 2-D (C3, 4096^2 ENO3): a 1-D tile has almost no ring, so fusing stages 1+2 costs ~no redundant arithmetic
out=gpurun_out/r03au; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 20 --repeats 3 --also C3 >> $out/ab.txt 2> $out/last.err; grep -E "fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; tail -1 $out/last.err >> $out/ab.txt; }
run HJ_FUSE12=0
run HJ_FUSE12=1
run HJ_FUSE12=1 HJ_F12_R=3
run HJ_FUSE12=1 HJ_F12_PAIR=0
python - <<'PY'
import json
for ln in open("gpurun_out/r03au/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:220]); continue
    d = json.loads(ln)
    print("   main 101^3:", d["value"], d["roofline"]["kernel"])
    for k, v in (d.get("also") or {}).items(): print("      also", k, {x: v.get(x) for x in ("value", "ms_per_step", "roofline_frac", "kernel", "error")}, v["roofline"]["kernel"])
PY
