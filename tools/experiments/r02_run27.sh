#!/bin/bash
# C3 (4096^2 ENO3) and 3-D ENO3 / intended WENO5: pair-kernel configuration candidates
out=gpurun_out/r02aa; mkdir -p $out; rm -f $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --repeats 3 --also C3,ENO3,WENO5 >> $out/ab.txt 2> $out/last.err; grep "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -4 >> $out/ab.txt; }
run HJ_PAIR=1
run HJ_PAIR=0
run HJ_PAIR=1 HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=1 HJ_PAIR_OCC=2
run HJ_PAIR=1 HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_PAIR_KH=2 HJ_PAIR_OCC=3
python - <<'PY'
import json
for ln in open("gpurun_out/r02aa/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f" % (d["value"], d["roofline"]["frac"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
