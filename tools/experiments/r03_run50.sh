#!/bin/bash
# round 3, run 50: medium grids (121^3 ... 181^3, 1.8 - 6 M cells): kernel configurations on the final build
out=gpurun_out/r03ax; mkdir -p $out; rm -rf $out/*
run() { echo "== n=$N $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --steps 40 --repeats 5 --n $N >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err >> $out/ab.txt; }; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for N in 121 141 161 181; do
  run HJ_X=0
  run HJ_PAIR=2
  run HJ_PAIR=0
  run HJ_PAIR=2 HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_PAIR_KH=2
  run HJ_PAIR=0 HJ_NT=512 HJ_R=4 HJ_KH=2 HJ_OCC=2 HJ_PD=2
  run HJ_PAIR=0 HJ_NT=256 HJ_R=2 HJ_KH=2 HJ_OCC=2 HJ_PD=2
  run HJ_PAIR=0 HJ_NT=512 HJ_R=1 HJ_KH=1 HJ_OCC=4 HJ_PD=2
  run HJ_PAIR=2 HJ_PAIR_RING=1
done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r03ax/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-70s %.4e  frac %.3f  us/step %.1f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3), end="  ")
    elif "tiling" in ln: print(ln.strip()[2:][5:100])
    else: print(n, ln.strip()[:100])
PY
