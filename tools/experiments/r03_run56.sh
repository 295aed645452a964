#!/bin/bash
# round 3, run 56: 513^3 with the tuned 15x130 tile: chunk count (choose_chunks picks 9 chunks of 57 planes = 1260 workgroups = 4.92 rounds)
out=gpurun_out/r03bd; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 HJ_AUTOTUNE=0 HJ_FULL_ROWS=130 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --n 513 --steps 20 --repeats 3 >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err; exit 1; }; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
run HJ_X=0
for tb in 700 840 980 1120 1400 1540 1680 2048 2560; do run HJ_TARGET_BLOCKS=$tb; done
run HJ_X=1
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r03bd/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-28s %.4e  frac %.4f  ms/step %.4f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"]), end="  ")
    elif "tiling" in ln: print(ln.strip()[2:][45:130])
PY
