#!/bin/bash
# C3 (4096^2 ENO3): two pairs per thread, chunk counts
out=gpurun_out/r02ae; mkdir -p $out; rm -f $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --repeats 1 --also C3 >> $out/ab.txt 2> $out/last.err; grep "pair tiling" $out/last.err | grep "E=([0-9]*,0,0)" | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
run HJ_PAIR=1
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_OCC=2
run HJ_PAIR=1 HJ_TARGET_BLOCKS=512
run HJ_PAIR=1 HJ_TARGET_BLOCKS=1536
run HJ_PAIR=1 HJ_TARGET_BLOCKS=3072
python - <<'PY'
import json
for ln in open("gpurun_out/r02ae/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
