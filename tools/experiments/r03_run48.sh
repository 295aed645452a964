#!/bin/bash
# round 3, run 48: the headline (201^3) once more through the knobs of the pair kernel on the final build: ring / ring depth / chunking
out=gpurun_out/r03av; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --steps 40 --repeats 7 >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
run HJ_X=0
run HJ_PAIR_RING=0
run HJ_PAIR_AH=1
run HJ_PAIR_AH=3
run HJ_TARGET_BLOCKS=231
run HJ_TARGET_BLOCKS=210
run HJ_TARGET_BLOCKS=273
run HJ_MIN_CHUNK=20
run HJ_LDS_PAD=0
run HJ_X=1
python - <<'PY'
import json
for ln in open("gpurun_out/r03av/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.4f  us/step %.2f  min/max %.4e %.4e" % (d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3, d["repeats"]["value_min"], d["repeats"]["value_max"]))
PY
