#!/bin/bash
# round 3, run 23: 513^3 / 401^3 with the tile capped below the configuration's 2048 cells (fewer round-quantisation losses:
# 133 tiles x 9 chunks = 1197 workgroups = 4.68 rounds of 256 today)
out=gpurun_out/r03w; mkdir -p $out; rm -rf $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 513 401; do
  for tc in 0 1056 1130 1400 1700; do EXTRA="--n $n" run HJ_TILE_CELLS=$tc; done
  EXTRA="--n $n" run HJ_TARGET_BLOCKS=1280
  EXTRA="--n $n" run HJ_TARGET_BLOCKS=1024
  EXTRA="--n $n" run HJ_TARGET_BLOCKS=768
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03w/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
