#!/bin/bash
# round 3, run 49: fraction of the roofline against grid size (as-shipped WENO5, final build): where do round-quantisation dips sit?
out=gpurun_out/r03aw; mkdir -p $out; rm -rf $out/*
for n in 121 141 161 181 201 221 241 261 281 301 321 341 361 381 401 431 461; do
  echo "== n=$n" >> $out/ab.txt
  HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --steps 20 --repeats 3 --n $n >> $out/ab.txt 2> $out/last.err || exit 1
  grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt
done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r03aw/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-8s %.4e  frac %.3f  us/step %.1f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3), end="  ")
    elif "tiling" in ln: print(ln.strip()[2:].replace("[hj] pair tiling NT=512 R=2 KH=2 PD=2 OCC=2 ", "")[:110])
PY
