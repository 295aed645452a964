#!/bin/bash
# round 4, call 44: fraction of the roofline against grid size on the final build (as-shipped WENO5, odeCFL3), 51^3 ... 1025^3
out=gpurun_out/r04_run44; mkdir -p $out; : > $out/ab.txt
for n in 51 101 151 201 251 301 351 401 451 513 601 769 1025; do
  echo "== n=$n" >> $out/ab.txt
  HJ_DEBUG=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --no-also --steps 20 --repeats 5 --n $n >> $out/ab.txt 2> $out/last.err || { tail -2 $out/last.err; exit 1; }
  grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt
done
python3 - <<'PY'
import json
n = None
for ln in open("gpurun_out/r04_run44/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-9s %.4e  frac %.3f  us/step %8.1f  %s" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3, d["roofline"]["kernel"][:24]), end="  ")
    elif "tiling" in ln: print(ln.strip()[2:].replace("[hj] pair tiling NT=512 R=2 KH=2 PD=2 OCC=2 ", "")[:100])
PY
