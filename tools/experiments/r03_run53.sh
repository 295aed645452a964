#!/bin/bash
# round 3, run 53: heavy stencils (ENO3, intended WENO5) on medium grids: pair (256,1) against one-cell-per-lane (256,2)
out=gpurun_out/r03ba; mkdir -p $out; rm -rf $out/*
for S in ENO3 WENO5 ENO2; do for N in 141 161 181 201 241; do for P in 1 0; do
  echo "== $S n=$N HJ_PAIR=$P" >> $out/ab.txt
  HJ_PAIR=$P timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --steps 30 --repeats 3 --n $N --scheme $S >> $out/ab.txt 2> $out/last.err || exit 1
done; done; done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r03ba/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-30s %.4e  frac %.3f  us/step %.1f  %s" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3, d["roofline"]["kernel"][:22]))
PY
