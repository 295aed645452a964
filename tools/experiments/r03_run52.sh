#!/bin/bash
# round 3, run 52: defaults after moving the pair kernel's threshold for the light stencils to 6.5 M cells; tests that depend on kernel choice
out=gpurun_out/r03az; mkdir -p $out; rm -rf $out/*
for N in 131 141 151 161 171 181 191 201; do
  echo "== n=$N" >> $out/ab.txt
  HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --steps 40 --repeats 5 --n $N >> $out/ab.txt 2> $out/last.err || exit 1
  grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt
done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r03az/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-10s %.4e  frac %.3f  us/step %.1f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3), end="  ")
    elif "tiling" in ln: print(ln.strip()[2:][5:110])
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
