#!/bin/bash
# round 3, run 58: repeatability of the launch-time tile-shape choice: six fresh processes at 513^3 and 401^3, the shapes chosen and the result
out=gpurun_out/r03bf; mkdir -p $out; rm -rf $out/*
for n in 513 401; do for i in 1 2 3 4 5 6; do
  echo "== n=$n run $i" >> $out/ab.txt
  HJ_AUTOTUNE_LOG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --n $n --steps 20 --warmup 5 --repeats 3 >> $out/ab.txt 2> $out/last.err || exit 1
  grep "chosen" $out/last.err >> $out/ab.txt
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/r03bf/ab.txt"):
    if ln.startswith("=="): print(ln.strip(), end="  ")
    elif ln.startswith("{"):
        d = json.loads(ln); print("%.4e frac %.4f" % (d["value"], d["roofline"]["frac"]), end="  ")
    elif "chosen" in ln: print(ln.split("stage")[1].split("ntiles")[0].strip().replace(": ", " "), end=" | ")
    if ln.startswith("=="): pass
print()
PY
