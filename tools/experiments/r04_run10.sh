#!/bin/bash
# round 4, call 10: does the launch-time tile-shape choice (on from 40 M cells) pay on the 201^3-class grids?  HJ_AUTOTUNE_MIN_MCELLS=6
out=gpurun_out/r04_run10; mkdir -p gpurun_out/r04_run10
for n in 201 251 301; do for mc in 40 6; do
  echo "== n=$n HJ_AUTOTUNE_MIN_MCELLS=$mc" >> $out/ab.txt
  HJ_AUTOTUNE_MIN_MCELLS=$mc HJ_AUTOTUNE_LOG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic --no-also --repeats 15 --steps 30 --n $n >> $out/ab.txt 2> $out/last.err
  grep -h "autotune" $out/last.err | grep -v "^$" | head -24 >> $out/ab.txt
done; done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r04_run10/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-40s %.4e  frac %.4f  us/launch %.2f  iqr %.4f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3, d["repeats"]["iqr_over_median"]))
    else: print("     ", ln.strip()[:150])
PY
