#!/bin/bash
# round 4, call 18b: paired chunk directions at 201^3 / 221^3, four alternations on one box
out=gpurun_out/r04_run18; mkdir -p gpurun_out/r04_run18; : > $out/ab3.txt
for n in 201 221; do for rep in 1 2 3 4; do for pd in 0 1; do
  echo "== n=$n HJ_PAIR_DIRS=$pd pass $rep" >> $out/ab3.txt
  HJ_PAIR_DIRS=$pd timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --repeats 31 --steps 20 --n $n >> $out/ab3.txt 2> $out/last.err || tail -3 $out/last.err >> $out/ab3.txt
done; done; done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r04_run18/ab3.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-36s %.4e frac %.4f iqr %.4f us/launch %.2f" % (n, d["value"], d["roofline"]["frac"], d["repeats"]["iqr_over_median"], d["ms_per_step"] * 1e3 / 3))
PY
