#!/bin/bash
out=gpurun_out/r02k; mkdir -p $out; rm -f $out/small.txt
for n in 51 65 101 129 151; do
  for fd in 0 1; do
    echo "== n=$n HJ_FORCE_DIRECT=$fd" >> $out/small.txt
    HJ_FORCE_DIRECT=$fd timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --n $n --steps 100 --repeats 3 >> $out/small.txt 2>> $out/small.err
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02k/small.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"]))
PY
