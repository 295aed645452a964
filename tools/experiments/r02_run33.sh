#!/bin/bash
# LDS halo ring on the other workloads and sizes
out=gpurun_out/r02ag; mkdir -p $out; rm -f $out/*
for ring in 0 1 0 1; do
echo "== ring=$ring" >> $out/ab.txt
HJ_PAIR_RING=$ring python bench.py --no-cpu-baseline --no-live-traffic --steps 20 --also WENO5,ENO3,ENO2,C3,151,251 >> $out/ab.txt 2>> $out/ab.err
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02ag/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f" % (d["value"], d["roofline"]["frac"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
