#!/bin/bash
# round 4, call 25: same-box A/B of the ENO3 selection forms (libhj_vOLDENO.so = the two-arm form of round 3, built from HEAD~; product = selected
# operand and coefficient), three alternations: C3 (4096^2 ENO3), 201^3 ENO3, 201^3 ENO2
out=gpurun_out/r04_run25; mkdir -p $out; : > $out/ab.txt
D=$PWD/levelsetpy_amd/csrc
for rep in 1 2 3; do for v in libhj_vOLDENO.so libhj_mi355x.so; do
  echo "== $v pass $rep" >> $out/ab.txt
  HJ_LIB=$D/$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also C3,ENO3,ENO2 --repeats 15 --steps 20 > $out/b.json 2> $out/b.err || tail -3 $out/b.err >> $out/ab.txt
  python3 - $out/b.json >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   " + " | ".join("%s %.4e (%.4f)" % (k, v["value"], v.get("roofline_frac") or 0) for k, v in d["also"].items() if isinstance(v, dict) and "value" in v))
PY
done; done
cat $out/ab.txt
