#!/bin/bash
# round 4, call 48: pair kernel with an 8-deep axis-0 queue shifted by two places once per loop pass (6 register moves per cell and pass instead of 12)
# against the previous build: pair-kernel tests, then 201^3 / 513^3 / C3 / C5 / intended WENO5, three alternations
out=gpurun_out/r04_run48; mkdir -p $out; : > $out/ab.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
D=$PWD/levelsetpy_amd/csrc
for rep in 1 2 3; do for v in libhj_vPREV.so libhj_mi355x.so; do
  HJ_LIB=$D/$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also WENO5,513,C3,C5 --repeats 15 --steps 20 > $out/b.json 2> $out/b.err || tail -3 $out/b.err
  python3 - $out/b.json $v $rep >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-18s pass %s  201^3 %.4e (%.4f, %.2f us) | " % (sys.argv[2], sys.argv[3], d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3) + " | ".join("%s %.4e (%.4f)" % (k[:11], v["value"], v.get("roofline_frac") or 0) for k, v in d["also"].items() if isinstance(v, dict) and "value" in v))
PY
done; done
cat $out/ab.txt
