#!/bin/bash
# round 4, call 41: intended WENO5 with the first-difference parts of the smoothness measures formed from the second differences (15 fp64 operations
# per cell less): whole GPU suite (oracle / golden tolerances), then 201^3 intended WENO5 against the previous build, three alternations
out=gpurun_out/r04_run41; mkdir -p $out; : > $out/ab.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
D=$PWD/levelsetpy_amd/csrc
for rep in 1 2 3; do for v in libhj_vPREV.so libhj_mi355x.so; do
  HJ_LIB=$D/$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also WENO5 --repeats 11 --steps 20 > $out/b.json 2> $out/b.err || tail -3 $out/b.err
  python3 - $out/b.json $v $rep >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
v = d["also"]["201^3 WENO5"]
print("%-20s pass %s  201^3 intended WENO5 %.4e  frac %.4f  valu %.3f" % (sys.argv[2], sys.argv[3], v["value"], v["roofline_frac"], v["roofline_valu"]["frac"]))
PY
done; done
cat $out/ab.txt
