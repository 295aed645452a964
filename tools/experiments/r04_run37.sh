#!/bin/bash
# round 4, call 37: the fp64 double-integrator object alone built with -mllvm -amdgpu-sched-strategy=max-ilp (product) against the default
# scheduling (libhj_vPREV.so): C3, three alternations; then the 2-D tests
out=gpurun_out/r04_run37; mkdir -p $out; : > $out/ab.txt
D=$PWD/levelsetpy_amd/csrc
for rep in 1 2 3; do for v in libhj_vPREV.so libhj_mi355x.so; do
  HJ_LIB=$D/$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also C3 --repeats 11 --steps 20 > $out/b.json 2> $out/b.err || tail -3 $out/b.err
  python3 - $out/b.json $v $rep >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
v = d["also"]["C3"]
print("%-20s pass %s  C3 %.4e  frac %.4f   (201^3 %.4e)" % (sys.argv[2], sys.argv[3], v["value"], v["roofline_frac"], d["value"]))
PY
done; done
cat $out/ab.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "integrator or 2d or c3 or 4096 or eighty" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $out/pytest.log
