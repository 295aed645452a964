#!/bin/bash
# round 4, call 26: the library built with other instruction-scheduling strategies of the AMDGPU back end (-mllvm -amdgpu-sched-strategy=max-ilp /
# max-memory-clause; iterative-ilp crashes the compiler) against the default: parity subset, then the headline and the other workloads, two alternations
out=gpurun_out/r04_run26; mkdir -p $out; : > $out/ab.txt
D=$PWD/levelsetpy_amd/csrc
for v in libhj_vS_max-ilp.so libhj_vS_max-memory-clause.so; do
  HJ_LIB=$D/$v timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or oracle or bitwise" > $out/pytest_$v.log 2>&1; echo "pytest $v rc=$?"; tail -1 $out/pytest_$v.log
done
for rep in 1 2; do for v in libhj_mi355x.so libhj_vS_max-ilp.so libhj_vS_max-memory-clause.so; do
  echo "== $v pass $rep" >> $out/ab.txt
  HJ_LIB=$D/$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also WENO5,513,C3,C5 --repeats 15 --steps 20 > $out/b.json 2> $out/b.err || tail -3 $out/b.err >> $out/ab.txt
  python3 - $out/b.json >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   201^3 %.4e (%.4f) | " % (d["value"], d["roofline"]["frac"]) + " | ".join("%s %.4e (%.4f)" % (k[:12], v["value"], v.get("roofline_frac") or 0) for k, v in d["also"].items() if isinstance(v, dict) and "value" in v))
PY
done; done
cat $out/ab.txt
