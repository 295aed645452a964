#!/bin/bash
# round 3, run 39: intended WENO5 with max(D1^2) of a stage's output reduced inside the producing launch (+ seam kernel) against
# the two-launch pre-pass in front of every stage (HJ_EPS_FUSE=0): bitwise test, then timing at 51^3 ... 401^3
out=gpurun_out/r03am; mkdir -p $out; rm -rf $out/*
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "epsilon_reduced or WENO5 or weno" > $out/test.txt 2>&1; rc=$?; echo "rc=$rc" >> $out/test.txt; tail -15 $out/test.txt
[ $rc -eq 0 ] || exit 1
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --no-live-traffic --steps 30 --repeats 5 --scheme WENO5 --no-also $EXTRA >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err; exit 1; }; }
for n in 51 101 201 401; do
  for f in 0 1; do EXTRA="--n $n" run HJ_EPS_FUSE=$f; done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03am/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  us/step %.2f spread %.3f  %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3, d["repeats"]["spread"], d["roofline"]["kernel"][:40]))
PY
