#!/bin/bash
# round 3, run 27: launch-time autotuned tile shape against the static choice (HJ_AUTOTUNE=0), sizes 301^3 ... 601^3
out=gpurun_out/r03aa; mkdir -p $out; rm -rf $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_AUTOTUNE_LOG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "autotune" $out/last.err >> $out/ab.txt; }
for n in 351 401 451 513 551 601; do
  EXTRA="--n $n" run HJ_AUTOTUNE=0
  EXTRA="--n $n" run HJ_AUTOTUNE=1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03aa/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
# C3 (4096^2 = 16.8 M cells: below the threshold) and the tests that depend on tilings
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c4 or 513 or c5 or pair_kernel or plain" > gpurun_out/r03aa/test.txt 2>&1; echo "rc=$?" >> gpurun_out/r03aa/test.txt; tail -3 gpurun_out/r03aa/test.txt
