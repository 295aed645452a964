#!/bin/bash
# round 3, run 28: the tile-shape tuner on the one-cell-per-lane path (C5 129^4 fp32, ENO3 / intended WENO5 at 401^3, 513^3)
out=gpurun_out/r03ab; mkdir -p $out; rm -rf $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_AUTOTUNE_LOG=1 timeout -k 10 400 python bench.py --no-cpu-baseline --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "autotune" $out/last.err >> $out/ab.txt; }
EXTRA="--n 201 --also C5" run HJ_AUTOTUNE=0
EXTRA="--n 201 --also C5" run HJ_AUTOTUNE=1
for sch in ENO3 WENO5 ENO2; do
  for n in 401 513; do
    EXTRA="--n $n --scheme $sch --no-also" run HJ_AUTOTUNE=0
    EXTRA="--n $n --scheme $sch --no-also" run HJ_AUTOTUNE=1
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03ab/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
    for k, v in (d.get("also") or {}).items(): print("      also", k, json.dumps(v)[:300])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c4 or 513 or c5 or pair_kernel or plain" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
