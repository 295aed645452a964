#!/bin/bash
# round 3, run 33: product build with the 4-D fp32 pair configuration (256,2,10) as the default for C5: A/B against the
# one-cell-per-lane kernel (HJ_PAIR=0), tuner on / off; the 4-D tests
out=gpurun_out/r03ag; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_AUTOTUNE_LOG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 20 --repeats 3 --also C5 >> $out/ab.txt 2> $out/last.err; grep -E "autotune" $out/last.err >> $out/ab.txt; }
run HJ_PAIR=0
run HJ_PAIR=1
run HJ_PAIR=1 HJ_AUTOTUNE=0
run HJ_PAIR=0 HJ_AUTOTUNE=0
python - <<'PY'
import json
for ln in open("gpurun_out/r03ag/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:220]); continue
    d = json.loads(ln)
    for k, v in (d.get("also") or {}).items(): print("      also", k, {x: v.get(x) for x in ("value", "ms_per_step", "roofline_frac", "kernel", "error")}, v.get("repeats"))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c5_129 or fp32_4d" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -5 $out/test.txt
