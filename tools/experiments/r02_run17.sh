#!/bin/bash
out=gpurun_out/r02q; mkdir -p $out; rm -f $out/*
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "pair_kernel" > $out/test.log 2>&1; echo "pytest rc=$?"; tail -12 $out/test.log
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --steps 40 --repeats 5 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for n in 101 201 301 513; do
  EXTRA="--n $n" run HJ_PAIR=0
  EXTRA="--n $n" run HJ_PAIR=1
done
EXTRA="--n 201" run HJ_PAIR=1 HJ_PAIR_OCC=3
EXTRA="--n 201" run HJ_PAIR=1 HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=1
EXTRA="--n 201" run HJ_PAIR=1 HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=2
EXTRA="--n 513" run HJ_PAIR=1 HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_PAIR_KH=2
python - <<'PY'
import json
for ln in open("gpurun_out/r02q/ab.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
grep "\[hj\]" $out/ab.err | sort | uniq -c
