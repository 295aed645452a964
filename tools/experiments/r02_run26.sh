#!/bin/bash
out=gpurun_out/r02z; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for n in 51 101 201; do
  EXTRA="--n $n" run HIP_FORCE_DEV_KERNARG=0
  EXTRA="--n $n" run HIP_FORCE_DEV_KERNARG=1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02z/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
