#!/bin/bash
out=gpurun_out/r02ap; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 40 --repeats 7 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for rep in 1 2; do for n in 201 301 513; do
  EXTRA="--n $n" run HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_mi355x.so
  EXTRA="--n $n" run HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_new.so
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/r02ap/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[-40:]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
