#!/bin/bash
# how far ahead the halo ring has to be parked: HJ_PAIR_AH = 3 (aligned with the neighbours' own loads), 2, 1
out=gpurun_out/r02aq; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 40 --repeats 7 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for rep in 1 2; do for n in 201 401 513; do
  for ah in 3 2 1; do EXTRA="--n $n" run HJ_PAIR_AH=$ah; done
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/r02aq/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[-40:]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
