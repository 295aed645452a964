#!/bin/bash
# round 3, run 18: rewritten direct kernel (unconditional stencil loads) against the tiled kernels on small grids
out=gpurun_out/r03r; mkdir -p $out; rm -rf $out/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "direct or kernels or tiled or fuzz or c4 or c5 or 513 or 129" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 100 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; }
for n in 31 41 51 65 81 101 129 151 201; do
  EXTRA="--n $n" run HJ_FORCE_DIRECT=0
  EXTRA="--n $n" run HJ_FORCE_DIRECT=1
done
for sch in ENO3 WENO5; do for n in 51 101; do
  EXTRA="--n $n --scheme $sch" run HJ_FORCE_DIRECT=0
  EXTRA="--n $n --scheme $sch" run HJ_FORCE_DIRECT=1
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/r03r/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  us/step %.2f spread %.3f %s" % (d["value"], d["roofline"]["frac"], 1e3*d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"][:28]))
PY
