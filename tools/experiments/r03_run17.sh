#!/bin/bash
# round 3, run 17: unfused pair kernel with a conflict-free LDS row pitch (E2 + 32 cells: pitch in pairs congruent to the row
# length mod 16) against the default pitch (E2 + 8)
out=gpurun_out/r03q; mkdir -p $out; rm -rf $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 201 301 401 513; do
  EXTRA="--n $n" run HJ_LDS_PAD=0
  EXTRA="--n $n" run HJ_LDS_PAD=1
done
EXTRA="--n 201 --scheme ENO3" run HJ_LDS_PAD=0
EXTRA="--n 201 --scheme ENO3" run HJ_LDS_PAD=1
EXTRA="--n 201 --scheme WENO5" run HJ_LDS_PAD=0
EXTRA="--n 201 --scheme WENO5" run HJ_LDS_PAD=1
python - <<'PY'
import json
for ln in open("gpurun_out/r03q/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
