#!/bin/bash
# round 3, run 26: 513^3 (and 451^3, 481^3, 551^3), product library, row length forced (KH = 2 of the default configuration)
out=gpurun_out/r03z; mkdir -p $out; rm -rf $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 513; do
  for fr in 0 74 86 104 116 130 148; do EXTRA="--n $n" run HJ_FULL_ROWS=$fr; done
done
for n in 451 481 551; do
  for fr in 0 104 130; do EXTRA="--n $n" run HJ_FULL_ROWS=$fr; done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03z/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
