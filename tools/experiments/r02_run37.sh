#!/bin/bash
# chunk count at 513^3 / 401^3 / 301^3 with the halo ring on
out=gpurun_out/r02ak; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 20 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 | cut -c1-200 >> $out/ab.txt; }
for tb in 0 400 700 950 1500 2100; do EXTRA="--n 513" run HJ_TARGET_BLOCKS=$tb; done
for tb in 0 504 760; do EXTRA="--n 401" run HJ_TARGET_BLOCKS=$tb; done
for tb in 0 480 720; do EXTRA="--n 301" run HJ_TARGET_BLOCKS=$tb; done
python - <<'PY'
import json
for ln in open("gpurun_out/r02ak/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
