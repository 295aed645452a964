#!/bin/bash
# small grids: pair kernel forced vs scalar kernel, and chunk-length knobs
out=gpurun_out/r02ab; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 40 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 51 101 129 151; do
  EXTRA="--n $n" run HJ_PAIR=1
  EXTRA="--n $n" run HJ_PAIR=2
  EXTRA="--n $n" run HJ_PAIR=2 HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_PAIR_KH=2 HJ_PAIR_OCC=2
  EXTRA="--n $n" run HJ_PAIR=2 HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=1 HJ_PAIR_OCC=2
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02ab/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:210]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
