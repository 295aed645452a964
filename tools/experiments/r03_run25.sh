#!/bin/bash
# round 3, run 25: tile-shape sweep (row length forced, KH = 2/3/4) at 201^3, 301^3, 601^3 with libhj_vFR.so
out=gpurun_out/r03y; mkdir -p $out; rm -rf $out/*
L=$PWD/levelsetpy_amd/csrc
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 201 301 601; do
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vFR.so
  for fr in 52 68 102 134 202; do
    for kh in 2 3; do
      EXTRA="--n $n" run HJ_LIB=$L/libhj_vFR.so HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=$kh HJ_PAIR_OCC=2 HJ_FULL_ROWS=$fr
    done
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03y/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
