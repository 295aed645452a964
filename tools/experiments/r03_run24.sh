#!/bin/bash
# round 3, run 24: long-row tiles at 513^3 / 401^3 (HBM-bound): 258- and 172-cell rows (2 KB / 1.4 KB contiguous per row)
# against the default 74-cell rows; needs more halo slots per thread (KH = 3, 4: libhj_vFR.so)
out=gpurun_out/r03x; mkdir -p $out; rm -rf $out/*
L=$PWD/levelsetpy_amd/csrc
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 513 401; do
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vFR.so
  for fr in 258 172 130 104; do
    EXTRA="--n $n" run HJ_LIB=$L/libhj_vFR.so HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=4 HJ_PAIR_OCC=2 HJ_FULL_ROWS=$fr
    EXTRA="--n $n" run HJ_LIB=$L/libhj_vFR.so HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=3 HJ_PAIR_OCC=2 HJ_FULL_ROWS=$fr
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03x/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
