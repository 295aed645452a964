#!/bin/bash
# two 256-thread workgroups per CU (2 pairs per thread) with the halo ring, against the default one 512-thread workgroup
out=gpurun_out/r02an; mkdir -p $out; rm -f $out/*
export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vU.so
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 | cut -c1-220 >> $out/ab.txt; }
for n in 201 401 513; do
  EXTRA="--n $n" run HJ_PAIR=1
  EXTRA="--n $n" run HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_OCC=2 HJ_PAIR_RING=1
  EXTRA="--n $n" run HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_OCC=2 HJ_PAIR_RING=0
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02an/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
