#!/bin/bash
# round 3, run 1: (a) this box's baseline, (b) one wave per SIMD pair-kernel shapes (256 threads x 4 / 3 pairs,
# 256 VGPRs + AGPR spill space) against the default (512,2,2,2), (c) the round-2 stage-fused kernel A/B
out=gpurun_out/r03a; mkdir -p $out; rm -f $out/*
L=$PWD/levelsetpy_amd/csrc
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 201 401 513; do
  EXTRA="--n $n" run HJ_PAIR=1
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vA.so HJ_PAIR_NT=256 HJ_PAIR_R=4 HJ_PAIR_KH=3 HJ_PAIR_OCC=1 HJ_PAIR_RING=0
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vA.so HJ_PAIR_NT=256 HJ_PAIR_R=4 HJ_PAIR_KH=3 HJ_PAIR_OCC=1 HJ_PAIR_RING=1
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vA.so HJ_PAIR_NT=256 HJ_PAIR_R=3 HJ_PAIR_KH=3 HJ_PAIR_OCC=1 HJ_PAIR_RING=0
  EXTRA="--n $n" run HJ_FUSE12=1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03a/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"]))
PY
