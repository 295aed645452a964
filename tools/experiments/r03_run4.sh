#!/bin/bash
# round 3, run 4: stage-fused pair kernel with more waves per SIMD: 768 threads x 1 pair (3 waves/SIMD), 1024 x 1 (4)
out=gpurun_out/r03d; mkdir -p $out; rm -f $out/*
L=$PWD/levelsetpy_amd/csrc
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 513 401 201; do
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vC.so HJ_FUSE12=1
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vC.so HJ_FUSE12=1 HJ_F12_NT=768 HJ_F12_R=1 HJ_F12_KH=2
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vC.so HJ_FUSE12=1 HJ_F12_NT=768 HJ_F12_R=1 HJ_F12_KH=1
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vC.so HJ_FUSE12=1 HJ_F12_NT=1024 HJ_F12_R=1 HJ_F12_KH=1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03d/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"]))
PY
