#!/bin/bash
# pair-kernel workgroup-shape candidates (variant library libhj_vG.so)
out=gpurun_out/r02v; mkdir -p $out; rm -f $out/*
export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vG.so
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 20 --repeats 3 $EXTRA >> $out/ab.txt 2> $out/last.err; grep "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -1 >> $out/ab.txt; }
for n in 201 401 513; do
  EXTRA="--n $n" run HJ_PAIR=1
  EXTRA="--n $n" run HJ_PAIR=2 HJ_PAIR_NT=1024 HJ_PAIR_R=1 HJ_PAIR_KH=1 HJ_PAIR_OCC=4
  EXTRA="--n $n" run HJ_PAIR=2 HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_OCC=2
  EXTRA="--n $n" run HJ_PAIR=2 HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=1 HJ_PAIR_OCC=3
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02v/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
