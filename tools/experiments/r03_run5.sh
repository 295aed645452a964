#!/bin/bash
# round 3, run 5: stage-fused pair kernel: rows of 64 cells (a half wave = one LDS row: conflict-free ds_read_b128),
# and a start delay for waves 4-7 (s_sleep N after the barrier) against the lock-step of LDS bursts and arithmetic
out=gpurun_out/r03e; mkdir -p $out; rm -f $out/*
L=$PWD/levelsetpy_amd/csrc
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 513 401; do
  EXTRA="--n $n" run HJ_FUSE12=1
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_E2=64
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_E2=32
  for v in 2 4 8; do
    EXTRA="--n $n" run HJ_LIB=$L/libhj_vS$v.so HJ_FUSE12=1
    EXTRA="--n $n" run HJ_LIB=$L/libhj_vS$v.so HJ_FUSE12=1 HJ_F12_E2=64
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03e/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"]))
PY
