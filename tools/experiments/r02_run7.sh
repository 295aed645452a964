#!/bin/bash
out=gpurun_out/r02g; mkdir -p $out; rm -f $out/ab.txt $out/ab.err
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 3 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for n in 201 513; do
  EXTRA="--n $n" run HJ_FUSE12=0
  EXTRA="--n $n" run HJ_FUSE12=0 HJ_FULL_ROWS=32
  EXTRA="--n $n" run HJ_FUSE12=0 HJ_FULL_ROWS=64
  EXTRA="--n $n" run HJ_FUSE12=0 HJ_FULL_ROWS=64 HJ_LDS_PAD=1
done
EXTRA="--n 513" run HJ_FUSE12=1 HJ_F12_E2=32
EXTRA="--n 513" run HJ_FUSE12=1 HJ_F12_E2=64
EXTRA="--n 513" run HJ_FUSE12=1 HJ_F12_E2=64 HJ_F12_R=4
python - <<'PY'
import json
for ln in open("gpurun_out/r02g/ab.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %-28s %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["metric"][-22:], d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
grep "\[hj\]" $out/ab.err
