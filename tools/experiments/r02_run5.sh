#!/bin/bash
out=gpurun_out/r02e; mkdir -p $out; rm -f $out/f12.txt $out/f12.err
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "stage" > $out/test.log 2>&1; echo "pytest rc=$?"; tail -3 $out/test.log
run() { echo "== $*" >> $out/f12.txt; env "$@" HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 3 $EXTRA >> $out/f12.txt 2>> $out/f12.err; }
for n in 301 513; do
  EXTRA="--n $n" run HJ_FUSE12=0
  EXTRA="--n $n" run HJ_FUSE12=1
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_R=2
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_R=2 HJ_F12_NT=768
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_R=2 HJ_F12_NT=1024 HJ_F12_KH=1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02e/f12.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %-28s %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["metric"][-22:], d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
grep "fused12" $out/f12.err | sort | uniq -c
