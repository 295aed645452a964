#!/bin/bash
out=gpurun_out/r02d; mkdir -p $out; rm -f $out/f12.txt
run() { echo "== $*" >> $out/f12.txt; env "$@" HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 3 $EXTRA >> $out/f12.txt 2>> $out/f12.err; }
for n in 201 301 401 513; do
  EXTRA="--n $n" run HJ_FUSE12=0
  EXTRA="--n $n" run HJ_FUSE12=1
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_R=2
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_R=4
done
EXTRA="--also C3" ; echo "== C3 fuse=0" >> $out/f12.txt; HJ_FUSE12=0 python bench.py --no-cpu-baseline --steps 20 --repeats 3 --also C3 >> $out/f12.txt 2>> $out/f12.err
echo "== C3 fuse=1" >> $out/f12.txt; HJ_FUSE12=1 HJ_DEBUG=1 python bench.py --no-cpu-baseline --steps 20 --repeats 3 --also C3 >> $out/f12.txt 2>> $out/f12.err
python - <<'PY'
import json
for ln in open("gpurun_out/r02d/f12.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %-28s %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["metric"][-22:], d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
grep "fused12" $out/f12.err | sort | uniq -c
