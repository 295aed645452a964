#!/bin/bash
out=gpurun_out/r02f; mkdir -p $out; rm -f $out/ab.txt $out/ab.err
run() { echo "== $*" >> $out/ab.txt; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --steps 40 --repeats 5 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for n in 101 201 401 513; do EXTRA="--n $n" run HJ_FUSE12=0; done
echo "== also" >> $out/ab.txt
HJ_FUSE12=0 python bench.py --no-cpu-baseline --steps 20 --also WENO5,ENO3,ENO2,C3,C5 >> $out/ab.txt 2>> $out/ab.err
python - <<'PY'
import json
for ln in open("gpurun_out/r02f/ab.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %-28s %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["metric"][-22:], d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
