#!/bin/bash
out=gpurun_out/r02r; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --steps 60 --repeats 7 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for rep in 1 2; do
for n in 151 201 251; do
  EXTRA="--n $n" run HJ_PAIR=0
  EXTRA="--n $n" run HJ_PAIR=1
  EXTRA="--n $n" run HJ_PAIR=1 HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=2
done
done
for p in 0 1; do
echo "== also HJ_PAIR=$p" >> $out/ab.txt
HJ_PAIR=$p python bench.py --no-cpu-baseline --steps 20 --also WENO5,ENO3,ENO2,C3,401 >> $out/ab.txt 2>> $out/ab.err
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02r/ab.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
