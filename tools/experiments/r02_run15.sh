#!/bin/bash
out=gpurun_out/r02o; mkdir -p $out; rm -f $out/*
for n in 151 201 251; do
  for cfg in "" "HJ_NT=256 HJ_R=2 HJ_OCC=2"; do
    echo "== n=$n $cfg" >> $out/ab.txt
    env $cfg HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --n $n --steps 60 --repeats 5 >> $out/ab.txt 2>> $out/ab.err
  done
done
echo "== also" >> $out/ab.txt
python bench.py --no-cpu-baseline --steps 20 --also ENO2,C3 >> $out/ab.txt 2>> $out/ab.err
python - <<'PY'
import json
for ln in open("gpurun_out/r02o/ab.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
grep "\[hj\]" $out/ab.err | sort | uniq -c
