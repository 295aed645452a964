#!/bin/bash
out=gpurun_out/r02l; mkdir -p $out; rm -f $out/*.txt
for n in 51 101; do
  for cfg in "" "HJ_NT=256 HJ_R=2" "HJ_MIN_CHUNK=2" "HJ_MIN_CHUNK=1" "HJ_NT=256 HJ_R=2 HJ_MIN_CHUNK=2" "HJ_NT=256 HJ_R=1 HJ_KH=2 HJ_OCC=6 HJ_MIN_CHUNK=2"; do
    echo "== n=$n $cfg" >> $out/small.txt
    env $cfg HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --n $n --steps 100 --repeats 3 >> $out/small.txt 2>> $out/small.err
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02l/small.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"]))
PY
grep "\[hj\]" $out/small.err
