#!/bin/bash
out=gpurun_out/r02n; mkdir -p $out; rm -f $out/*
for n in 201; do
  for tb in 0 504 1008 1512 2268 3024 4536; do
    echo "== n=$n HJ_TARGET_BLOCKS=$tb" >> $out/tb.txt
    HJ_TARGET_BLOCKS=$tb HJ_MIN_CHUNK=2 HJ_DEBUG=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --n $n --steps 60 --repeats 3 >> $out/tb.txt 2>> $out/tb.err
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02n/tb.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"]))
PY
grep "\[hj\]" $out/tb.err
