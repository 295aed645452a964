#!/bin/bash
out=gpurun_out/r02b; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
tail -3 $out/test.log
for np in 1 0; do
  for n in 201 401 513; do
    echo "== HJ_NO_PLAIN=$np n=$n" >> $out/plain_ab.txt
    HJ_NO_PLAIN=$np timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --n $n --steps 40 --repeats 5 >> $out/plain_ab.txt 2>> $out/plain_ab.err
  done
  echo "== HJ_NO_PLAIN=$np also" >> $out/plain_ab.txt
  HJ_NO_PLAIN=$np timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --also WENO5,ENO3,ENO2,C3,C5 >> $out/plain_ab.txt 2>> $out/plain_ab.err
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02b/plain_ab.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %-28s %.4e  frac %.3f  spread %.3f" % (d["metric"][-22:], d["value"], d["roofline"]["frac"], d["repeats"]["spread"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
