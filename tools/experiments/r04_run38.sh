#!/bin/bash
# round 4, call 38: paired chunk directions on C5 (129^4 fp32: two chunks per tile column) and C3 (2-D), HJ_PAIR_DIRS=1 against 0, two alternations
out=gpurun_out/r04_run38; mkdir -p $out; : > $out/ab.txt
for rep in 1 2; do for pd in 0 1; do
  HJ_PAIR_DIRS=$pd timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also C5,C3 --repeats 9 --steps 20 > $out/b.json 2> $out/b.err || tail -3 $out/b.err
  python3 - $out/b.json $pd $rep >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("HJ_PAIR_DIRS=%s pass %s  " % (sys.argv[2], sys.argv[3]) + " | ".join("%s %.4e (%.4f)" % (k, v["value"], v["roofline_frac"]) for k, v in d["also"].items() if isinstance(v, dict) and "value" in v))
PY
done; done
cat $out/ab.txt
