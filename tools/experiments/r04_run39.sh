#!/bin/bash
# round 4, call 39: C5 chunk count (HJ_TARGET_BLOCKS = chunks x 2288 tiles): 1, 2 (the planner's choice), 3, 4 chunks per tile column
out=gpurun_out/r04_run39; mkdir -p $out; : > $out/ab.txt
for rep in 1 2; do for tb in 0 2288 6864 9152; do
  HJ_TARGET_BLOCKS=$tb HJ_DEBUG=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also C5 --repeats 7 --steps 20 > $out/b.json 2> $out/b.err || tail -3 $out/b.err
  python3 - $out/b.json $tb $rep >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
v = d["also"]["C5"]
print("HJ_TARGET_BLOCKS=%-5s pass %s  C5 %.4e (%.4f)  %.3f ms/launch" % (sys.argv[2], sys.argv[3], v["value"], v["roofline_frac"], v["ms_per_step"] / 3))
PY
  grep -h "pair tiling" $out/b.err | grep "E=(5,6,34)" | tail -1 | cut -c1-200 >> $out/ab.txt
done; done
cat $out/ab.txt
