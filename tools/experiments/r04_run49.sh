#!/bin/bash
# round 4, call 49: ablation (wrong results): the pair kernel WITHOUT its per-plane barrier (tune build -DHJ_ABLATE_NOSYNC) against the same tune build with it --
# an upper bound for what halving the barriers (two planes staged per barrier) could buy at 201^3 / 513^3
out=gpurun_out/r04_run49; mkdir -p $out; : > $out/ab.txt
D=$PWD/levelsetpy_amd/csrc
for rep in 1 2; do for n in 201 513; do for v in libhj_vBASE.so libhj_vNOSYNC.so; do
  HJ_LIB=$D/$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --no-also --repeats 15 --steps 20 --n $n > $out/b.json 2> $out/b.err || { echo "$v n=$n failed: $(tail -1 $out/b.err | cut -c1-150)" >> $out/ab.txt; continue; }
  python3 - $out/b.json $v $n >> $out/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-18s n=%s  %.4e  frac %.4f  us/launch %.2f" % (sys.argv[2], sys.argv[3], d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3))
PY
done; done; done
cat $out/ab.txt
