#!/bin/bash
# round 4, call 17: what would sharing the warm-up planes of a chunk with its neighbour buy?  Ablation (wrong results): the first 3 / 6 planes
# of the register queue's fill re-read plane p_begin (cache hits) instead of their own planes -- an upper bound for "paired chunks marching in
# opposite directions" 
out=gpurun_out/r04_run17; mkdir -p gpurun_out/r04_run17
D=$PWD/levelsetpy_amd/csrc
for rep in 1 2; do for n in 201 513; do for v in AP0 AP3 AP6; do
  echo "== n=$n $v pass $rep" >> $out/ab.txt
  HJ_LIB=$D/libhj_v$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic --no-also --repeats 25 --steps 30 --n $n >> $out/ab.txt 2> $out/last.err || tail -2 $out/last.err >> $out/ab.txt
done; done; done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r04_run17/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-28s %.4e  frac %.4f  us/launch %.2f  iqr %.4f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3, d["repeats"]["iqr_over_median"]))
    else: print("   ", ln.strip()[:150])
PY
