#!/bin/bash
# pair-kernel tilings at 201/401/513 and a tile-shape / chunk sweep at 513 and 401
out=gpurun_out/r02u; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 20 --repeats 3 $EXTRA >> $out/ab.txt 2> $out/last.err; grep "pair tiling" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 201 401 513; do EXTRA="--n $n" run HJ_PAIR=1; done
for n in 513 401; do
  for fr in 128 172 256 514; do EXTRA="--n $n" run HJ_PAIR=1 HJ_FULL_ROWS=$fr; done
  for tb in 512 1024 2048 4096; do EXTRA="--n $n" run HJ_PAIR=1 HJ_TARGET_BLOCKS=$tb; done
  EXTRA="--n $n" run HJ_PAIR=1 HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=3
  EXTRA="--n $n" run HJ_PAIR=1 HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=1
  EXTRA="--n $n" run HJ_PAIR=1 HJ_PAIR_OCC=1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02u/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
