#!/bin/bash
# round 3, run 37: small grids after the bound reduction left the launches: chunk length / workgroup count of the tiled kernel at 51^3, 101^3
out=gpurun_out/r03ak; mkdir -p $out; rm -rf $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 400 python bench.py --no-cpu-baseline --no-live-traffic --steps 100 --repeats 5 --no-also $EXTRA >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err; exit 1; }; grep -E "tiling" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 51 101; do
  EXTRA="--n $n" run HJ_X=0
  for mc in 1 2 3 4 6 8; do EXTRA="--n $n" run HJ_MIN_CHUNK=$mc; done
  for wc in 2 4 10 16; do EXTRA="--n $n" run HJ_WARMUP_COST=$wc; done
  EXTRA="--n $n" run HJ_NT=256 HJ_R=2 HJ_KH=2 HJ_OCC=2 HJ_PD=2
  EXTRA="--n $n" run HJ_NT=256 HJ_R=2 HJ_KH=2 HJ_OCC=2 HJ_PD=2 HJ_MIN_CHUNK=2
  EXTRA="--n $n" run HJ_PAIR=2
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03ak/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  us/step %.2f spread %.3f  %s" % (d["value"], d["ms_per_step"] * 1e3, d["repeats"]["spread"], d["roofline"]["kernel"][:24]))
PY
