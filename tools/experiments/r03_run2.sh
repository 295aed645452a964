#!/bin/bash
# round 3, run 2: stage-fused pair kernel (hj_fused12v.h): bitwise tests, then A/B against the unfused default
# and the round-2 stage-fused kernel; exact ENO selectors: golden tests with frac = 0
out=gpurun_out/r03b; mkdir -p $out; rm -f $out/*
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "stage_fused or stage_fusion" > $out/test12.txt 2>&1; echo "rc=$?" >> $out/test12.txt; tail -5 $out/test12.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "golden or generic_path or eno" > $out/test_eno.txt 2>&1; echo "rc=$?" >> $out/test_eno.txt; tail -15 $out/test_eno.txt
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 201 401 513; do
  EXTRA="--n $n" run HJ_FUSE12=0
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_PAIR=0
  EXTRA="--n $n" run HJ_FUSE12=1
  EXTRA="--n $n" run HJ_FUSE12=1 HJ_F12_NT=256 HJ_F12_R=4 HJ_F12_KH=4
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03b/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"]))
PY
