#!/bin/bash
# round 3, run 7: what bounds the stage-fused pair kernel?  Ablation builds (WRONG results, timing only):
# 1 no barrier, 2 no LDS stencil reads, 4 no global loads in the loop, 3 = 1+2, 7 = all three
out=gpurun_out/r03g; mkdir -p $out; rm -f $out/*
L=$PWD/levelsetpy_amd/csrc
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "fused_term or opt_traj" > $out/test_terms.txt 2>&1; echo "rc=$?" >> $out/test_terms.txt; tail -5 $out/test_terms.txt
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 3 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 513; do
  EXTRA="--n $n" run HJ_FUSE12=0
  EXTRA="--n $n" run HJ_FUSE12=1
  for v in 1 2 4 3 7; do EXTRA="--n $n" run HJ_LIB=$L/libhj_vAB$v.so HJ_FUSE12=1; done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03g/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"]))
PY
