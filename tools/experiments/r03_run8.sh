#!/bin/bash
# round 3, run 8: halo slots of the contiguous axis dealt layer-fastest (row-contiguous loads) against the old
# layer-major deal, unfused pair kernel and stage-fused pair kernel, same box
out=gpurun_out/r03h; mkdir -p $out; rm -f $out/*
L=$PWD/levelsetpy_amd/csrc
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "stage_fused or stage_fusion or pair_kernel or plain_stage" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 201 401 513; do
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vOLD.so
  EXTRA="--n $n" run HJ_FUSE12=0
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vOLD.so HJ_FUSE12=1
  EXTRA="--n $n" run HJ_FUSE12=1
done
EXTRA="--n 201 --scheme ENO3" run HJ_LIB=$L/libhj_vOLD.so
EXTRA="--n 201 --scheme ENO3" run HJ_FUSE12=0
EXTRA="--n 201 --scheme WENO5" run HJ_LIB=$L/libhj_vOLD.so
EXTRA="--n 201 --scheme WENO5" run HJ_FUSE12=0
python - <<'PY'
import json
for ln in open("gpurun_out/r03h/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"]))
PY
