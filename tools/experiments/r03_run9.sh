#!/bin/bash
# round 3, run 9: (a) stage-fused pair kernel with every VMEM op issued right after the barrier and the stores deferred by one
# iteration (lib D) against the first version (product lib of this snapshot); (b) cost of the bit-faithful ENO selectors:
# lean build (HJ_ENO_EXACT=0) against the product, C3 4096^2 ENO3 and 201^3 ENO3 / ENO2
out=gpurun_out/r03i; mkdir -p $out; rm -f $out/*
L=$PWD/levelsetpy_amd/csrc
HJ_LIB=$L/libhj_vD.so timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "stage_fused or stage_fusion" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; grep -E "tiling|fused12" $out/last.err | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
for n in 513 401 201; do
  EXTRA="--n $n" run HJ_FUSE12=0
  EXTRA="--n $n" run HJ_FUSE12=1
  EXTRA="--n $n" run HJ_LIB=$L/libhj_vD.so HJ_FUSE12=1
done
for sch in ENO3 ENO2; do
  EXTRA="--n 201 --scheme $sch" run HJ_FUSE12=0
  EXTRA="--n 201 --scheme $sch" run HJ_LIB=$L/libhj_vLEAN.so
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03i/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f kernel %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"], d["roofline"]["kernel"]))
PY
echo "== C3 exact" ; timeout -k 10 200 python tools/bench_configs.py c3 2>&1 | grep C3
echo "== C3 lean" ; HJ_LIB=$L/libhj_vLEAN.so timeout -k 10 200 python tools/bench_configs.py c3 2>&1 | grep C3
