#!/bin/bash
# round 3, run 12: whole GPU suite on the NumPy-order ENO build + what the ENO paths cost now
out=gpurun_out/r03l; mkdir -p $out; rm -f $out/*
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/test_gpu.txt 2>&1; echo "rc=$?" >> $out/test_gpu.txt; tail -4 $out/test_gpu.txt
cat gpurun_out/eno_bitwise.txt | awk '{s+=($2!="differing=0")} END {print "comparisons with differing cells:", s}'
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err; }
for sch in ENO3 ENO2; do EXTRA="--n 201 --scheme $sch" run HJ_FUSE12=0; done
python - <<'PY'
import json
for ln in open("gpurun_out/r03l/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
timeout -k 10 200 python tools/bench_configs.py c3 2>&1 | grep C3
