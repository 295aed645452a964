#!/bin/bash
# does the 201^3 number depend on how long the GPU has been busy (clock ramp)?
out=gpurun_out/r02ai; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
EXTRA="--steps 20 --warmup 5" run HJ_BENCH_SPINUP=300
EXTRA="--steps 20 --warmup 5" run HJ_BENCH_SPINUP=3000
EXTRA="--steps 20 --warmup 5" run HJ_BENCH_SPINUP=20000
EXTRA="--steps 200 --warmup 20" run HJ_BENCH_SPINUP=300
EXTRA="--steps 2000 --warmup 20 --repeats 3" run HJ_BENCH_SPINUP=300
python - <<'PY'
import json
for ln in open("gpurun_out/r02ai/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
