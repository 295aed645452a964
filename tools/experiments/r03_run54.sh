#!/bin/bash
# round 3, run 54: halo-ring LDS stores without the per-slot predicate (surplus slots shadow slot 0: duplicate writes of the same value)
# against the predicated form (tune build libhj_vPRED.so, -DHJ_PRED_HALO): 201^3, 401^3, C3, C5; the tests that depend on tilings
out=gpurun_out/r03bb; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --no-live-traffic --steps 30 --repeats 5 --also C3,C5,401 >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err; exit 1; }; }
run HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vPRED.so
run HJ_X=0
run HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vPRED.so
run HJ_X=0
python - <<'PY'
import json
for ln in open("gpurun_out/r03bb/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:120]); continue
    d = json.loads(ln)
    print("   201^3 %.4e frac %.4f" % (d["value"], d["roofline"]["frac"]), "  ".join("%s %.4f" % (k.split()[0], v["roofline_frac"]) for k, v in d["also"].items()))
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
