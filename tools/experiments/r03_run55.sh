#!/bin/bash
# round 3, run 55: pendulum drift with ONE division (product) against two (tune build -DHJ_PENDULUM_TWO_DIV): C5; the 4-D tests
out=gpurun_out/r03bc; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ab.txt; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 30 --repeats 5 --also C5 >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err; exit 1; }; }
run HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vDIV2.so
run HJ_X=0
run HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vDIV2.so
run HJ_X=0
python - <<'PY'
import json
for ln in open("gpurun_out/r03bc/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:120]); continue
    d = json.loads(ln)
    print("   ", "  ".join("%s %.4f (%.4e)" % (k.split()[0], v["roofline_frac"], v["value"]) for k, v in d["also"].items()))
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "4d or pendulum or c5" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
