#!/bin/bash
# 4-D fp32 (C5) with the pair kernel: correctness of the 4-D tests, then throughput per configuration
out=gpurun_out/r02ad; mkdir -p $out; rm -f $out/*
export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vQ.so
HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=5 HJ_PAIR_OCC=2 timeout -k 10 400 python -m pytest tests -m gpu -q -k "4d or c5 or C5 or fp32 or pendulum" > $out/t1.log 2>&1; tail -3 $out/t1.log
HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=7 HJ_PAIR_OCC=2 timeout -k 10 400 python -m pytest tests -m gpu -q -k "4d or c5 or C5 or fp32 or pendulum" > $out/t2.log 2>&1; tail -3 $out/t2.log
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --repeats 1 --also C5 >> $out/ab.txt 2> $out/last.err; grep "tiling" $out/last.err | grep "E=([0-9]*,[0-9]*,[1-9]" | sort | uniq -c | sort -rn | head -2 >> $out/ab.txt; }
run HJ_PAIR=1
run HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=5 HJ_PAIR_OCC=2
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=7 HJ_PAIR_OCC=2
run HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=5 HJ_PAIR_OCC=2
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=7 HJ_PAIR_OCC=2 HJ_LDS_LIMIT=150000
python - <<'PY'
import json
for ln in open("gpurun_out/r02ad/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:230]); continue
    d = json.loads(ln)
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
