#!/bin/bash
# round 3, run 30: C5 (129^4 fp32) through the two-cells-per-lane kernel (never instantiated in 4-D before): configurations
# (threads, pairs per thread, halo slots per thread), with and without the LDS halo ring; tune build libhj_vC5P.so
out=gpurun_out/r03ad; mkdir -p $out; rm -rf $out/*
export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_vC5P.so
run() { echo "== $*" >> $out/ab.txt; env "$@" HJ_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --n 101 --steps 8 --repeats 2 --also C5 >> $out/ab.txt 2> $out/last.err; grep -E "pair|tiling|E=\(" $out/last.err | sort | uniq -c | sort -rn | head -4 >> $out/ab.txt; tail -2 $out/last.err >> $out/ab.txt; }
run HJ_PAIR=1
run HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=5
run HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=5 HJ_PAIR_RING=1
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=7
run HJ_PAIR_NT=512 HJ_PAIR_R=2 HJ_PAIR_KH=7 HJ_PAIR_RING=1
run HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=6
run HJ_PAIR_NT=1024 HJ_PAIR_R=1 HJ_PAIR_KH=4
python - <<'PY'
import json
for ln in open("gpurun_out/r03ad/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:220]); continue
    d = json.loads(ln)
    for k, v in (d.get("also") or {}).items(): print("      also", k, {x: v.get(x) for x in ("value", "ms_per_step", "roofline_frac", "kernel", "error")})
PY
HJ_PAIR_NT=512 HJ_PAIR_R=1 HJ_PAIR_KH=5 timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c5_129 or (fp32_4d_tiled and ASSHIPPED)" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -5 $out/test.txt
