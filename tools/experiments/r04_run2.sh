#!/bin/bash
# round 4, call 2: (1) achievable streaming rates of this box (tools/ubench/bw2), (2) the pair kernel as TWO 256-thread x 2-pair
# workgroups per CU in 3-D against the default one 512-thread workgroup (tune build P256), (3) per-workgroup stamps of the default
# 201^3 launch.
out=gpurun_out/r04_run2; mkdir -p $out; rm -f $out/*
tools/ubench/bw2 > $out/bw2.txt 2>&1; tail -1 $out/bw2.txt
V=$PWD/levelsetpy_amd/csrc/libhj_vP256.so
run() { tag=$1; n=$2; shift; shift; echo "== n=$n $tag" >> $out/ab.txt; env HJ_DEBUG=1 "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic --no-also --repeats 9 --steps 30 --n $n >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err >> $out/ab.txt; }; grep -h "tiling" $out/last.err | head -2 >> $out/ab.txt; }
for n in 201 513; do
  run "default (product lib)" $n HJ_X=0
  run "tune lib, default config" $n HJ_LIB=$V
  run "256x2 pairs, KH 2, two WGs/CU, no ring" $n HJ_LIB=$V HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_OCC=2
  run "256x2 pairs, KH 2, two WGs/CU, ring" $n HJ_LIB=$V HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_OCC=2 HJ_PAIR_RING=1
  run "256x2 pairs, KH 3, two WGs/CU, no ring" $n HJ_LIB=$V HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=3 HJ_PAIR_OCC=2
  run "256x2 pairs, KH 3, two WGs/CU, ring" $n HJ_LIB=$V HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=3 HJ_PAIR_OCC=2 HJ_PAIR_RING=1
  run "256x2 pairs, KH 2, three WGs/CU (168 VGPRs)" $n HJ_LIB=$V HJ_PAIR_NT=256 HJ_PAIR_R=2 HJ_PAIR_KH=2 HJ_PAIR_OCC=3
done
HJ_TIMING_DUMP=$out/t201.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic --no-also --n 201 --steps 3 --warmup 3 --repeats 1 > /dev/null 2> $out/t201.err
python tools/pair_timing.py $out/t201.txt > $out/pair_timing_201.txt 2>&1; rm -f $out/t201.txt
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r04_run2/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-60s %.4e  frac %.4f  us/launch %.2f  iqr %.4f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3, d["repeats"]["iqr_over_median"]))
    else: print("     ", ln.strip()[:200])
PY
