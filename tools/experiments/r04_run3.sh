#!/bin/bash
# round 4, call 3: prologue variants of the pair kernel (tune builds PBxy: x = HJ_PROLOGUE_V2, y = HJ_EARLY_ARGS), 201^3 and 513^3,
# oracle parity of the variant that changes most, and its per-workgroup stamps
out=gpurun_out/r04_run3; mkdir -p $out; rm -f $out/*
D=$PWD/levelsetpy_amd/csrc
HJ_LIB=$D/libhj_vPB11.so timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "c2_201 or c1_51" > $out/pytest_pb11.log 2>&1; echo "pytest PB11 rc=$?" >> $out/ab.txt; tail -2 $out/pytest_pb11.log >> $out/ab.txt
run() { tag=$1; n=$2; shift; shift; echo "== n=$n $tag" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic --no-also --repeats 15 --steps 30 --n $n >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err >> $out/ab.txt; }; }
for rep in 1 2; do for n in 201 513; do for v in PB00 PB10 PB01 PB11; do
  run "$v (pass $rep)" $n HJ_LIB=$D/libhj_v$v.so
done; done; done
for v in PB00 PB11; do
HJ_LIB=$D/libhj_v$v.so HJ_TIMING_DUMP=$out/t201.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic --no-also --n 201 --steps 3 --warmup 3 --repeats 1 > /dev/null 2> $out/t201.err
python tools/pair_timing.py $out/t201.txt > $out/pair_timing_201_$v.txt 2>&1; rm -f $out/t201.txt
done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r04_run3/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-40s %.4e  frac %.4f  us/launch %.2f  iqr %.4f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3, d["repeats"]["iqr_over_median"]))
    else: print("     ", ln.strip()[:200])
PY
grep -h "prolog\|prologue stamps\|end  :" $out/pair_timing_201_PB00.txt | head -6; echo; grep -h "prolog\|prologue stamps\|end  :" $out/pair_timing_201_PB11.txt | head -6
