#!/bin/bash
# round 3, run 62: ablation of the headline launch (201^3 and 513^3, pair kernel (512,2,2) + ring; tune builds -DHJ_ABLATE=bits, results wrong by
# construction): 1 no Hamiltonian / dissipation arithmetic, 2 no stencil LDS reads, 8 no halo loads, NS no barrier
out=gpurun_out/r03bj; mkdir -p $out; rm -rf $out/*
for n in 201 513; do for v in HB0 HB1 HB2 HB8 HB3 HB11 HBNS; do
  echo "== n=$n $v" >> $out/ab.txt
  HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_v$v.so HJ_AUTOTUNE=0 HJ_FULL_ROWS=$([ $n = 513 ] && echo 130 || echo 0) timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-also --n $n --steps 30 --repeats 3 >> $out/ab.txt 2> $out/last.err || { tail -2 $out/last.err >> $out/ab.txt; }
done; done
python - <<'PY'
import json
n = None
for ln in open("gpurun_out/r03bj/ab.txt"):
    if ln.startswith("=="): n = ln.strip(); continue
    if ln.startswith("{"):
        d = json.loads(ln); print("%-16s %.4e  frac %.4f  us/launch %.2f" % (n, d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3 / 3))
    else: print(n, ln.strip()[:120])
PY
