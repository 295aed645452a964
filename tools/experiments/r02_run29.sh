#!/bin/bash
# cache-policy variants of the pair kernel: nt on the y0 loads (A), on the output stores (B), on both (C)
out=gpurun_out/r02ac; mkdir -p $out; rm -f $out/*
L=$PWD/levelsetpy_amd/csrc
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for n in 201 401 513; do
  EXTRA="--n $n" run HJ_PAIR=1
  for v in A B C; do EXTRA="--n $n" run HJ_LIB=$L/libhj_v$v.so; done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02ac/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:210]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
