#!/bin/bash
# A/B on one box: variants of the pair kernel's 16-byte store (A asm+nops+memory clobber, B no clobber, C no nops, D builtin)
out=gpurun_out/r02t; mkdir -p $out; rm -f $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
L=$PWD/levelsetpy_amd/csrc
for n in 201 401 513; do
  EXTRA="--n $n" run HJ_PAIR=0
  EXTRA="--n $n" run HJ_PAIR=1
  for v in D F; do EXTRA="--n $n" run HJ_PAIR=1 HJ_LIB=$L/libhj_v$v.so; done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r02t/ab.txt"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
