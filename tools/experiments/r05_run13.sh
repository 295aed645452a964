#!/bin/bash
# round 5, call 13: ablation of the headline pair kernel (tune builds; bits: 1 no Hamiltonian arithmetic, 2 no LDS stencil reads, 8 no halo loads,
# 16 no halo LDS stores) at 201^3 and 513^3 -- what does the streaming skeleton (27) take of the launch?
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run13; rm -rf $out; mkdir -p $out
for n in 201 513; do
for rep in 1 2; do
  for v in 0 1 2 3 24 27; do
    HJ_LIB=$root/levelsetpy_amd/csrc/libhj_vHA$v.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 20 --warmup 5 --repeats 15 --n $n > $out/b_${n}_$v.json 2> $out/b_${n}_$v.err
    python3 - $out/b_${n}_$v.json $n $v >> $out/summary.txt <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("n=%s ablate %2s: %.4e  frac %.4f  ms/step %.4f" % (sys.argv[2], sys.argv[3], d["value"], d["roofline"]["frac"], d["ms_per_step"]))
except Exception as e:
    print("n=%s ablate %s: failed %r" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
done
cat $out/summary.txt
