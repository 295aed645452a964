#!/bin/bash
# round 3, run 36: launches of hj_rk_step no longer reduce a CFL bound nobody reads (dt comes from the static bound): A/B against
# HJ_KEEP_BOUNDS=1 (every launch ends in wave_max -> LDS -> 3 atomicMax per workgroup on one cache line), tiled and direct kernels
out=gpurun_out/r03aj; mkdir -p $out; rm -rf $out/*
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --no-live-traffic --steps 50 --repeats 5 $EXTRA >> $out/ab.txt 2> $out/last.err || { tail -3 $out/last.err; exit 1; }; }
for n in 31 51 101 151 201; do
  for kb in 1 0; do
    EXTRA="--n $n --no-also" run HJ_KEEP_BOUNDS=$kb
    [ $n -le 101 ] && EXTRA="--n $n --no-also" run HJ_KEEP_BOUNDS=$kb HJ_FORCE_DIRECT=1
  done
done
for kb in 1 0; do EXTRA="--n 401 --also 513,C3,C5 --steps 20" run HJ_KEEP_BOUNDS=$kb; done
python - <<'PY'
import json
for ln in open("gpurun_out/r03aj/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()[:200]); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  us/step %.2f spread %.3f  %s" % (d["value"], d["roofline"]["frac"], d["ms_per_step"] * 1e3, d["repeats"]["spread"], d["roofline"]["kernel"][:24]))
    for k, v in (d.get("also") or {}).items(): print("      also %-22s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ode_cfl or hjipde or rk" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
