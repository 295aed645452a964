#!/bin/bash
# round 4, call 24: bit-faithful ENO3 with the third-order term's operand and coefficient SELECTED (one product, one sum) instead of two
# arms the compiler turned into divergent branches: bitwise tests against the reference's goldens, then C3 (4096^2 ENO3) and 201^3 ENO3 / ENO2
out=gpurun_out/r04_run24; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "eno or ENO or golden or c3 or 4096 or bitwise or oracle" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-traffic --also C3,ENO3,ENO2 --repeats 15 --steps 20 > $out/bench_$rep.json 2> $out/bench_$rep.err || tail -3 $out/bench_$rep.err
python3 - $out/bench_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["also"].items():
    if isinstance(v, dict) and "value" in v: print("%-40s %.4e  frac %.4f  %s" % (k, v["value"], v.get("roofline_frac") or 0, (v.get("roofline_valu") or {}).get("frac", "")))
PY
done
