#!/bin/bash
# after the setup rework of both kernels: GPU suites, then the size sweep and the other workloads
out=gpurun_out/r02y; mkdir -p $out; rm -f $out/*
timeout -k 10 400 python -m pytest tests -m gpu -q > $out/t1.log 2>&1; tail -2 $out/t1.log
HJ_PAIR=2 timeout -k 10 400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_configs.py::test_pair_kernel_bitwise_equals_scalar_kernel > $out/t2.log 2>&1; tail -2 $out/t2.log
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for n in 51 65 101 129 151 201 251 301 401 513; do EXTRA="--n $n" run HJ_PAIR=1; done
echo "== also" >> $out/ab.txt
python bench.py --no-cpu-baseline --steps 20 --also WENO5,ENO3,ENO2,C3,C5 >> $out/ab.txt 2>> $out/ab.err
python - <<'PY'
import json
for ln in open("gpurun_out/r02y/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
    for k, v in d.get("also", {}).items():
        print("      also %-26s %.4e frac %.3f" % (k, v.get("value", 0), v.get("roofline_frac", 0)))
PY
