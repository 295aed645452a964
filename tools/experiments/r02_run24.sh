#!/bin/bash
# prologue rework of the pair kernel: correctness (bitwise pair test + forced-pair suite subset), stamps, throughput
out=gpurun_out/r02x; mkdir -p $out; rm -f $out/*
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "pair or virtual or plain" > $out/t1.log 2>&1; tail -2 $out/t1.log
HJ_PAIR=2 timeout -k 10 400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_configs.py::test_pair_kernel_bitwise_equals_scalar_kernel > $out/t2.log 2>&1; tail -2 $out/t2.log
for n in 201; do
HJ_PAIR=1 HJ_TIMING_DUMP=$out/t$n.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --n $n --steps 3 --warmup 3 --repeats 1 > $out/b$n.json 2> $out/b$n.err
python tools/pair_timing.py $out/t$n.txt > $out/s$n.txt; grep -A6 "stage=3" $out/s$n.txt | cut -c1-330
done
run() { echo "== $* $EXTRA" >> $out/ab.txt; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 30 --repeats 5 $EXTRA >> $out/ab.txt 2>> $out/ab.err; }
for n in 201 301 401 513; do EXTRA="--n $n" run HJ_PAIR=1; done
python - <<'PY'
import json
for ln in open("gpurun_out/r02x/ab.txt"):
    if not ln.startswith("{"): print(ln.rstrip()); continue
    d = json.loads(ln)
    print("   %.4e  frac %.3f  ms/step %.4f spread %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["repeats"]["spread"]))
PY
