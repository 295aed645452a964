#!/bin/bash
# round 4, call 19: full GPU suite on the build with size-gated paired chunk directions, then the default bench line
out=gpurun_out/r04_run19; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; tail -3 $out/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"
cut -c1-900 $out/bench_default.json
