#!/bin/bash
# round 4, call 28: final validation of the tree as committed: build() from scratch is NOT run here (the .so travels); whole GPU suite, smoke(), default bench
out=gpurun_out/r04_run28; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; tail -2 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 10 600 python3 bench.py > $out/bench_noflags.json 2> $out/bench_noflags.err; echo "bench (no flags) rc=$?"; cut -c1-300 $out/bench_noflags.json
