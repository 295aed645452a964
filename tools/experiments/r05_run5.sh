#!/bin/bash
# round 5, call 5: pair4 kernel with column PAIRS instead of single halo cells; the three compile-time tiles against each other and the generic kernel
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run5; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pair4 or 4d or c5 or pendulum or four_d" > $out/tests.log 2>&1; tail -3 $out/tests.log
export C5_STEPS=20 C5_WARMUP=40
for rep in 1 2; do
  for v in 0 1 2 old; do
    echo "== tile $v" >> $out/c5.txt
    if [ $v = old ]; then HJ_PAIR4=0 python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1
    else HJ_TILE4_SEL=$v HJ_DEBUG=1 python3 tools/bench_configs.py c5 >> $out/c5.txt 2>&1; fi
  done
done
grep -v "amdgpu.ids" $out/c5.txt
