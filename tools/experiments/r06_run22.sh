#!/bin/bash
# round 6, final library: long runs of the two fuzzers that cover this round's host-side machinery
mkdir -p gpurun_out
log=gpurun_out/r06_fuzz_long.log
: > $log
for spec in "tests/fuzz_trace.py 420 8800 gpu" "tests/fuzz_parity.py 300 6711" "tests/fuzz_big.py 200 6813"; do
  echo "== $spec" | tee -a $log
  timeout -k 10 700 python $spec > gpurun_out/r06_fuzz_one.log 2>&1; rc=$?
  tail -1 gpurun_out/r06_fuzz_one.log | cut -c1-700 | tee -a $log
  [ $rc -ne 0 ] && { tail -5 gpurun_out/r06_fuzz_one.log | tee -a $log; exit $rc; }
done
exit 0
