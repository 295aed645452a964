#!/bin/bash
# round 6: the randomised parity scripts over the new kernels (transposed march, 4-D full-row kernel), a few minutes each
mkdir -p gpurun_out
out=gpurun_out/r06_fuzz.log
: > $out
for f in "fuzz_parity.py 150 6101" "fuzz_slabs.py 150 6202" "fuzz_big.py 150 6303" "fuzz_solver.py 60 6404" "fuzz_terms.py 60 6505"; do
  echo "== tests/$f" >> $out
  timeout -k 10 400 python tests/$f > gpurun_out/r06_fuzz_one.log 2>&1; rc=$?
  grep -c " ok$" gpurun_out/r06_fuzz_one.log >> $out
  grep "MISMATCH\|FAILED\|Error\|error" gpurun_out/r06_fuzz_one.log | head -5 >> $out
  tail -2 gpurun_out/r06_fuzz_one.log >> $out
  grep -o "kernel [a-z_0-9]* *([a-z 0-9-]*)\|kernel [a-z_0-9]*" gpurun_out/r06_fuzz_one.log | sort | uniq -c >> $out
  [ $rc -ne 0 ] && { echo "rc=$rc" >> $out; cp gpurun_out/r06_fuzz_one.log gpurun_out/r06_fuzz_fail.log; }
done
cat $out
