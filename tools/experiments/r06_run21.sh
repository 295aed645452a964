#!/bin/bash
# round 6, final library: the randomised parity scripts once more (incl. the callback tracer's)
set -o pipefail
mkdir -p gpurun_out
log=gpurun_out/r06_fuzz_final.log
: > $log
run() {
  echo "== $*" | tee -a $log
  timeout -k 10 $(( $2 + 120 )) python "$@" > gpurun_out/r06_fuzz_one.log 2>&1
  rc=$?
  grep -c "" gpurun_out/r06_fuzz_one.log | tee -a $log
  tail -2 gpurun_out/r06_fuzz_one.log | cut -c1-900 | tee -a $log
  grep -o "kernel [a-z_0-9]*[^ ]*[ (a-zA-Z0-9]*)\?" gpurun_out/r06_fuzz_one.log | sort | uniq -c | sort -rn | head -12 | tee -a $log
  return $rc
}
run tests/fuzz_parity.py 120 6111 && run tests/fuzz_slabs.py 120 6212 && run tests/fuzz_big.py 120 6313 && run tests/fuzz_solver.py 50 6414 && run tests/fuzz_terms.py 50 6515 && run tests/fuzz_trace.py 120 7500 gpu
