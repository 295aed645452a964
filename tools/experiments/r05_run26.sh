#!/bin/bash
# non-uniform chunk counts: full GPU suite, then the thin slab with and without
mkdir -p gpurun_out
( while true; do date >> gpurun_out/r26_heartbeat.txt; sleep 45; done ) & HB=$!
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r26_gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r26_gpu_tests.log
if [ $rc -eq 0 ]; then
  out=gpurun_out/r26_thin.txt; : > $out
  for nu in 1 0; do
    echo "== HJ_NONUNIFORM=$nu" >> $out
    HJ_NONUNIFORM=$nu timeout -k 10 200 python tools/thin_slab_ring.py 513 8 plain,sub,deep 2>&1 | grep "^N=" >> $out
    HJ_NONUNIFORM=$nu timeout -k 10 200 python tools/thin_slab_ring.py 513 4 plain,sub,deep 2>&1 | grep "^N=" >> $out
  done
  cat $out
fi
kill $HB
exit $rc
