#!/bin/bash
# full GPU suite after the round's changes
mkdir -p gpurun_out
( while true; do date >> gpurun_out/r25_heartbeat.txt; sleep 45; done ) & HB=$!
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r25_gpu_tests.log 2>&1
rc=$?
kill $HB
tail -5 gpurun_out/r25_gpu_tests.log
exit $rc
