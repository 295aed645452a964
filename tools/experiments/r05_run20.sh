#!/bin/bash
# bw3 copy sweep + the default bench command (PMC child passes for C3 / C5 included), timed
mkdir -p gpurun_out
( while true; do date >> gpurun_out/r20_heartbeat.txt; sleep 45; done ) & HB=$!
timeout -k 10 300 tools/ubench/bw3 > gpurun_out/r20_bw3.txt 2>&1
tail -1 gpurun_out/r20_bw3.txt
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r20_bench.json 2> gpurun_out/r20_bench.err
rc=$?
echo "bench rc=$rc wall=$(( $(date +%s) - t0 )) s"
kill $HB
exit $rc
