#!/bin/bash
# end-of-round check: smoke(), the whole GPU suite, the default bench command
mkdir -p gpurun_out
( while true; do date >> gpurun_out/r32_heartbeat.txt; sleep 45; done ) & HB=$!
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r32_smoke.txt 2>&1; s=$?
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r32_gpu_tests.log 2>&1; t=$?
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r32_bench.json 2> gpurun_out/r32_bench.err; b=$?
echo "smoke rc=$s tests rc=$t bench rc=$b wall=$(( $(date +%s) - t0 )) s"
tail -1 gpurun_out/r32_smoke.txt; tail -2 gpurun_out/r32_gpu_tests.log
kill $HB
exit $(( s + t + b ))
