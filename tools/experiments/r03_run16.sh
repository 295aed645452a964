#!/bin/bash
# round 3, run 16: phase stamps of the stage-fused pair kernel at 513^3 (diagnostic build libhj_vST.so)
out=gpurun_out/r03p; mkdir -p $out; rm -rf $out/*
L=$PWD/levelsetpy_amd/csrc
HJ_LIB=$L/libhj_vST.so HJ_FUSE12=1 HJ_TIMING_DUMP=$PWD/$out/dump.txt HJ_BENCH_SPINUP=10 timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --no-live-traffic --steps 3 --warmup 1 --repeats 1 --n 513 > $out/bench.json 2> $out/err.txt
python3 tools/f12_stamps.py $out/dump.txt | tee $out/stamps.txt
tail -c 3000000 $out/dump.txt > $out/dump_tail.txt; rm -f $out/dump.txt
