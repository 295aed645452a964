#!/bin/bash
out=gpurun_out/r02m; mkdir -p $out; rm -f $out/*
for n in 201 513; do
HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_stamp.so HJ_TIMING_DUMP=$PWD/$out/dump$n.txt HJ_BENCH_SPINUP=3 timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --n $n --steps 2 --warmup 1 --repeats 1 > $out/b$n.json 2> $out/b$n.err
python tools/stamp_summary.py $out/dump$n.txt
done
