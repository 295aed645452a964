#!/bin/bash
out=gpurun_out/r02w; mkdir -p $out; rm -f $out/*
for n in 201 401; do
HJ_PAIR=1 HJ_TIMING_DUMP=$out/t$n.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --n $n --steps 3 --warmup 3 --repeats 1 > $out/b$n.json 2> $out/b$n.err
python tools/pair_timing.py $out/t$n.txt > $out/s$n.txt; cat $out/s$n.txt
done
