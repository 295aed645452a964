#!/bin/bash
# per-workgroup timing (setup / loop / epilogue) at 201^3 and 51^3-class sizes after the DPP reductions
out=gpurun_out/r02ao; mkdir -p $out; rm -f $out/*
for n in 201; do
HJ_TIMING_DUMP=$out/t$n.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --n $n --steps 3 --warmup 3 --repeats 1 > /dev/null 2> $out/b$n.err
python tools/pair_timing.py $out/t$n.txt | grep "launch\|start\|end  \|prolog\|loop :\|epilog\|prologue stamps" | cut -c1-330
done
for n in 101; do
HJ_TIMING_DUMP=$out/t$n.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --n $n --steps 3 --warmup 3 --repeats 1 > /dev/null 2> $out/b$n.err
python tools/block_timing.py $out/t$n.txt | head -12
done
rm -f $out/t*.txt
