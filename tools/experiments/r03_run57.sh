#!/bin/bash
# round 3, run 57: randomised bitwise stress of the pair kernel against the one-cell-per-lane kernel (300 cases, two seeds) and the
# run-to-run determinism check on the final build
out=gpurun_out/r03be; mkdir -p $out; rm -rf $out/*
timeout -k 10 500 python tools/stress_pair.py 300 11 > $out/stress_a.txt 2>&1; echo "rc=$?" >> $out/stress_a.txt; tail -3 $out/stress_a.txt
timeout -k 10 500 python tools/stress_pair.py 300 12 > $out/stress_b.txt 2>&1; echo "rc=$?" >> $out/stress_b.txt; tail -3 $out/stress_b.txt
timeout -k 10 300 python tools/det_check.py > $out/det.txt 2>&1; echo "rc=$?" >> $out/det.txt; tail -4 $out/det.txt
