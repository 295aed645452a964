#!/bin/bash
# round 3, run 11: NumPy-order ENO path: bitwise against the reference goldens?
out=gpurun_out/r03k; mkdir -p $out; rm -f $out/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "eno_paths_bitwise or golden" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -8 $out/test.txt
cat gpurun_out/eno_bitwise.txt
