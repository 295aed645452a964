#!/bin/bash
# round 3, run 6: fused term kernels (tests + timing), then run 5's stage-fused A/B
out=gpurun_out/r03f; mkdir -p $out; rm -f $out/*
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "term_ or fused_term" > $out/test_terms.txt 2>&1; echo "rc=$?" >> $out/test_terms.txt; tail -5 $out/test_terms.txt
timeout -k 10 300 python tools/term_timing.py 201 > $out/term_timing.txt 2>&1; cat $out/term_timing.txt
bash tools/experiments/r03_run5.sh
