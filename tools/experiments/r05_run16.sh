#!/bin/bash
# lean ENO3 in the two-pair shape on 2-D grids: parity + C3 fast A/B
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -q -m gpu -k "fast_eno" > gpurun_out/r16_tests.log 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --also "C3,C3 fast" > gpurun_out/r16_bench_big.log 2>&1
HJ_PAIR_NT=256 HJ_PAIR_R=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --also "C3 fast" > gpurun_out/r16_bench_small.log 2>&1
tail -2 gpurun_out/r16_tests.log
