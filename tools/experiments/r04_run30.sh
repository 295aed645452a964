#!/bin/bash
# round 4, call 30: whole GPU suite with the pair kernel and paired chunk directions forced at every size (HJ_PAIR=2 HJ_PAIR_DIRS=1
# HJ_EPS_FUSE_MIN_CELLS=0), incl. the slab steppers (launches over plane ranges that start below 0 / end beyond n) and the billion-cell grid
out=gpurun_out/r04_run30; mkdir -p $out
HJ_PAIR=2 HJ_PAIR_DIRS=1 HJ_EPS_FUSE_MIN_CELLS=0 timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $out/pytest_forced.log 2>&1; echo "forced rc=$?"; tail -8 $out/pytest_forced.log | cut -c1-300
