#!/bin/bash
# round 3, run 68: the whole GPU suite with the pair kernel forced at every size and the in-launch epsilon reduction at every size
out=gpurun_out/r03bp; mkdir -p $out; rm -rf $out/*
HJ_PAIR=2 HJ_EPS_FUSE_MIN_CELLS=0 timeout -k 10 1100 python -m pytest tests -q -m gpu > $out/test_pair2.txt 2>&1; echo "rc=$?" >> $out/test_pair2.txt; tail -12 $out/test_pair2.txt
HJ_PAIR=2 HJ_PAIR_RING=1 timeout -k 10 1100 python -m pytest tests -q -m gpu > $out/test_ring1.txt 2>&1; echo "rc=$?" >> $out/test_ring1.txt; tail -12 $out/test_ring1.txt
