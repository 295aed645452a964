#!/bin/bash
# round 3, run 67: thin slabs with the 256-thread kernel shapes (2-3 workgroups per CU: finer round quantisation) on the self ring
out=gpurun_out/r03bo; mkdir -p $out; rm -rf $out/*
run() { echo "== $*" >> $out/ring_all.txt; env "$@" timeout -k 10 400 python3 tools/thin_slab_ring.py 513 4,8 sub,deep >> $out/ring_all.txt 2> $out/ring.err || { tail -5 $out/ring.err >> $out/ring_all.txt; }; }
run HJ_X=0
run HJ_PAIR=0 HJ_NT=256 HJ_R=2 HJ_KH=2 HJ_OCC=2 HJ_PD=2
run HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_PAIR_KH=2
run HJ_PAIR_RING=0
grep -v "version\|Hostname\|Librccl" $out/ring_all.txt
