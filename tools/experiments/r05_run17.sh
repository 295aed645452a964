#!/bin/bash
# thin slabs (65 planes of 513^3): tile / chunk geometry sweep on the RCCL self ring, both steppers
mkdir -p gpurun_out
out=gpurun_out/r17_thin.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python tools/thin_slab_ring.py 513 8 sub,deep >> $out 2>&1 || echo "FAILED" >> $out; }
run HJ_X=0
run HJ_MIN_CHUNK=80
run HJ_TILE_CELLS=1024
run HJ_TILE_CELLS=1024 HJ_MIN_CHUNK=80
run HJ_TILE_CELLS=1024 HJ_MIN_CHUNK=30
run HJ_PAIR_NT=256 HJ_PAIR_R=1
run HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_MIN_CHUNK=80
run HJ_PAIR_NT=256 HJ_PAIR_R=1 HJ_MIN_CHUNK=30
run HJ_PAIR=0
run HJ_PAIR=0 HJ_MIN_CHUNK=80
run HJ_TILE_CELLS=1400 HJ_MIN_CHUNK=80
run HJ_SLAB_SCHEDULE=serial HJ_MIN_CHUNK=80
run HJ_SLAB_SCHEDULE=serial HJ_TILE_CELLS=1024 HJ_MIN_CHUNK=80
cat $out
