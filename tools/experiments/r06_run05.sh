#!/bin/bash
# round 6: the slabs of C5 (129^4 fp32) on the self ring -- never timed before (missing 1)
mkdir -p gpurun_out
out=gpurun_out/r06_c5_slabs.log
: > $out
HJ_DEBUG=1 timeout -k 10 300 python tools/thin_slab_ring.py 129 8,4,2 sub,deep,plain C5 2>&1 | grep "^\[hj\]\|^C5" >> $out || exit 1
cat $out
