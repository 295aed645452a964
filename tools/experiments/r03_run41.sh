#!/bin/bash
# round 3, run 41: thin-slab self ring (513^3 over N = 2, 4, 8) after the unread-bound change; timeline at N = 8
out=gpurun_out/r03ao; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp; root=$PWD
timeout -k 10 400 python3 tools/thin_slab_ring.py 513 2,4,8 > $out/ring_all.txt 2> $out/ring.err || { tail -5 $out/ring.err; exit 1; }
cat $out/ring_all.txt
HJ_KEEP_BOUNDS=1 timeout -k 10 400 python3 tools/thin_slab_ring.py 513 2,4,8 > $out/ring_all_keep.txt 2>> $out/ring.err; cat $out/ring_all_keep.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace -- python3 $root/tools/thin_slab_ring.py 513 8 sub > $root/$out/ring.txt 2>> $root/$out/ring.err
cd $root; python3 tools/timeline.py $out/trace 0 30 > $out/timeline.txt; cat $out/timeline.txt
