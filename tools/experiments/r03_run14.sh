#!/bin/bash
# round 3, run 14: kernel timeline of the 65-plane slab (513^3 over 8 ranks) on the self ring, per-substep schedule
out=gpurun_out/r03n; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp; root=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace -- python3 $root/tools/thin_slab_ring.py 513 8 sub > $root/$out/ring.txt 2> $root/$out/ring.err
cd $root; cat $out/ring.txt
python3 tools/timeline.py $out/trace 0 48 > $out/timeline.txt; cat $out/timeline.txt
python3 tools/thin_slab_ring.py 513 2,4,8 > $out/ring_all.txt 2>> $out/ring.err; cat $out/ring_all.txt
