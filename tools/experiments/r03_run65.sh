#!/bin/bash
# round 3, run 65: kernel timeline of the deep-halo schedule on the 129-plane slab (N = 4), self ring
out=gpurun_out/r03bm; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp; root=$PWD; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace -- python3 $root/tools/thin_slab_ring.py 513 4 deep > $root/$out/ring.txt 2> $root/$out/ring.err
cd $root; grep "N=" $out/ring.txt; python3 tools/timeline.py $out/trace 0 26 > $out/timeline.txt; cat $out/timeline.txt
