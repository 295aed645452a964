#!/bin/bash
# round 4, call 7: kernel timeline of the 65-plane slab (513^3 over 8 ranks) on the self ring, gated schedule (and overlap for reference)
out=gpurun_out/r04_run7; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp; root=$PWD; cd /tmp
for sched in gated overlap; do
HJ_SLAB_SCHEDULE=$sched HJ_DEBUG=2 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace_$sched -- python3 $root/tools/thin_slab_ring.py 513 8 sub > $root/$out/ring_$sched.txt 2> $root/$out/ring_$sched.err
cd $root; grep "N=8" $out/ring_$sched.txt; grep -h "gated launch\|tiling" $out/ring_$sched.err | head -3
python3 tools/timeline.py $out/trace_$sched 0 24 > $out/timeline_$sched.txt; cat $out/timeline_$sched.txt
rm -rf $out/trace_$sched; cd /tmp
done
