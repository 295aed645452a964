#!/bin/bash
# (HJ_SLAB_DEEP_SCHEDULE was an experimental knob of this call; the schedule lost and was removed again: profiles/r04_deep_single_schedule.txt)
# round 4, call 21: kernel timeline of the deep-halo step on the self ring, 201-plane slab of 201^2 planes: two-stream schedule against
# the one-stream schedule (HJ_SLAB_DEEP_SCHEDULE=single)
out=gpurun_out/r04_run21; mkdir -p $out
export TMPDIR=/tmp; root=$PWD; cd /tmp
for sched in streams single; do
HJ_SLAB_DEEP_SCHEDULE=$sched HJ_DEBUG=2 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace_$sched -- python3 $root/tools/thin_slab_ring.py 201 1 deep > $root/$out/ring_$sched.txt 2> $root/$out/ring_$sched.err
cd $root; grep "N=1" $out/ring_$sched.txt; grep -h "tiling" $out/ring_$sched.err | sort | uniq -c | head -8
python3 tools/timeline.py $out/trace_$sched 0 30 > $out/timeline_$sched.txt; cat $out/timeline_$sched.txt
rm -rf $out/trace_$sched; cd /tmp
done
