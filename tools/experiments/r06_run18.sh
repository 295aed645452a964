#!/bin/bash
# round 6: the per-substep schedules on the 65-plane self ring once more (serial / overlap / gated) x (axis-0 / transposed interior)
mkdir -p gpurun_out
out=gpurun_out/r06_ring_schedules.log
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python tools/thin_slab_ring.py 513 8 sub 2>&1 | grep "^C4\|^N=" >> $out || exit 1; }
run HJ_XP=0
run HJ_XP=0 HJ_SLAB_SCHEDULE=serial
run HJ_XP=2 HJ_SLAB_SCHEDULE=serial
run HJ_XP=0 HJ_SLAB_SCHEDULE=gated
run HJ_XP=2 HJ_SLAB_SCHEDULE=overlap
run HJ_XP=0 HJ_SLAB_SCHEDULE=overlap2
run HJ_XP=2 HJ_SLAB_SCHEDULE=overlap2
cat $out
