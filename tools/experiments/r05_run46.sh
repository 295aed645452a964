#!/bin/bash
# late round 5: the whole GPU suite under global knobs that move every launch to another kernel shape (as r04_run45 / 46 did for round 4's code):
# this round added the bound pass, the local Lax-Friedrichs evaluation, deltaT from device memory and the direct-kernel default to all of them
mkdir -p gpurun_out
o=gpurun_out/r46_knobs.txt; : > $o
run() { name=$1; shift; env "$@" timeout -k 10 500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r46_$name.log 2>&1; echo "$name ($*): rc=$? $(tail -1 gpurun_out/r46_$name.log)" >> $o; date >> gpurun_out/r46_heartbeat.txt; }
run pair2 HJ_PAIR=2 HJ_MIN_CHUNK=2
run pair0 HJ_PAIR=0 HJ_MIN_CHUNK=1
run tune0 HJ_AUTOTUNE_MIN_MCELLS=0
run direct HJ_DIRECT_BELOW=140000
cat $o
