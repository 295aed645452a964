#!/bin/bash
# round 6: the RCCL self exchange (48 us for 2 x 6.3 MB through ONE channel) sits on the ring's critical path -- p2p channel counts
mkdir -p gpurun_out
out=gpurun_out/r06_rccl_channels.log
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python tools/thin_slab_ring.py 513 8 sub,deep 2>&1 | grep "^N=" >> $out || exit 1; }
run HJ_XP=1
run HJ_XP=1 NCCL_NCHANNELS_PER_PEER=4
run HJ_XP=1 NCCL_NCHANNELS_PER_PEER=8
run HJ_XP=1 NCCL_MIN_P2P_NCHANNELS=8 NCCL_MAX_P2P_NCHANNELS=8
run HJ_XP=1 NCCL_MIN_NCHANNELS=8 NCCL_NCHANNELS_PER_PEER=8
run HJ_XP=0 NCCL_NCHANNELS_PER_PEER=8
run HJ_XP=1 NCCL_NCHANNELS_PER_PEER=8 HJ_SLAB_SCHEDULE=serial
run HJ_XP=0 NCCL_NCHANNELS_PER_PEER=8 HJ_SLAB_SCHEDULE=serial
run HJ_XP=1 NCCL_NCHANNELS_PER_PEER=8 HJ_SLAB_SCHEDULE=gated
cat $out
