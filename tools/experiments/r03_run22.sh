#!/bin/bash
# round 3, run 22: RCCL point-to-point channel count on the thin-slab self ring (N = 8, 65 planes): the exchange kernel moves
# 2 x 6.3 MB in 49 us with the default channel count
out=gpurun_out/r03v; mkdir -p $out; rm -rf $out/*
for ch in 0 2 4 8 16 32; do
  echo "== NCCL_MIN_P2P_NCHANNELS=$ch" >> $out/ring.txt
  if [ $ch = 0 ]; then timeout -k 10 300 python3 tools/thin_slab_ring.py 513 8,4 sub >> $out/ring.txt 2> $out/err_$ch.txt
  else NCCL_MIN_P2P_NCHANNELS=$ch NCCL_MAX_P2P_NCHANNELS=$ch timeout -k 10 300 python3 tools/thin_slab_ring.py 513 8,4 sub >> $out/ring.txt 2> $out/err_$ch.txt; fi
done
grep -v "version\|Hostname\|Librccl" $out/ring.txt
