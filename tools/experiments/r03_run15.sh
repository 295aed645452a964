#!/bin/bash
# round 3, run 15: thin slab (N = 8, 65 planes) with smaller tiles (one round of workgroups, one chunk) on the self ring
out=gpurun_out/r03o; mkdir -p $out; rm -rf $out/*
for tc in 0 1122 1400 900 700; do
  echo "== HJ_TILE_CELLS=$tc" >> $out/ring.txt
  HJ_TILE_CELLS=$tc HJ_DEBUG=1 timeout -k 10 300 python3 tools/thin_slab_ring.py 513 8,4 sub >> $out/ring.txt 2> $out/err_$tc.txt
  grep "tiling" $out/err_$tc.txt | sort | uniq -c | sort -rn | head -3 >> $out/ring.txt
done
grep -v "version\|Hostname\|Librccl" $out/ring.txt
