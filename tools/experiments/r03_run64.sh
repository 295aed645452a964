#!/bin/bash
# round 3, run 64: per-substep slab schedule "overlap2": like overlap, but interior_s also waits for the exchange of substep s-1 (edges_s first on free CUs)
out=gpurun_out/r03bl; mkdir -p $out; rm -rf $out/*
for sch in overlap overlap2 serial; do
  echo "== HJ_SLAB_SCHEDULE=$sch" >> $out/ring_all.txt
  HJ_SLAB_SCHEDULE=$sch timeout -k 10 400 python3 tools/thin_slab_ring.py 513 4,8 sub >> $out/ring_all.txt 2> $out/ring.err || { tail -5 $out/ring.err; exit 1; }
done
grep -v "version\|Hostname\|Librccl" $out/ring_all.txt
