#!/bin/bash
# round 3, run 66: deep-halo step on one compute stream (HJ_SLAB_DEEP=serial) against the two-stream form (overlap): slab tests, self ring N = 2, 4, 8
out=gpurun_out/r03bn; mkdir -p $out; rm -rf $out/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "slab or virtual or ring or deep" > $out/test.txt 2>&1; rc=$?; echo "rc=$rc" >> $out/test.txt; tail -4 $out/test.txt
[ $rc -eq 0 ] || exit 1
for dp in overlap serial; do
  echo "== HJ_SLAB_DEEP=$dp" >> $out/ring_all.txt
  HJ_SLAB_DEEP=$dp timeout -k 10 400 python3 tools/thin_slab_ring.py 513 2,4,8 deep >> $out/ring_all.txt 2> $out/ring.err || { tail -5 $out/ring.err; exit 1; }
done
grep -v "version\|Hostname\|Librccl" $out/ring_all.txt
export TMPDIR=/tmp; root=$PWD; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace -- python3 $root/tools/thin_slab_ring.py 513 4 deep > $root/$out/ring.txt 2>> $root/$out/ring.err
cd $root; python3 tools/timeline.py $out/trace 0 22 > $out/timeline.txt; cat $out/timeline.txt
