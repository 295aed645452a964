#!/bin/bash
# round 3, run 42: per-substep slab schedule "serial" (edges alone, then the interior with the exchange under it) against "overlap"
out=gpurun_out/r03ap; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp; root=$PWD
for sch in overlap serial; do
  echo "== HJ_SLAB_SCHEDULE=$sch" >> $out/ring_all.txt
  HJ_SLAB_SCHEDULE=$sch timeout -k 10 400 python3 tools/thin_slab_ring.py 513 2,4,8 sub >> $out/ring_all.txt 2> $out/ring.err || { tail -5 $out/ring.err; exit 1; }
done
grep -v "version\|Hostname\|Librccl" $out/ring_all.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace -- python3 $root/tools/thin_slab_ring.py 513 8 sub > $root/$out/ring.txt 2>> $root/$out/ring.err
cd $root; python3 tools/timeline.py $out/trace 0 24 > $out/timeline.txt; cat $out/timeline.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "slab or virtual or ring" > $out/test.txt 2>&1; echo "rc=$?" >> $out/test.txt; tail -3 $out/test.txt
