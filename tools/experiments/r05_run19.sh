#!/bin/bash
# thin slab, plain grid: per-workgroup timing of the pair kernel
mkdir -p gpurun_out
out=gpurun_out/r19_timing.txt
: > $out
for mc in 4 30 80; do
  echo "== HJ_MIN_CHUNK=$mc" >> $out
  HJ_MIN_CHUNK=$mc HJ_TIMING_DUMP=/tmp/td$mc.txt timeout -k 10 120 python tools/thin_slab_ring.py 513 8 plain 2>&1 | grep "^N=" >> $out
  python tools/pair_timing.py /tmp/td$mc.txt >> $out 2>&1
done
cat $out
