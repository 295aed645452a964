#!/bin/bash
# round 6: tile shape of the transposed march on the 65-plane slab (plain = no ring): HJ_FULL_ROWS forces the row extent E2, E1 = 2048 / E2
mkdir -p gpurun_out
out=gpurun_out/r06_xp_shapes.log
: > $out
for e2 in 34 46 58 66 74 86 104 130 172 258; do
  echo "== HJ_XP=1 HJ_FULL_ROWS=$e2" >> $out
  HJ_XP=1 HJ_FULL_ROWS=$e2 HJ_DEBUG=1 timeout -k 10 120 python tools/thin_slab_ring.py 513 8 plain 2>&1 | grep "^\[hj\] trans\|^N=" >> $out || exit 1
done
for mc in 20 50 70 120; do
  echo "== HJ_XP=1 HJ_MIN_CHUNK=$mc" >> $out
  HJ_XP=1 HJ_MIN_CHUNK=$mc HJ_DEBUG=1 timeout -k 10 120 python tools/thin_slab_ring.py 513 8 plain 2>&1 | grep "^\[hj\] trans\|^N=" >> $out || exit 1
done
cat $out
