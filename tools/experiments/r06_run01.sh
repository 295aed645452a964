#!/bin/bash
# round 6, first GPU call: parity of the transposed march, then the 65-plane slab of C4 with and without it
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > gpurun_out/r06_t1.log 2>&1
rc=$?
tail -5 gpurun_out/r06_t1.log
[ $rc -ne 0 ] && exit $rc
for xp in 0 1; do
  echo "== HJ_XP=$xp" >> gpurun_out/r06_thin1.log
  HJ_XP=$xp HJ_DEBUG=1 timeout -k 10 300 python tools/thin_slab_ring.py 513 8 sub,deep,plain >> gpurun_out/r06_thin1.log 2>&1 || exit 1
done
grep -v "^\[hj\] autotune" gpurun_out/r06_thin1.log | tail -30
