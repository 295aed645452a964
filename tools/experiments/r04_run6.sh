#!/bin/bash
# round 4, call 6: the gated slab schedule -- correctness on the self ring (short timeouts: a gate that never opens must not hang
# the box), then the thin-slab timing of all per-substep schedules at 513^3 / N
out=gpurun_out/r04_run6; mkdir -p $out; rm -f $out/*
timeout -k 10 120 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "gated" > $out/pytest_gated.log 2>&1; rc=$?; echo "pytest gated rc=$rc"; tail -4 $out/pytest_gated.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -x -q -m gpu -k "self_ring or slab" > $out/pytest_slab.log 2>&1; rc=$?; echo "pytest slab rc=$rc"; tail -4 $out/pytest_slab.log
[ $rc -ne 0 ] && exit 1
for sched in overlap serial gated; do
  echo "== HJ_SLAB_SCHEDULE=$sched" | tee -a $out/ring.txt
  HJ_SLAB_SCHEDULE=$sched timeout -k 10 200 python tools/thin_slab_ring.py 513 2,4,8 sub 2>> $out/ring.err | tee -a $out/ring.txt
done
