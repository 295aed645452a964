#!/bin/bash
# (HJ_SLAB_DEEP_SCHEDULE was an experimental knob of this call; the schedule lost and was removed again: profiles/r04_deep_single_schedule.txt)
# round 4, call 20: the deep-halo step as ONE launch per stage on the ctx stream (HJ_SLAB_DEEP_SCHEDULE=single) against the two-stream
# schedule: bitwise tests (virtual ranks, RCCL self ring, 3-D and 4-D), then the self-ring timing of a 201-plane slab of 201^2 planes
# (one rank of the weak-scaling leg) and of the 513^3 / N slabs
out=gpurun_out/r04_run20; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_configs.py tests/test_gpu_user_ham.py -x -q -m gpu -k "deep or self_ring or ring" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for sch in streams single; do
  echo "== HJ_SLAB_DEEP_SCHEDULE=$sch pass $rep" | tee -a $out/ring.txt
  HJ_SLAB_DEEP_SCHEDULE=$sch timeout -k 10 200 python3 tools/thin_slab_ring.py 201 1 deep,sub 2>&1 | grep "N=" | tee -a $out/ring.txt
  HJ_SLAB_DEEP_SCHEDULE=$sch timeout -k 10 300 python3 tools/thin_slab_ring.py 513 2,4 deep 2>&1 | grep "N=" | tee -a $out/ring.txt
done; done
