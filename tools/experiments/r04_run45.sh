#!/bin/bash
# round 4, call 45: whole GPU suite with the launch-time tile tuner active at EVERY size (HJ_AUTOTUNE_MIN_MCELLS=0: tile shapes rotate while the tests
# compare results bitwise) and, separately, with 2-plane chunks and the deepest halo ring (HJ_MIN_CHUNK=2 HJ_PAIR_AH=3 HJ_PAIR=2)
out=gpurun_out/r04_run45; mkdir -p $out
HJ_AUTOTUNE_MIN_MCELLS=0 timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $out/pytest_tuner.log 2>&1; echo "tuner everywhere rc=$?"; tail -2 $out/pytest_tuner.log | cut -c1-200
HJ_MIN_CHUNK=2 HJ_PAIR_AH=3 HJ_PAIR=2 HJ_PAIR_RING=1 timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $out/pytest_chunks.log 2>&1; echo "2-plane chunks + ring rc=$?"; tail -2 $out/pytest_chunks.log | cut -c1-200
