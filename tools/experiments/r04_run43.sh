#!/bin/bash
# round 4, call 43: BASELINE C4 (513^3, one odeCFL3 step) and C5 (129^4 fp32, one term evaluation) against the CPU oracle at FULL size (one host core, minutes)
out=gpurun_out/r04_run43; mkdir -p $out
free -g | head -2
timeout -k 10 1100 python3 tests/diag/full_size_oracle_check.py both 2>&1 | grep -v amdgpu.ids | tee $out/full_size.txt
