#!/bin/bash
# round 3, run 3: counters of the stage-fused pair kernel at 513^3 (and of the unfused default for reference)
HJ_FUSE12=1 bash tools/r02_pmc.sh f12v_513 --n 513 > gpurun_out/r03c_f12v.txt 2>&1
HJ_FUSE12=0 bash tools/r02_pmc.sh unf_513 --n 513 > gpurun_out/r03c_unf.txt 2>&1
tail -60 gpurun_out/r03c_f12v.txt
