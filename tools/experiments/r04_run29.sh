#!/bin/bash
# round 4, call 29: grids that need the 288 GB of the card: 1025^3 and 1291^3 fp64 (8.6 / 17.2 GB per array), tiled against direct kernel, bitwise
out=gpurun_out/r04_run29; mkdir -p $out
timeout -k 10 500 python3 tools/big_grid_check.py 1025 > $out/big_1025.txt 2>&1; echo "1025 rc=$?"; grep -v amdgpu.ids $out/big_1025.txt | tail -4
timeout -k 10 500 python3 tools/big_grid_check.py 1291 > $out/big_1291.txt 2>&1; echo "1291 rc=$?"; grep -v amdgpu.ids $out/big_1291.txt | tail -4
