#!/bin/bash
# round 3, run 71: PCIe-inclusive rate of the NumPy-in / NumPy-out API and the split (foreign-callback) path on the final build
out=gpurun_out/r03bs; mkdir -p $out; rm -rf $out/*
timeout -k 10 300 python tools/pcie_rate.py > $out/pcie.txt 2>&1; echo "rc=$?" >> $out/pcie.txt; tail -8 $out/pcie.txt
timeout -k 10 300 python tools/split_bench.py > $out/split.txt 2>&1; echo "rc=$?" >> $out/split.txt; tail -8 $out/split.txt
