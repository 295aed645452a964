#!/bin/bash
out=gpurun_out/r02c; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "stage" > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
tail -25 $out/test.log
