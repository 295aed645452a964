#!/bin/bash
out=gpurun_out/r02j; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/test.log 2>&1; echo "pytest rc=$?"; tail -30 $out/test.log
