#!/bin/bash
# round 3, run 38: whole GPU suite + the driver's bench command after the C5 pair default and the unread-bound change
out=gpurun_out/r03al; mkdir -p $out; rm -rf $out/*
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/test.txt 2>&1; rc=$?; echo "rc=$rc" >> $out/test.txt; tail -5 $out/test.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 1500 $out/bench.json
