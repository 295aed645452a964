#!/bin/bash
# round 3, run 46: final validation of the round: whole GPU suite, the driver's bench command, tools/profile_round.sh
out=gpurun_out/r03at; mkdir -p $out; rm -rf $out/*
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/test.txt 2>&1; rc=$?; echo "rc=$rc" >> $out/test.txt; tail -4 $out/test.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 900 bash tools/profile_round.sh r03prof > $out/profile_round.txt 2>&1; echo "profile rc=$?"; tail -30 $out/profile_round.txt
