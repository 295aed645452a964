#!/bin/bash
# round 4, call 22: randomised bitwise stress of the pair kernel incl. chunk pairs marching in opposite directions (tools/stress_pair.py,
# 400 cases, two seeds), then run-to-run determinism of the default 201^3 step (tools/det_check.py)
out=gpurun_out/r04_run22; mkdir -p $out
timeout -k 10 500 python3 tools/stress_pair.py 200 41 > $out/stress_a.txt 2>&1; echo "stress a rc=$?"; tail -2 $out/stress_a.txt
timeout -k 10 500 python3 tools/stress_pair.py 200 42 > $out/stress_b.txt 2>&1; echo "stress b rc=$?"; tail -2 $out/stress_b.txt
timeout -k 10 200 python3 tools/det_check.py > $out/det.txt 2>&1; echo "det rc=$?"; tail -3 $out/det.txt
