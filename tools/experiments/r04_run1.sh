#!/bin/bash
# round 4, call 1: the new GPU tests (full-size oracle parity, C5 ring of slabs), the self-launching bench on one card
# (gloo + torch transport rehearsal, then the refusal), and the default bench line with the parity / API legs.
out=gpurun_out/r04_run1; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu > $out/pytest_round4.log 2>&1; echo "pytest round4 rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest_round4.log
HJ_BENCH_ONE_DEVICE=1 HJ_BENCH_BACKEND=gloo HJ_SLAB_TRANSPORT=torch HJ_BENCH_SPINUP=20 timeout -k 10 300 python3 bench.py --gpus 2 --global-n 129 --steps 5 --warmup 2 --repeats 3 > $out/launch2.json 2> $out/launch2.err; echo "launch2 rc=$?" | tee -a $out/summary.txt
HJ_BENCH_ONE_DEVICE=1 HJ_BENCH_BACKEND=gloo HJ_SLAB_TRANSPORT=torch HJ_BENCH_SPINUP=20 timeout -k 10 300 python3 bench.py --gpus 2 --workload C5 --global-n 33 --steps 5 --warmup 2 --repeats 3 > $out/launch2_c5.json 2> $out/launch2_c5.err; echo "launch2 c5 rc=$?" | tee -a $out/summary.txt
timeout -k 10 120 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $out/refuse2.json 2> $out/refuse2.err; echo "refuse2 rc=$? (expected 2)" | tee -a $out/summary.txt
HJ_BENCH_SPINUP=50 timeout -k 10 300 python3 bench.py --gpus 1 --workload C5 --steps 5 --warmup 2 --repeats 3 > $out/slab1_c5.json 2> $out/slab1_c5.err; echo "slab1 c5 (self ring, RCCL) rc=$?" | tee -a $out/summary.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee -a $out/summary.txt
cat $out/summary.txt
