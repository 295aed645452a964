#!/bin/bash
# round 3, run 20: does RCCL accept two ranks on ONE GPU?  If so the native multi-rank slab path (ncclCommInitRank with
# nranks = 2, send/recv between distinct ranks) can be exercised end to end on a single card.
out=gpurun_out/r03t; mkdir -p $out; rm -rf $out/*
export HJ_BENCH_ONE_DEVICE=1 HJ_BENCH_WATCHDOG_S=240 HJ_BENCH_COLLECTIVE_TIMEOUT_S=120 HJ_BENCH_SPINUP=20
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --global-n 129 > $out/two_ranks.json 2> $out/two_ranks.err; echo "rc=$?"
tail -c 1500 $out/two_ranks.json; echo; grep -i "error\|duplicate\|invalid\|bench_slab\|Traceback" $out/two_ranks.err | head -12
