#!/bin/bash
# round 4, call 27: the self-launching bench with FOUR ranks on one card (gloo + torch transport: RCCL refuses several ranks per device), C4 at 257^3
# and C5 at 41^4, and the fall-through from the native transport (which fails on every rank here) to the torch one
out=gpurun_out/r04_run27; mkdir -p $out
export HJ_BENCH_ONE_DEVICE=1 HJ_BENCH_BACKEND=gloo
HJ_SLAB_TRANSPORT=torch timeout -k 10 400 python3 bench.py --gpus 4 --global-n 257 --steps 6 --warmup 2 --repeats 3 > $out/c4_4rank.json 2> $out/c4_4rank.err; echo "C4 4 ranks rc=$?"; cut -c1-600 $out/c4_4rank.json
HJ_SLAB_TRANSPORT=torch timeout -k 10 400 python3 bench.py --gpus 4 --workload C5 --global-n 41 --steps 4 --warmup 1 --repeats 3 > $out/c5_4rank.json 2> $out/c5_4rank.err; echo "C5 4 ranks rc=$?"; cut -c1-600 $out/c5_4rank.json
timeout -k 10 400 python3 bench.py --gpus 3 --global-n 129 --steps 4 --warmup 1 --repeats 3 > $out/c4_3rank_fallthrough.json 2> $out/c4_3rank_fallthrough.err; echo "C4 3 ranks, transports in order rc=$?"; cut -c1-400 $out/c4_3rank_fallthrough.json
