#!/bin/bash
# round 3, run 21: the multi-rank slab leg of bench.py as 2 and 4 PROCESSES sharing one GPU (torch.distributed over gloo, the
# 'torch' halo transport on the real HIP kernels): orchestration, self check against the single domain, timing, JSON
out=gpurun_out/r03u; mkdir -p $out; rm -rf $out/*
export HJ_BENCH_ONE_DEVICE=1 HJ_BENCH_BACKEND=gloo HJ_SLAB_TRANSPORT=torch HJ_BENCH_WATCHDOG_S=300 HJ_BENCH_COLLECTIVE_TIMEOUT_S=120 HJ_BENCH_SPINUP=10
for np_ in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np_ --master-addr 127.0.0.1 --master-port 2953$np_ bench.py --gpus $np_ --steps 5 --warmup 2 --repeats 2 --global-n 129 > $out/ranks$np_.json 2> $out/ranks$np_.err; echo "np=$np_ rc=$?"
  tail -c 1800 $out/ranks$np_.json; echo; grep -i "error\|bench_slab\|Traceback" $out/ranks$np_.err | head -8
done
