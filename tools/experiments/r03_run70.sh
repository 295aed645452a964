#!/bin/bash
# round 3, run 70: multi-rank rehearsal on the final build: 2 and 4 processes sharing one GPU over gloo (torch halo transport), and the
# same with the native transports requested first (they cannot be set up on one device: every rank fails -> the leg must fall through)
out=gpurun_out/r03br; mkdir -p $out; rm -rf $out/*
export HJ_BENCH_ONE_DEVICE=1 HJ_BENCH_BACKEND=gloo HJ_BENCH_WATCHDOG_S=300 HJ_BENCH_COLLECTIVE_TIMEOUT_S=120 HJ_BENCH_SPINUP=10
for np_ in 2 4; do
  HJ_SLAB_TRANSPORT=torch timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np_ --master-addr 127.0.0.1 --master-port 2953$np_ bench.py --gpus $np_ --steps 5 --warmup 2 --repeats 2 --global-n 129 > $out/ranks$np_.json 2> $out/ranks$np_.err; echo "np=$np_ rc=$?"
  python - $out/ranks$np_.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "error")}, d.get("config", {}).get("parallelism"), d.get("slab_check_max_abs_diff", d.get("config", {}).get("slab_check_max_abs_diff")))
PY
done
# default transport order (native first): on one device the native set-up fails on every rank
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --global-n 129 > $out/fall.json 2> $out/fall.err; echo "fallthrough rc=$?"
tail -c 600 $out/fall.json; echo; grep -i "bench_slab" $out/fall.err | head -6
