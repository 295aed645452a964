#!/bin/bash
# late round 5: 2-rank rehearsal of bench.py on ONE card over gloo (the torch halo transport), a 129^3 global grid -- after the last dist.py changes
out=gpurun_out/r40; mkdir -p $out; rm -rf $out/*
export HJ_BENCH_ONE_DEVICE=1 HJ_BENCH_BACKEND=gloo HJ_BENCH_WATCHDOG_S=300 HJ_BENCH_COLLECTIVE_TIMEOUT_S=120 HJ_BENCH_SPINUP=10
HJ_SLAB_TRANSPORT=torch timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --global-n 129 > $out/ranks2.json 2> $out/ranks2.err; echo "launcher np=2 rc=$?"
HJ_SLAB_TRANSPORT=torch timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --global-n 129 > $out/self2.json 2> $out/self2.err; echo "self-launched np=2 rc=$?"
python - $out/ranks2.json $out/self2.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "error")}, d.get("config", {}).get("parallelism"))
PY
