#!/bin/bash
# round 6: the bench's slab leg with its new diagnostics (one-GPU rehearsals), then the default bench line
mkdir -p gpurun_out
HJ_BENCH_SPINUP=40 timeout -k 10 300 python bench.py --gpus 1 --workload C5 --global-n 65 --steps 10 --warmup 2 --repeats 5 > gpurun_out/r06_slab_c5_65.json 2> gpurun_out/r06_slab_c5_65.err || { tail -20 gpurun_out/r06_slab_c5_65.err; exit 1; }
HJ_BENCH_SPINUP=40 HJ_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --gpus 1 --global-n 257 --steps 10 --warmup 2 --repeats 5 > gpurun_out/r06_slab_c4_257.json 2> gpurun_out/r06_slab_c4_257.err || { tail -20 gpurun_out/r06_slab_c4_257.err; exit 1; }
timeout -k 10 900 python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err || { tail -20 gpurun_out/r06_bench_a.err; exit 1; }
python - <<'PY'
import json
for f in ("gpurun_out/r06_slab_c5_65.json", "gpurun_out/r06_slab_c4_257.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], json.dumps(d["also"])[:1500])
d = json.loads(open("gpurun_out/r06_bench_a.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["frac_from_value"])
print("\n".join(d["summary"]))
k = [k for k in d["also"] if "CFL" in k][0]
print(k, {kk: vv for kk, vv in d["also"][k].items() if kk != "repeats"})
PY
