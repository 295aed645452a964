#!/bin/bash
# round 3, run 10: whole GPU suite + the driver's bench command on the current tree
out=gpurun_out/r03j; mkdir -p $out; rm -f $out/*
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/test_gpu.txt 2>&1; echo "rc=$?" >> $out/test_gpu.txt; tail -4 $out/test_gpu.txt
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03j/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline")}, indent=1)[:3000])
for k, v in d.get("also", {}).items():
    print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "roofline_frac", "kernel", "roofline_valu", "error")})
PY
