#!/bin/bash
# round 3, run 19: whole GPU suite (tiled = direct now asserted bitwise) + the driver's bench command
out=gpurun_out/r03s; mkdir -p $out; rm -rf $out/*
timeout -k 10 1500 python -m pytest tests -q -m gpu > $out/test_gpu.txt 2>&1; echo "rc=$?" >> $out/test_gpu.txt; tail -12 $out/test_gpu.txt
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03s/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_hip_events"], d["roofline"]["kernel_ms_rocprof"])
for k, v in d.get("also", {}).items():
    print(k, v.get("value"), v.get("roofline_frac"), v.get("kernel"), (v.get("roofline_valu") or {}).get("frac"))
PY
