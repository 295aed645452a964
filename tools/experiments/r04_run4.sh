#!/bin/bash
# round 4, call 4: whole GPU suite on the new build (prologue rework, plan / stream / tiling caches, HostView results for NumPy
# callers, HJIPDE_solve on a device tensor), then the default bench line
out=gpurun_out/r04_run4; mkdir -p $out; rm -f $out/*
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $out/summary.txt
tail -6 $out/pytest_gpu.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee -a $out/summary.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_run4/bench_default.json").read().strip().splitlines()[-1])
print("value %.4e frac %.4f of_achievable %s iqr %.4f parity %s" % (d["value"], d["roofline"]["frac"], d["roofline"].get("frac_of_achievable"), d["repeats"]["iqr_over_median"], d.get("parity", {}).get("max_abs_diff")))
print(d.get("achievable_streaming_rates"))
for k, v in d["also"].items():
    print("%-55s %.4e frac %.4f ms/step %.4f %s %s" % (k, v.get("value", 0), v.get("roofline_frac", 0), v.get("ms_per_step", 0), v.get("vs_raw_c_loop", ""), v.get("vs_tensor_in", "")))
PY
