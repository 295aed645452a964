#!/bin/bash
# round 4, call 12: whole GPU suite on the final build, smoke(), the default bench line
out=gpurun_out/r04_run12; mkdir -p gpurun_out/r04_run12
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee $out/summary.txt
tail -4 $out/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/summary.txt; tail -1 $out/smoke.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?" | tee -a $out/summary.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_run12/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4e frac %.4f of_achievable %.4f iqr %.4f parity %s traffic %.4g (alg %.4g)" % (d["value"], r["frac"], r.get("frac_of_achievable") or 0, d["repeats"]["iqr_over_median"], d.get("parity", {}).get("max_abs_diff"), r.get("traffic") or 0, r["algorithmic_bytes_per_launch"]))
print(d.get("achievable_streaming_rates"))
for k, v in d["also"].items():
    print("%-55s %.4e frac %.4f ms/step %.4f %s" % (k, v.get("value", 0), v.get("roofline_frac", 0), v.get("ms_per_step", 0), {kk: round(vv, 4) for kk, vv in v.items() if kk.startswith("vs_")} or ""), (v.get("roofline_valu") or {}).get("frac", ""), v.get("error", ""))
PY
