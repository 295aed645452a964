#!/bin/bash
# round 4, call 9: run-time (hipRTC) Hamiltonians -- GPU tests, then the bench legs (run-time kernel vs built-in vs split path)
out=gpurun_out/r04_run9; mkdir -p gpurun_out/r04_run9
timeout -k 10 600 python -m pytest tests/test_gpu_user_ham.py -x -q -m gpu > $out/pytest_user.log 2>&1; echo "pytest user ham rc=$?"; tail -15 $out/pytest_user.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --also RTC > $out/bench_rtc.json 2> $out/bench_rtc.err; echo "bench rc=$?"; tail -3 $out/bench_rtc.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_run9/bench_rtc.json").read().strip().splitlines()[-1])
print("headline %.4e frac %.4f" % (d["value"], d["roofline"]["frac"]))
for k, v in d["also"].items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "roofline_frac", "vs_builtin", "vs_run_time_hamiltonian", "kernel", "leg_wall_s_incl_compile", "error")})
PY
