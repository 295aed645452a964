#!/bin/bash
# round 4, call 5: stream gating probe (hipStreamWaitValue64 on a value a running kernel publishes; CU-masked stream), the
# NumPy-in loop after the np.ndim fix, round-4 tests
out=gpurun_out/r04_run5; mkdir -p $out; rm -f $out/*
timeout -k 5 60 tools/probes/stream_gate_probe > $out/stream_gate_probe.txt 2>&1; echo "probe rc=$?"; cat $out/stream_gate_probe.txt
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "numpy or convection" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --also API > $out/bench_api.json 2> $out/bench_api.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_run5/bench_api.json").read().strip().splitlines()[-1])
print("value %.4e" % d["value"])
for k, v in d["also"].items():
    print("%-55s %.4e ms/step %.4f vs_raw %s vs_tensor %s" % (k, v.get("value", 0), v.get("ms_per_step", 0), v.get("vs_raw_c_loop", ""), v.get("vs_tensor_in", "")))
PY
