#!/bin/bash
# range-dependent alpha after the round-5 host-side rework: parity tests, the bench leg, a timeline
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -q -m gpu -x -k "range or rtc or runtime or dynamic or alpha or hamiltonian or native" > gpurun_out/r23_tests.log 2>&1 || { tail -30 gpurun_out/r23_tests.log; exit 1; }
tail -2 gpurun_out/r23_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --also RANGE,RTC > gpurun_out/r23_bench.json 2> gpurun_out/r23_bench.err
tail -1 gpurun_out/r23_bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:(v.get('value'),v.get('ms_per_step'),v.get('vs_split_path')) for k,v in d['also'].items() if isinstance(v,dict)})"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_range -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --also RANGE > /tmp/range_bench.json 2> /tmp/range.err
python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/kt_range 0 60 > $GRAFT_REPO_ROOT/gpurun_out/r23_range_timeline.txt 2>&1
