#!/bin/bash
# timeline of the range-dependent-alpha step at 201^3 (where do 0.297 ms go?)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_range -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --also RANGE > /tmp/range_bench.json 2> /tmp/range.err
python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/kt_range 0 400 > $GRAFT_REPO_ROOT/gpurun_out/r22_range_timeline.txt 2>&1
tail -1 /tmp/range_bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['also'].items() if isinstance(v,dict)})"
