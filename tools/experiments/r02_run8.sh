#!/bin/bash
out=gpurun_out/r02h; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/test.log 2>&1; echo "pytest rc=$?"; tail -3 $out/test.log
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --scheme WENO5 --steps 20 --repeats 2 > $GRAFT_REPO_ROOT/$out/weno5.json 2> $GRAFT_REPO_ROOT/$out/weno5.err
cd $GRAFT_REPO_ROOT
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_weno5.csv
cut -c1-100,250-400 $out/kernel_stats_weno5.csv | head -5
python -c "
import json; d=json.load(open('$out/weno5.json')); print(d['value'], d['roofline']['frac'], d['ms_per_step'])"
