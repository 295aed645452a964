#!/bin/bash
out=gpurun_out/r02i; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for nb in 320 640 1280; do
HJ_EPS_BLOCKS=$nb rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats$nb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --scheme WENO5 --steps 20 --repeats 2 > $GRAFT_REPO_ROOT/$out/weno5_$nb.json 2> $GRAFT_REPO_ROOT/$out/weno5.err
f=$(find $GRAFT_REPO_ROOT/$out/stats$nb -name "*kernel_stats.csv" | head -1)
echo "== $nb"; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
