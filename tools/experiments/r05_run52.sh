#!/bin/bash
# late round 5: which of the library's kernel instantiations does the GPU suite actually launch?  The whole suite under rocprofv3 --kernel-trace --stats;
# only the (name, calls) table travels back (the traces are deleted on the box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf /tmp/r52; ( while true; do date >> gpurun_out/r52_heartbeat.txt; sleep 45; done ) & HB=$!
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r52 -o suite -- python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r52_tests.log 2>&1; echo "suite under rocprofv3 rc=$? $(grep -E "passed|failed" gpurun_out/r52_tests.log | tail -1)"
python3 - <<'PY'
import csv, glob, collections
calls = collections.Counter()
files = glob.glob("/tmp/r52/**/*kernel_stats.csv", recursive=True)
for f in files:
    for row in csv.DictReader(open(f)):
        calls[row["Name"]] += int(row["Calls"])
with open("gpurun_out/r52_kernels_launched.txt", "w") as o:
    for k, v in sorted(calls.items()):
        o.write("%d\t%s\n" % (v, k))
print("%d stats files, %d distinct kernel names, %d launches" % (len(files), len(calls), sum(calls.values())))
PY
rm -rf /tmp/r52; kill $HB
