#!/bin/bash
# round 5, call 11: kernel trace of the data-dependent-alpha step (bench.py --also RANGE) -- which launches make up its 0.46 ms
root=$PWD; export TMPDIR=/tmp
out=$root/gpurun_out/r05_run11; rm -rf $out; mkdir -p $out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline --no-live-traffic --also RANGE --steps 20 --warmup 5 --repeats 5 > $out/bench.json 2> $out/err.txt
cd $root
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-120s calls %6s avg_us %9.2f  %5.1f %%" % (r["Name"][:120], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
