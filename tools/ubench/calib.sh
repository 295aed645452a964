#!/bin/bash
# calibrate FETCH_SIZE / WRITE_SIZE on known byte counts: triad/copy at 8 and 16 B per lane (tools/ubench/bw)
root=$PWD; export TMPDIR=/tmp
# the binaries are built here (not tracked)
for b in bw xcc; do /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $root/tools/ubench/$b $root/tools/ubench/$b.hip || exit 1; done
 d=$root/gpurun_out/calib; rm -rf $d; mkdir -p $d; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d/$c -- $root/tools/ubench/bw > /dev/null 2> $d/$c.err
  python3 - "$d/$c" "$c" <<'PY'
import csv, glob, sys, collections
d, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-14s %-40s mean=%.1f KiB over %d launches" % (c, k[-40:], sum(v)/len(v), len(v)))
PY
done
