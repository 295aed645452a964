#!/bin/bash
# thin slabs: the ceiling (same grid without the ring) and kernel timelines of the deep stepper
mkdir -p gpurun_out
out=gpurun_out/r18_thin.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python tools/thin_slab_ring.py 513 8 plain 2>&1 | grep "^N=" >> $out || echo "FAILED" >> $out; }
run HJ_X=0
run HJ_MIN_CHUNK=80
run HJ_MIN_CHUNK=30
run HJ_MIN_CHUNK=16
run HJ_MIN_CHUNK=12
cd /tmp && export TMPDIR=/tmp
for mc in 4 80; do
  HJ_MIN_CHUNK=$mc timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$mc -- python3 $GRAFT_REPO_ROOT/tools/thin_slab_ring.py 513 8 deep > /tmp/kt$mc.log 2>&1
  f=$(find /tmp/kt$mc -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $mc >> $GRAFT_REPO_ROOT/$out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-60]["Start_Timestamp"])
print("== deep stepper timeline, HJ_MIN_CHUNK=%s (us; last 60 launches)" % sys.argv[2])
for r in rows[-60:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %9.1f dur %7.1f q=%s grid=%s wg=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r["Kernel_Name"][:70]))
PY
done
cat $GRAFT_REPO_ROOT/$out
