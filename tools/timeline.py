#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel-trace CSV as a timeline (start/end relative us, queue, kernel)."""
import csv, glob, sys
d = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0; count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][-60:], r.get("Grid_Size", "")))
rows.sort()
rows = rows[-(skip + count):len(rows) - skip] if skip else rows[-count:]
t0 = rows[0][0]
for s, e, q, st, k, gs in rows:
    print("%9.1f %9.1f  dur %7.1f  q=%s s=%s grid=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, gs, k))
