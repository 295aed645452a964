#!/usr/bin/env python3
"""Mean counter values per dispatch, per kernel, from a rocprofv3 --pmc CSV directory."""
import csv, glob, os, sys, collections
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        k = k.split("(")[0][:70]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in agg.items():
    if "fused" not in k and "direct" not in k and "max_d1" not in k and "seam" not in k:
        continue
    print(k)
    for name, vals in sorted(c.items()):
        print("   %-28s n=%4d mean=%.6g" % (name, len(vals), sum(vals) / len(vals)))
