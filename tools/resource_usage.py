#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) one line per kernel."""
import re, sys, subprocess
txt = sys.stdin.read()
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "LDS Size \\[bytes/block\\]", "TotalSGPRs"):
        m = re.search(r"remark:\s+%s: (\d+)" % key, line)
        if m and cur is not None:
            cur[key.split(" ")[0].replace("\\", "")] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("hj::", "").replace("void ", "")
    if len(sys.argv) > 1 and not re.search(sys.argv[1], n):
        continue
    print("%-100s vgpr=%3d sgpr=%3d scratch=%4d occ=%d" % (n[:100], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1)))
