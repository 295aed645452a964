#!/usr/bin/env python3
"""Phase clocks of the stage-fused pair kernel (-DHJ_F12_STAMP build, HJ_TIMING_DUMP=file): per wave, shader cycles of
the six phases of a plane iteration summed over the loop.  usage: f12_stamps.py dumpfile"""
import sys
import numpy as np
names = ["0 wait H values + stage y", "1 barrier", "2 stores + loads + ghost fix", "3 stage 1 (A slots)", "4 stage 2 (interior slots)", "5 wait own + rotate"]
launches, cur, hdr = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur and hdr and hdr.startswith("# fused12_pair"):
            launches.append((hdr, np.array([r for r in cur if len(r) == 12], dtype=np.float64)))
        hdr, cur = line.strip(), []
    else:
        cur.append([float(x) for x in line.split()])
if cur and hdr and hdr.startswith("# fused12_pair"):
    launches.append((hdr, np.array([r for r in cur if len(r) == 12], dtype=np.float64)))
hdr, a = launches[-1]
print(hdr)
a = a[a[:, 8] > 0]
ph, cyc, wall, iters, ns2 = a[:, 2:8], a[:, 8], a[:, 9], a[:, 10], a[:, 11]
print("waves: %d   iterations per workgroup (p50): %d   loop: %.0f cycles per iteration (p50), shader clock %.2f GHz (p50)"
      % (len(a), np.median(iters), np.median(cyc / iters), np.median(cyc / (wall * 10.0)) / 1e3))
for sel, tag in ((ns2 >= 2, "waves with two stage-2 slots"), (ns2 == 1, "waves with one stage-2 slot"), (ns2 >= 0, "all waves")):
    if not sel.any():
        continue
    p = ph[sel] / iters[sel][:, None]
    tot = cyc[sel] / iters[sel]
    print("%s (%d):" % (tag, sel.sum()))
    for k in range(6):
        print("   phase %-30s %7.0f cycles per iteration (p50)  %5.1f %%" % (names[k], np.median(p[:, k]), 100 * np.median(p[:, k] / tot)))
