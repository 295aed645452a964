#!/usr/bin/env python3
"""Summarise an HJ_TIMING_DUMP file: per launch, when workgroups start and end (us, 100 MHz clock)."""
import sys
import numpy as np
launches, cur, hdr = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur:
            launches.append((hdr, np.array(cur, dtype=np.float64)))
        hdr, cur = line.strip(), []
    else:
        cur.append([float(x) for x in line.split()])
if cur:
    launches.append((hdr, np.array(cur, dtype=np.float64)))
for hdr, a in launches[int(sys.argv[2]) if len(sys.argv) > 2 else -3:]:
    t0 = a[:, 1].min()
    st, en = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0
    dur = en - st
    print(hdr)
    print("  start  us: min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
    print("  end    us: min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (en.min(), np.median(en), np.percentile(en, 90), en.max()))
    print("  dur    us: min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
    for x in range(8):
        m = a[:, 3] == x
        if m.any():
            print("  xcc %d: n=%3d  dur p50 %.1f max %.1f  end max %.1f" % (x, m.sum(), np.median(dur[m]), dur[m].max(), en[m].max()))
    chunks = np.unique(a[:, 4])
    print("  by chunk (dur p50):", " ".join("%d:%.1f" % (c, np.median(dur[a[:, 4] == c])) for c in chunks))
    order = np.argsort(-dur)[:8]
    print("  slowest blocks (logical id, chunk, tile, dur):", [(int(a[i, 0]), int(a[i, 4]), int(a[i, 0]) % int(hdr.split("ntiles=")[1].split()[0]), round(dur[i], 1)) for i in order])
if len(sys.argv) > 3:   # tile map of the last launch: rows = tile index on axis 1, cols = axis 2 (mean over chunks)
    hdr, a = launches[-1]
    nt = int(hdr.split("ntiles=")[1].split()[0]); n2 = int(sys.argv[3])
    dur = (a[:, 2] - a[:, 1]) / 100.0
    tile = a[:, 0].astype(int) % nt
    print("  tile map (mean dur us), rows = axis-1 tile, cols = axis-2 tile")
    for t1 in range(nt // n2):
        print("   ", " ".join("%5.1f" % dur[tile == t1 * n2 + t2].mean() for t2 in range(n2)))
    ch = a[:, 4].astype(int)
    print("  chunk x axis-1 tile row (mean dur)")
    for c in np.unique(ch):
        print("   ", c, " ".join("%5.1f" % dur[(ch == c) & (tile // n2 == t1)].mean() for t1 in range(nt // n2)))
