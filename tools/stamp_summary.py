#!/usr/bin/env python3
"""Summarise the phase stamps of an HJ_STAMP build's HJ_TIMING_DUMP file (last launches): per workgroup the
shader-clock time wave 0 and the last wave spent in the four phases of a plane iteration:
  A issue loads + LDS staging writes   B waiting at the barrier   C halo issue + LDS stencil reads + arithmetic + store
  D y0 issue + queue rotation."""
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
for hdr, a in launches[-3:]:
    dur = (a[:, 2] - a[:, 1]) / 100.0
    print(hdr, " wg duration us: p50 %.1f max %.1f" % (np.median(dur), dur.max()))
    for name, off in (("wave 0   ", 5), ("last wave", 9)):
        ph = a[:, off:off + 4]
        tot = ph.sum(axis=1)
        frac = ph / tot[:, None]
        print("   %s cycles/launch p50 %.0f   A %.2f  B(barrier) %.2f  C(compute) %.2f  D %.2f" %
              (name, np.median(tot), *[np.median(frac[:, k]) for k in range(4)]))
