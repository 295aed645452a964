#!/usr/bin/env python3
"""Summarise the stamps of an HJ_STAMP build's HJ_TIMING_DUMP file (last launches).  Per workgroup: wall-clock
(100 MHz) start / loop start / loop end / end, the loop's shader-cycle count (-> effective shader clock), and the
share of wave 0's loop time in the phases of a plane iteration:
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
    t0 = a[:, 1].min()
    st, en = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0
    l0, l1 = (a[:, 10] - t0) / 100.0, (a[:, 11] - t0) / 100.0
    cyc = a[:, 12]
    print(hdr)
    print("   launch: first start 0, last end %.1f us;  per workgroup (p50): prologue %.1f us, loop %.1f us, epilogue %.1f us"
          % (en.max(), np.median(l0 - st), np.median(l1 - l0), np.median(en - l1)))
    print("   shader clock inside the loop: %.2f GHz (p50)" % np.median(cyc / ((l1 - l0) * 1e3)))
    ph = a[:, 5:9]
    frac = ph / ph.sum(axis=1)[:, None]
    print("   wave 0 loop phases: A %.2f  B(barrier) %.2f  C(compute) %.2f  D %.2f" % tuple(np.median(frac[:, k]) for k in range(4)))
    print("   start p10/p50/p90: %.1f %.1f %.1f   end p10/p50/p90: %.1f %.1f %.1f" %
          (*np.percentile(st, [10, 50, 90]), *np.percentile(en, [10, 50, 90])))
