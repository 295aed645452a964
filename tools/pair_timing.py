#!/usr/bin/env python3
"""Per-workgroup timing of the pair kernel from an HJ_TIMING_DUMP file (columns: logical id, start, end, xcc, chunk,
ph0..ph3, hw_id, loop start, loop end, -): where do the workgroups of one launch spend their time, and which finish late."""
import sys
import numpy as np
launches, cur, hdr = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur: launches.append((hdr, np.array(cur, dtype=np.float64)))
        hdr, cur = line.strip(), []
    else:
        cur.append([float(x) for x in line.split()])
if cur: launches.append((hdr, np.array(cur, dtype=np.float64)))
sel = launches[int(sys.argv[2]):] if len(sys.argv) > 2 else launches[-3:]
for hdr, a in sel:
    nt = int(hdr.split("ntiles=")[1].split()[0])
    t0 = a[:, 1].min()
    st, en = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0
    l0, l1 = (a[:, 10] - t0) / 100.0, (a[:, 11] - t0) / 100.0
    dur = en - st
    print(hdr)
    pc = lambda x: "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (x.min(), *np.percentile(x, [10, 50, 90]), x.max())
    print("  start:", pc(st)); print("  end  :", pc(en)); print("  prolog:", pc(l0 - st)); print("  loop :", pc(l1 - l0)); print("  epilog:", pc(en - l1))
    if a[:, 5].max() > 0:
        ph = [(a[:, k] - t0) / 100.0 for k in (5, 6, 7, 8)]
        two = (a[:, 12] - t0) / 100.0
        print("  prologue stamps p50 (us after wg start): queue loads issued %.2f | cell consts %.2f | halo index + syncthreads_or %.2f | halo loads issued %.2f | loop start %.2f | first 2 planes done %.2f"
              % tuple(np.median(x - st) for x in (ph[0], ph[1], ph[2], ph[3], l0, two)))
    hw = a[:, 9].astype(np.int64)
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    for x in range(8):
        m = a[:, 3] == x
        if m.any(): print("  xcc %d: n=%3d dur p50 %.1f max %.1f | loop p50 %.1f | distinct (se,sh,cu) %d" % (x, m.sum(), np.median(dur[m]), dur[m].max(), np.median((l1 - l0)[m]), len(set(zip(se[m], sh[m], cu[m])))))
    ch = a[:, 4].astype(int)
    print("  loop p50 by chunk:", " ".join("%d:%.1f" % (c, np.median((l1 - l0)[ch == c])) for c in np.unique(ch)))
    tile = a[:, 0].astype(int) % nt
    print("  loop p50 by tile :", " ".join("%.1f" % np.median((l1 - l0)[tile == t]) for t in range(nt)))
    key = list(zip(a[:, 3].astype(int), se, sh, cu))
    from collections import Counter
    cnt = Counter(key)
    print("  workgroups per CU: ", sorted(Counter(cnt.values()).items()))
