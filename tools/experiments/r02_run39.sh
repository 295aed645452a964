#!/bin/bash
# phase stamps of the pair kernel's plane loop (HJ_STAMP build): A = load issue + LDS staging, B = barrier, C = halo issue +
# stencil reads + arithmetic + store, D = y0 issue + queue rotation
out=gpurun_out/r02am; mkdir -p $out; rm -f $out/*
export HJ_LIB=$PWD/levelsetpy_amd/csrc/libhj_stamp.so
for n in 201 513; do
HJ_TIMING_DUMP=$out/t$n.txt timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --n $n --steps 3 --warmup 3 --repeats 1 > /dev/null 2> $out/b$n.err
python - $out/t$n.txt <<'PY'
import sys, numpy as np
launches, cur, hdr = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur: launches.append((hdr, np.array(cur, dtype=np.float64)))
        hdr, cur = line.strip(), []
    else: cur.append([float(x) for x in line.split()])
if cur: launches.append((hdr, np.array(cur, dtype=np.float64)))
for hdr, a in launches[-3:]:
    ph = a[:, 5:9]; fr = ph / ph.sum(axis=1)[:, None]
    t0 = a[:, 1].min(); l0, l1 = (a[:, 10] - t0) / 100.0, (a[:, 11] - t0) / 100.0
    print(hdr, "| loop p50 %.1f us | wave-0 phases: A %.2f B(barrier) %.2f C(compute) %.2f D %.2f | cycles/loop-us %.0f" %
          (np.median(l1 - l0), *[np.median(fr[:, k]) for k in range(4)], np.median(ph.sum(axis=1) / ((l1 - l0)))))
PY
done
