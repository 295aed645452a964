#!/usr/bin/env python3
"""Randomised bitwise stress of the pair kernel (double buffer and LDS ring) against the one-cell-per-lane kernel:
random 2-D / 3-D shapes, periodic axes, towardZero flags, schemes, RK orders.  Round 3: the intended WENO5 reduces its
epsilon inside the producing launches at every size here (HJ_EPS_FUSE_MIN_CELLS=0) and a fourth variant runs the
one-cell-per-lane kernel with the two-launch pre-pass in front of every stage (HJ_EPS_FUSE=0).  Round 4: two more variants
march the chunks of a tile column pairwise in opposite directions (HJ_PAIR_DIRS=1, with and without the ring; only -DHJ_MAYDOWN=1 builds honour it).
usage: stress_pair.py [cases] [seed]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    nd = int(rng.integers(2, 4))
    n = tuple(int(v) for v in rng.integers(7, 75 if nd == 3 else 400, size=nd))
    pd = [d for d in range(nd) if rng.random() < 0.4]
    scheme = ["ENO2", "ENO3", "WENO5", "WENO5_ASSHIPPED"][int(rng.integers(0, 4))]
    lo, hi = -np.ones((nd, 1)), np.ones((nd, 1))
    g = L.createGrid(lo, hi, np.array(n, dtype=np.int64).reshape(-1, 1), pd if pd else None)
    ham, par = (_ffi.HAM_DUBINS_REL, [1., 1., 1., 2.]) if nd == 3 else (_ffi.HAM_DOUBLE_INTEGRATOR, [1.5, 0, 0, 0])
    data = L.shapeSphere(g, np.zeros((nd, 1)), .4) + 0.05 * rng.standard_normal(n)
    order = int(rng.integers(1, 4))
    res = {}
    ah = str(int(rng.integers(1, 4)))
    os.environ["HJ_EPS_FUSE_MIN_CELLS"] = "0"
    VARIANTS = ("0", "2", "2r", "0n", "2p", "2rp")
    for flag in VARIANTS:
        os.environ["HJ_PAIR"] = flag[0]
        os.environ["HJ_PAIR_RING"] = "1" if "r" in flag else "0"
        os.environ["HJ_PAIR_DIRS"] = "1" if "p" in flag else "0"
        os.environ["HJ_EPS_FUSE"] = "0" if flag.endswith("n") else "1"
        os.environ["HJ_PAIR_AH"] = ah
        dg = DeviceGrid(g, "float64"); dg.bind_stream()
        y = dg.to_device(data)
        nxt, w0, w1 = dg.empty(), dg.empty(), dg.empty()
        tout, dtout = C.c_double(), C.c_double()
        cur = y
        for _ in range(2):
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, _ffi.SCHEME_IDS[scheme], ham, _ffi.darr(par), 0., 1e9, 0.8, 1e300, 0,
                                         dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, (dg.empty() if cur is y else cur)
        dg.sync()
        res[flag] = (cur.clone(), tout.value)
    ok = all(torch.equal(res["0"][0], res[f][0]) and res["0"][1] == res[f][1] for f in VARIANTS[1:])
    if not ok:
        bad += 1
        print("MISMATCH", n, pd, scheme, order, [float((res["0"][0] - res[f][0]).abs().max()) for f in VARIANTS[1:]], flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
