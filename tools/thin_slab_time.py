#!/usr/bin/env python3
"""Compute-side cost of one rank's slab of the 513^3 grid at N = 2, 4, 8 (virtual rank in the middle of the
decomposition, both neighbours present, exchange replaced by a no-op): ms per RK3 step of the deep-halo schedule (the only one an external transport can serve) against the ideal 1/N of the undivided step.  One GPU; no communication is timed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
steps = 30
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T,
                 n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
dxs = [float(v) for v in np.asarray(g.dx).ravel()]
sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
par = [1.0, 1.0, 1.0, 2.0]
xs0 = torch.linspace(-.75, 3.25, n, device="cuda", dtype=torch.float64)
for world in (1, 2, 4, 8):
    r = world // 2 if world > 1 else 0
    slab = SlabDecomposition(n, world, r, False)
    for deep in (True,):
        st = NativeSlabStepper(g, slab, sid, _ffi.HAM_DUBINS_REL, par, dxs, order=3, deep=deep, external=lambda s: None)
        st.set_alpha_max([st.alpha_local[d] for d in range(3)])
        y = torch.randn((slab.end - slab.begin, n, n), device="cuda", dtype=torch.float64) * 0.01 + \
            (xs0[slab.begin:slab.end, None, None] ** 2).expand(-1, n, n)
        st.set_state(y)
        t = 0.0
        for _ in range(5):
            t, _ = st.step(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            t, _ = st.step(t)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / steps
        cells = (slab.end - slab.begin) * n * n
        print("N=%d rank %d (%d planes) %-12s %.3f ms/step  %.3e cell-substeps/s per rank  frac %.3f" %
              (world, r, slab.end - slab.begin, "deep" if deep else "per-substep", ms, cells * 3 / (ms * 1e-3),
               cells * 64 / (ms * 1e-3) / 8e12), flush=True)
        if hasattr(st, "close"): st.close()
