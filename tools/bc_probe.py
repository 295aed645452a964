#!/usr/bin/env python3
"""How much of the 201^3 launch is the ghost-cell tail?  Same grid, different boundary conditions."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid
n = int(os.environ.get("N", "201"))
for pd in ([2], [0, 1, 2], [1, 2], [0, 2], None):
    g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T,
                     n * np.ones((3, 1), dtype=np.int64), pd, low_mem=True)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    dg = DeviceGrid(g); dg.bind_stream()
    cur = dg.to_device(d0).clone(); nxt = dg.empty(); w1 = dg.empty()
    tout, dtout = C.c_double(), C.c_double(); parv = _ffi.darr([1., 1., 1., 2.])
    t = 0.0
    def one(cur, nxt, t):
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, 3, 0, parv, t, 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(nxt), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
        return nxt, cur, tout.value
    for _ in range(5): cur, nxt, t = one(cur, nxt, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): cur, nxt, t = one(cur, nxt, t)
    torch.cuda.synchronize(); sec = time.perf_counter() - t0
    print("periodic dims %-10s %.4f ms/substep  %.3e cell-substeps/s" % (pd, 1e3 * sec / 300, n ** 3 * 300 / sec), flush=True)
