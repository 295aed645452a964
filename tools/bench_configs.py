#!/usr/bin/env python3
"""Throughput of the other BASELINE configs on one GPU (parity-test cases, not the headline):
C3 double integrator 4096^2 ENO3, C4 Dubins 513^3 (single GPU), C5 double pendulum 129^4 fp32."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid

def run(name, g, ham, par, scheme, dtype, d0, steps=10, warmup=2):
    dg = DeviceGrid(g, dtype); dg.bind_stream()
    cur = dg.to_device(d0).clone(); nxt = dg.empty(); w1 = dg.empty()
    tout, dtout = C.c_double(), C.c_double(); parv = _ffi.darr(par + [0, 0, 0])
    t = 0.0
    def one(cur, nxt, t):
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, _ffi.SCHEME_IDS[scheme], ham, parv, t, 1e9, 0.8, 1e300, 0,
                                     dg.ptr(cur), dg.ptr(nxt), dg.ptr(nxt), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
        return nxt, cur, tout.value
    for _ in range(warmup): cur, nxt, t = one(cur, nxt, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): cur, nxt, t = one(cur, nxt, t)
    torch.cuda.synchronize(); sec = time.perf_counter() - t0
    ok = bool(torch.isfinite(cur).all())
    n = dg.numel; bps = (64 if dtype == "float64" else 32) / 3
    v = n * 3 * steps / sec
    print("%-44s %s %-16s %.3e cell-substeps/s  %.3f ms/step  frac_of_8TBps=%.3f finite=%s" %
          (name, dtype, scheme, v, 1e3 * sec / steps, v * bps / 8e12, ok), flush=True)

which = sys.argv[1:] or ["c3", "c4", "c5"]
if "c3" in which:
    n = 4096
    g = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), n * np.ones((2, 1), dtype=np.int64), None, low_mem=True)
    d0 = L.shapeSphere(g, np.zeros((2, 1)), .25)
    for sch in ("ENO3", "WENO5_ASSHIPPED"):
        run("C3 double integrator 4096^2", g, _ffi.HAM_DOUBLE_INTEGRATOR, [1.0], sch, "float64", d0)
if "c4" in which:
    n = 513
    g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T,
                     n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    for sch in ("WENO5_ASSHIPPED", "WENO5"):
        run("C4 Dubins 513^3 (one GPU)", g, _ffi.HAM_DUBINS_REL, [1.0, 1.0, 1.0, 2.0], sch, "float64", d0, steps=5)
if "c5" in which:
    n = int(os.environ.get("C5_N", "129"))
    gmin = np.array([[-np.pi, -8, -np.pi, -8]]).T
    gmax = np.array([[np.pi * (1 - 2 / n), 8 * (1 - 2 / n), np.pi * (1 - 2 / n), 8 * (1 - 2 / n)]]).T
    g = L.createGrid(gmin, gmax, n * np.ones((4, 1), dtype=np.int64), [0, 1, 2, 3], low_mem=True)
    d0 = L.shapeSphere(g, np.zeros((4, 1)), .5).astype(np.float32)
    run("C5 double pendulum %d^4 (one GPU)" % n, g, _ffi.HAM_DOUBLE_PENDULUM, [1.0], "WENO5_ASSHIPPED",
        "float32", torch.as_tensor(d0), steps=int(os.environ.get("C5_STEPS", "3")), warmup=int(os.environ.get("C5_WARMUP", "1")))
