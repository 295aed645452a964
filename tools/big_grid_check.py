#!/usr/bin/env python3
"""A grid that only a 288 GB card holds: n^3 fp64 (default 1025^3 = 8.6 GB per array, four arrays in flight), one odeCFL3 step of the
Dubins problem through the tiled kernels against the direct kernel (bitwise), plus the throughput of the tiled path.
Initial data are formed on the device (no host array of that size is ever made).
usage: big_grid_check.py [n] [scheme]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
scheme = sys.argv[2] if len(sys.argv) > 2 else "WENO5_ASSHIPPED"
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
x0 = torch.linspace(-.75, 3.25, n, dtype=torch.float64, device="cuda").reshape(n, 1, 1)
x1 = torch.linspace(-1.25, 1.25, n, dtype=torch.float64, device="cuda").reshape(1, n, 1)
x2 = torch.linspace(-np.pi, np.pi * (1 - 2 / n), n, dtype=torch.float64, device="cuda").reshape(1, 1, n)
y0 = (torch.sqrt(x0 * x0 + x1 * x1) - 0.5 + 0.05 * torch.sin(3 * x2) * torch.cos(2 * x0)).contiguous()
print("grid %d^3: %.2f GB per array, %.1f GB allocated" % (n, y0.numel() * 8 / 1e9, torch.cuda.memory_allocated() / 1e9), flush=True)
par, sid = _ffi.darr([1., 1., 1., 2.]), _ffi.SCHEME_IDS[scheme]
outs = {}
for tag, env in (("tiled", {}), ("direct", {"HJ_FORCE_DIRECT": "1"})):
    for k in ("HJ_FORCE_DIRECT",):
        os.environ.pop(k, None)
    os.environ.update(env)
    dg = DeviceGrid(g, "float64"); dg.bind_stream()
    nxt, w0, w1 = torch.empty_like(y0), torch.empty_like(y0), torch.empty_like(y0)
    tout, dtout = C.c_double(), C.c_double()
    reps = 6 if tag == "tiled" else 1
    for r in range(reps):
        if r == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, par, 0., 1e9, 0.8, 1e300, 0, dg.ptr(y0), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1),
                                     C.byref(tout), C.byref(dtout)))
    torch.cuda.synchronize()
    if tag == "tiled":
        ms = 1e3 * (time.perf_counter() - t0) / (reps - 1)
        ext = (C.c_int * 4)(); dg.lib.hj_last_tile(dg.ctx, ext)
        print("tiled: %s, tile (%d planes, %d x %d), %.2f ms per RK3 step = %.3e cell-substeps/s = %.3f of 8 TB/s" %
              (dg.lib.hj_last_kernel(dg.ctx).decode(), ext[0], ext[1], ext[2], ms, 3 * n ** 3 / (ms * 1e-3), 3 * n ** 3 * 64 / 3 / (ms * 1e-3) / 8e12), flush=True)
    assert bool(torch.isfinite(nxt).all())
    outs[tag] = (nxt, float(tout.value), float(dtout.value))
    del w0, w1
    torch.cuda.empty_cache()
same = torch.equal(outs["tiled"][0], outs["direct"][0])
print("tiled == direct kernel bitwise: %s; t %r / %r; max |y1 - y0| = %.3e; peak memory %.1f GB" %
      (same, outs["tiled"][1], outs["direct"][1], float((outs["tiled"][0] - y0).abs().max()), torch.cuda.max_memory_allocated() / 1e9))
sys.exit(0 if same and outs["tiled"][1:] == outs["direct"][1:] else 1)
