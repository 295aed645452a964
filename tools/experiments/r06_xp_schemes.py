"""round 6: the transposed march against the axis-0 march on a thin grid (65 x 513 x 513, no ring), per scheme and dtype -- does the auto rule pay everywhere?"""
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid
n = [65, 513, 513]
g = L.createGrid(np.array([[-2., -1.25, -np.pi]]).T, np.array([[2., 1.25, np.pi * (1 - 2 / n[2])]]).T, np.array(n, dtype=np.int64).reshape(-1, 1), 2, low_mem=True)
x0 = torch.as_tensor(np.asarray(g.vs[0]).ravel(), device="cuda").reshape(-1, 1, 1)
x1 = torch.as_tensor(np.asarray(g.vs[1]).ravel(), device="cuda").reshape(1, -1, 1)
d64 = ((x0 * x0 + x1 * x1).sqrt() - 0.5).expand(*n).contiguous()
for dtype in ("float64", "float32"):
    d0 = d64.to(getattr(torch, dtype))
    for scheme in ("WENO5_ASSHIPPED", "ENO2", "ENO3"):
        res = {}
        for xp in ("0", "2"):
            os.environ["HJ_XP"] = xp
            dg = DeviceGrid(g, dtype); dg.bind_stream()
            cur, nxt, w1 = d0.clone(), torch.empty_like(d0), torch.empty_like(d0)
            tout, dtout = C.c_double(), C.c_double()
            par = _ffi.darr([1., 1., 1., 2.]); sid = _ffi.SCHEME_IDS[scheme]
            t = [0.]
            def one():
                global cur, nxt
                _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, par, t[0], 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(nxt), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
                cur, nxt = nxt, cur; t[0] = float(tout.value)
            for _ in range(30): one()
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): one()
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
            res[xp] = (best, dg.lib.hj_last_kernel(dg.ctx).decode())
        print("%-8s %-16s axis-0 %.4f ms   transposed %.4f ms   (%+.1f %%)   %s" % (dtype, scheme, res["0"][0] * 1e3, res["2"][0] * 1e3, 100 * (res["0"][0] / res["2"][0] - 1), res["2"][1]), flush=True)
