"""round 6: axis-0 march against the transposed march on thin 3-D grids of several shapes (no neighbours), with the two launch plans beside
the times -- the data the auto rule (hj_inst.hip, launch_scheme) is calibrated on.  argv: shapes as n0xn1xn2 ..."""
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi, dist
from levelsetpy_amd.context import DeviceGrid
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(20, 640, 520)]
sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
for n in shapes:
    g = L.createGrid(np.array([[-2., -1.25, -np.pi]]).T, np.array([[2., 1.25, np.pi * (1 - 2 / n[2])]]).T, np.array(n, dtype=np.int64).reshape(-1, 1), 2, low_mem=True)
    x0 = torch.as_tensor(np.asarray(g.vs[0]).ravel(), device="cuda").reshape(-1, 1, 1)
    x1 = torch.as_tensor(np.asarray(g.vs[1]).ravel(), device="cuda").reshape(1, -1, 1)
    d0 = ((x0 * x0 + x1 * x1).sqrt() - 0.5).expand(*n).contiguous()
    res, plans = {}, {}
    for xp in ("0", "2", "1"):
        os.environ["HJ_XP"] = xp
        p = dist.plan_substep(n, [0, 0, 1], "float64", sid, _ffi.HAM_DUBINS_REL, 1, 0, n[0])
        plans[xp] = "%d wg x (%d + 6) tile %s" % (p["workgroups"], p["chunk_planes"], "x".join(str(e) for e in p["tile"][1:] if e) if len(p["tile"]) > 2 else p["tile"])
        dg = DeviceGrid(g, "float64"); dg.bind_stream()
        cur, nxt, w1 = d0.clone(), torch.empty_like(d0), torch.empty_like(d0)
        tout, dtout = C.c_double(), C.c_double()
        par = _ffi.darr([1., 1., 1., 2.])
        t = [0.]
        def one():
            global cur, nxt
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, par, t[0], 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(nxt), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur; t[0] = float(tout.value)
        for _ in range(30 if xp != "1" else 120): one()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): one()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
        res[xp] = (best, dg.lib.hj_last_kernel(dg.ctx).decode())
    print("%-14s axis-0 %.4f ms [%s]   transposed %.4f ms [%s]  (%+.1f %%)   auto %.4f ms -> %s" % (
        "x".join(map(str, n)), res["0"][0] * 1e3, plans["0"], res["2"][0] * 1e3, plans["2"], 100 * (res["0"][0] / res["2"][0] - 1), res["1"][0] * 1e3,
        "transposed" if "axis 1" in res["1"][1] else "axis-0"), flush=True)
