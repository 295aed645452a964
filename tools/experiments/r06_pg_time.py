import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid
n = [72, 72, 72, 129]
for pd in ([0, 1, 2, 3], [0, 2], None):
    pdl = pd or []
    gmin = np.array([[-np.pi, -8, -np.pi, -8]]).T
    gmax = np.array([[np.pi * (1 - 2 / n[0]) if 0 in pdl else np.pi, 8 * (1 - 2 / n[1]) if 1 in pdl else 8., np.pi * (1 - 2 / n[2]) if 2 in pdl else np.pi, 8 * (1 - 2 / n[3]) if 3 in pdl else 8.]]).T
    g = L.createGrid(gmin, gmax, np.array(n, dtype=np.int64).reshape(-1, 1), pd, low_mem=True)
    xs = [torch.as_tensor(np.asarray(v).ravel(), device="cuda", dtype=torch.float32) for v in g.vs]
    shp = lambda d: [-1 if k == d else 1 for k in range(4)]
    d0 = (sum((xs[d] ** 2).reshape(shp(d)) for d in range(4)).sqrt() - 0.5).contiguous()
    for flat in ("1", "0"):
        os.environ["HJ_FLAT4"] = flat
        dg = DeviceGrid(g, "float32"); dg.bind_stream()
        cur, nxt, w1 = d0.clone(), torch.empty_like(d0), torch.empty_like(d0)
        tout, dtout = C.c_double(), C.c_double()
        par = _ffi.darr([1., 0., 0., 0.]); sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
        t = [0.]
        def one():
            global cur, nxt
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DOUBLE_PENDULUM, par, t[0], 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(nxt), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur; t[0] = float(tout.value)
        for _ in range(10): one()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): one()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        cells = float(np.prod(n))
        print("pd=%s HJ_FLAT4=%s  %.3f ms per step  %.3e cell-substeps/s  frac %.3f  (%s)" % (pd, flat, best * 1e3, cells * 3 / best, cells * 32 / best / 8e12, dg.lib.hj_last_kernel(dg.ctx).decode()), flush=True)
