"""round 6: BASELINE C5's system (4-D double pendulum, fp32) as FOREIGN Python callbacks -- lambdas around dynamics.DoublePendulum4D -- traced by the
library, against the built-in kernel and the split path; odeCFL3 single steps on device tensors.  argv: n0 n1 n2 n3"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
from levelsetpy_amd.context import device_grid
n = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [72, 72, 72, 129]
gmin = np.array([[-np.pi, -8, -np.pi, -8]]).T
gmax = np.array([[np.pi * (1 - 2 / n[0]), 8 * (1 - 2 / n[1]), np.pi * (1 - 2 / n[2]), 8 * (1 - 2 / n[3])]]).T
g = L.createGrid(gmin, gmax, np.array(n, dtype=np.int64).reshape(-1, 1), [0, 1, 2, 3], low_mem=True)
sysd = L.DoublePendulum4D(g, 1.0)
xs = [torch.as_tensor(np.asarray(v).ravel(), device="cuda", dtype=torch.float32) for v in g.vs]
shp = lambda d: [-1 if k == d else 1 for k in range(4)]
y0 = (sum((xs[d] ** 2).reshape(shp(d)) for d in range(4)).sqrt() - 0.5).contiguous().reshape(-1, 1)
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
res = {}
for kind in ("builtin", "traced", "split"):
    if kind == "split":
        os.environ["HJ_TRACE"] = "0"
    if kind == "builtin":
        sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
    else:
        sd = L.Bundle(dict(grid=g, hamFunc=lambda t, d, p, s: sysd.hamiltonian(t, d, p, s), partialFunc=lambda t, d, lo, hi, s, dim: sysd.dissipation(t, d, lo, hi, s, dim),
                           dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
    y, t = y0.clone(), 0.0
    t0 = time.perf_counter()
    for _ in range(2):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    k = 3 if kind == "split" else 20
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(k):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t1) / k
    dg = device_grid(g, "float32")
    res[kind] = (ms, y.clone(), t)
    cells = float(np.prod(n))
    print("%-8s %8.3f ms per step  %.3e cell-substeps/s  frac %.3f  first two steps %.2f s  kernel %s" % (
        kind, ms, cells * 3 / (ms * 1e-3), cells * 3 * 32 / 2 / (ms * 1e-3) / 8e12, first, dg.lib.hj_last_kernel(dg.ctx).decode()), flush=True)
    os.environ.pop("HJ_TRACE", None)
print(L.explain_plan(sd)["path"] if False else "")
