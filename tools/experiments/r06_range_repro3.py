import os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
import bench
g = bench.dubins_grid(L, 201, 201)
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
res = {}
for kind in ("traced", "hand"):
    sysd = bench._RangeSystem(g, 0.7, torch)
    if kind == "hand":
        L.register_native_hamiltonian("bench_range", 3, bench.RANGE_SRC, nparams=1).attach(sysd, params=lambda o: [o.c])
    sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
    y, t = bench.device_sdf(torch, g, 0.5, ignore=(2,)).reshape(-1, 1), 0.0
    try:
        for k in range(1200):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
            if k % 100 == 99:
                print(kind, k + 1, "t", t, "finite", bool(torch.isfinite(y).all()), "max|y|", float(y.abs().max()), flush=True)
    except Exception as e:
        print(kind, "FAILED at step", k, repr(e)[:200], "finite", bool(torch.isfinite(y).all()), "max|y|", float(y.abs().max()))
    res[kind] = y
