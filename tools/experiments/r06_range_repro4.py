import os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
import bench
g = bench.dubins_grid(L, 201, 201)
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
order = sys.argv[1].split(",")
sysd = bench._RangeSystem(g, 0.7, torch)
for kind in order:
    if kind == "hand":
        L.register_native_hamiltonian("bench_range", 3, bench.RANGE_SRC, nparams=1).attach(sysd, params=lambda o: [o.c])
    if kind == "split":
        os.environ["HJ_TRACE"] = "0"
    sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
    y, t = bench.device_sdf(torch, g, 0.5, ignore=(2,)).reshape(-1, 1), 0.0
    try:
        for k in range(3):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
    finally:
        os.environ.pop("HJ_TRACE", None)
    nwin, per = (3, 10) if kind == "split" else (9, 100)
    try:
        for w in range(nwin):
            torch.cuda.synchronize()
            for k in range(per):
                t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
            torch.cuda.synchronize()
        print(kind, "ok", t, bool(torch.isfinite(y).all()), flush=True)
    except Exception as e:
        print(kind, "FAILED in window", w, "step", k, repr(e)[:120], flush=True)
