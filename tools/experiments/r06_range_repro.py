import os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 201
g = bench.dubins_grid(L, n, n)
sysd = bench._RangeSystem(g, 0.7, torch)
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
y, t = bench.device_sdf(torch, g, 0.5, ignore=(2,)).reshape(-1, 1), 0.0
from levelsetpy_amd import trace_ham as TH
tr = TH.trace_callbacks(g, sysd.hamiltonian, sysd.dissipation, sd)
print(tr.source); print(tr.params, tr.uses_range)
os.environ["HJ_TRACE"] = "0"
sd0 = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
y0_, t0_ = y.clone(), 0.0
for k in range(4):
    t0_, y0_, _ = L.odeCFL3(L.termLaxFriedrichs, [t0_, 1e9], y0_, op, sd0)
print("split steps done", t0_)
del os.environ["HJ_TRACE"]
try:
    for k in range(3):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
        print("step", k, t)
except Exception:
    traceback.print_exc()
