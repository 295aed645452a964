#!/usr/bin/env python3
"""201^3: a Hamiltonian whose alpha reads the costate range under the three Lax-Friedrichs variants -- fused (run-time kernel) against the split
path (Python callbacks on device arrays), odeCFL3 singleStep calls.  (tests/test_gpu_round5.py's BurgersDriftLocal; test infrastructure import)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import levelsetpy_amd as L
from test_gpu_round5 import _burgers_src

n = int(sys.argv[1]) if len(sys.argv) > 1 else 201


class Sys(object):
    def __init__(self, g, c):
        self.grid, self.c = g, c
        self.x0 = torch.as_tensor(np.ascontiguousarray(np.asarray(g.xs[0])), device="cuda")
        self.ax0 = (self.c * self.x0).abs()

    def hamiltonian(self, t, data, p, sd=None):
        return 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + self.c * self.x0 * p[1]

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        lo, hi = dmin[dim], dmax[dim]
        if torch.is_tensor(lo) or torch.is_tensor(hi):
            lo = lo if torch.is_tensor(lo) else torch.as_tensor(float(lo), device="cuda", dtype=torch.float64)
            hi = hi if torch.is_tensor(hi) else torch.as_tensor(float(hi), device="cuda", dtype=torch.float64)
            a = torch.maximum(lo.abs(), hi.abs())
        else:
            a = max(abs(float(lo)), abs(float(hi)))
        return a + self.ax0 if dim == 1 else a


g = L.createGrid(-np.ones((3, 1)), np.ones((3, 1)), n * np.ones((3, 1), dtype=np.int64), None, low_mem=False)
d0 = torch.as_tensor(np.asarray(L.shapeSphere(g, np.zeros((3, 1)), 0.5)).reshape(-1, 1), device="cuda")
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
res = {}
for kind, fn in (("GLF", L.artificialDissipationGLF), ("LLF", L.artificialDissipationLLF), ("LLLF", L.artificialDissipationLLLF)):
    for path in ("split", "fused"):
        s = Sys(g, 0.7)
        if path == "fused":
            L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1).attach(s, params=lambda o: [o.c])
        sd = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=fn, CoStateCalc=L.upwindFirstWENO5))
        y, t = d0, 0.0
        for _ in range(3):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
        k = 5 if path == "split" else 40
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
        torch.cuda.synchronize()
        res[(kind, path)] = 1e3 * (time.perf_counter() - t0) / k
    print("%-4s %d^3: split %.3f ms/step, fused %.4f ms/step = %.1fx" % (kind, n, res[(kind, "split")], res[(kind, "fused")], res[(kind, "split")] / res[(kind, "fused")]), flush=True)
