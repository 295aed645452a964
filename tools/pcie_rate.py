#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in API: odeCFL3 with NumPy in / NumPy out at 201^3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetpy_amd as L
n = 201
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T,
                 n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
s = L.DubinsVehicleRel(g, 1, 1)
sd = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation,
                   dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
y = L.expand(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5).flatten(), 1)
t = 0.
for _ in range(2):
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
k = 10
t0 = time.perf_counter()
for _ in range(k):
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
sec = time.perf_counter() - t0
print("NumPy-in/NumPy-out odeCFL3 at %d^3: %.2f ms per step -> %.3e cell-substeps/s (PCIe both ways every call)"
      % (n, 1e3 * sec / k, n ** 3 * 3 * k / sec))
import torch
yt = torch.as_tensor(y, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(k):
    t, yt, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], yt, op, sd)
torch.cuda.synchronize()
sec = time.perf_counter() - t0
print("tensor-in/tensor-out odeCFL3 at %d^3: %.2f ms per step -> %.3e cell-substeps/s (state stays in HBM)"
      % (n, 1e3 * sec / k, n ** 3 * 3 * k / sec))
