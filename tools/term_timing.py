#!/usr/bin/env python3
"""termNormal / termReinit / termConvection at 201^3 on device tensors: ONE fused launch (hj_term_*, round 3) against the
array path they replace (derivatives from hj_lf_split_begin / hj_upwind, then elementwise torch launches) -- the latter
is still what a foreign derivFunc gets.  Writes one line per term; tools/experiments/r03_run6.sh stores the output in
profiles/r03_term_kernels_201.txt."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 201
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T,
                 n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
phi = torch.as_tensor(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5), device="cuda")
phi = phi * (1.0 + 0.3 * torch.sin(2 * torch.linspace(-.75, 3.25, n, device="cuda", dtype=torch.float64)).reshape(n, 1, 1))
y = phi.reshape(-1, 1)
native = L.upwindFirstWENO5
foreign = lambda grid, data, dim: native(grid, data, dim)   # noqa: E731
speed = 0.5 + 0.3 * torch.cos(phi)
vel = [0.7, -0.4 + 0.5 * torch.sin(phi), -0.2]
cases = [("termNormal (array speed)", L.termNormal, dict(speed=speed)),
         ("termReinit (subcell fix 1)", L.termReinit, dict(initial=phi, subcell_fix_order=1)),
         ("termReinit (smeared sign)", L.termReinit, dict(initial=phi, subcell_fix_order=0)),
         ("termConvection", L.termConvection, dict(velocity=vel))]
print("%d^3 fp64, as-shipped WENO5 derivatives, ms per call (median of 7), device tensors in and out" % n)
for name, fn, extra in cases:
    res = []
    for deriv in (native, foreign):
        sd = L.Bundle(dict(grid=g, derivFunc=deriv, **extra))
        for _ in range(2):
            fn(0., y, sd)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn(0., y, sd)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res.append(1e3 * sorted(ts)[3])
    print("%-28s fused kernel %8.3f ms   array path %8.3f ms   x%.1f" % (name, res[0], res[1], res[1] / res[0]))
