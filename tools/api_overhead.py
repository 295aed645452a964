#!/usr/bin/env python3
"""Per-step cost through the drop-in API (odeCFL3 on a CUDA tensor over a time span) vs the raw C loop."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L
n = int(os.environ.get("N", "201"))
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
sys_ = L.DubinsVehicleRel(g, 1, 1)
sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
y0 = torch.as_tensor(np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)), device="cuda").reshape(-1, 1)
for label, opt in (("span (native loop)", dict(factorCFL=.8, singleStep='off')),
                   ("span with a postTimeStep hook (Python loop)", dict(factorCFL=.8, singleStep='off', postTimeStep=lambda t, y, s: (y, s)))):
    op = L.odeCFLset(L.Bundle(opt))
    L.odeCFL3(L.termLaxFriedrichs, [0., 0.05], y0, op, sd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 0.5], y0, op, sd)
    torch.cuda.synchronize(); sec = time.perf_counter() - t0
    dt = 0.8 / ((1 + 1.25) / float(g.dx[0]) + (1 + 3.25) / float(g.dx[1]) + 2 / float(g.dx[2]))
    steps = int(np.ceil(0.5 / dt))
    print("%-48s t=%.4f  ~%d steps  %.3f ms/step  %.3e cell-substeps/s" % (label, float(t), steps, 1e3 * sec / steps, n ** 3 * 3 * steps / sec), flush=True)
