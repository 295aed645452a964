#!/usr/bin/env python3
"""Backward reachable tube of the two-Dubins-vehicle pursuit game (the reference's Notes/rcbrt.ipynb,
`air3D`), written exactly as a LevelSetPy user would write it -- only the import changes.

    python examples/air3d_brt.py [n] [t_end]

Needs an MI355X (the package has no CPU fallback).  Prints the time per tau interval and the volume of
the tube; the value function stays on the GPU for the whole solve.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import levelsetpy_amd as lsp   # was: from LevelSetPy... import *

n = int(sys.argv[1]) if len(sys.argv) > 1 else 101
t_end = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0

# grid as in the notebook: x in [-.75, 3.25], y in [-1.25, 1.25], heading periodic
gmin = np.array([[-.75, -1.25, -np.pi]]).T
gmax = np.array([[3.25, 1.25, np.pi]]).T
N = n * np.ones((3, 1), dtype=np.int64)
gmax[2] *= (1 - 2 / N[2])                      # periodic axis: drop the duplicate node (notebook cell 3)
g = lsp.createGrid(gmin, gmax, N, 2)

# target set: cylinder of radius 0.5 around the evader, any heading
data0 = lsp.shapeCylinder(g, 2, np.zeros((3, 1)), 0.5)

dubins = lsp.DubinsVehicleRel(g, 1, 1)          # unit speeds and turn rates
schemeData = lsp.Bundle(dict(grid=g, accuracy='veryHigh',
                             hamFunc=dubins.hamiltonian, partialFunc=dubins.dissipation,
                             dissFunc=lsp.artificialDissipationGLF, CoStateCalc=lsp.upwindFirstWENO5))
tau = np.linspace(0, t_end, 11)
extra = lsp.Bundle(dict(quiet=True, keepLast=True))

t0 = time.perf_counter()
brt, tau_out, _ = lsp.HJIPDE_solve(data0, tau, schemeData, 'minVOverTime', extra)
sec = time.perf_counter() - t0

cell = float(np.prod(np.asarray(g.dx)))
print("grid %d^3, t_end %.2f: %.2f s for %d tau intervals" % (n, t_end, sec, len(tau_out) - 1))
print("volume of the initial set   %.4f" % (cell * np.count_nonzero(np.asarray(data0) <= 0)))
print("volume of the reachable tube %.4f" % (cell * np.count_nonzero(np.asarray(brt) <= 0)))
