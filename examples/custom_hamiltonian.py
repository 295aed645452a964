#!/usr/bin/env python3
"""A system the library was NOT built with, whose dissipation coefficients depend on the costate RANGE -- the general protocol of the
reference's artificialDissipationGLF, which hands partialFunc the minimum and maximum of the one-sided derivatives over the grid
(ExplicitIntegration/Dissipation/artificial_diss_glf.py:80-99).

    python examples/custom_hamiltonian.py [n] [steps]

H(x, p) = |p|^2 / 2 + c x_0 p_1   (a convex Hamiltonian with a drift),   alpha_d = max |dH/dp_d| over the costate range.

The system is written ONCE as the reference wants it -- a Python object with .hamiltonian / .dissipation on arrays.  Three ways to run it:
(1) HJ_TRACE=0: the split path (derivative kernels -> these callbacks -> a dissipation kernel), what every foreign callable took until
round 6; (2) default: the library TRACES the callbacks (calls them once with symbolic arrays), compiles the recorded expression with
hipRTC, checks the kernel against the callbacks on the first data and runs fused -- nothing to write; (3) the same pair written by hand as
a device expression and attached to the object (for callbacks the tracer refuses: Python control flow on array values, reductions, ...).
Fused: a range pass + the fused substep per RK stage, deltaT from the first stage's bound exactly as ode_cfl_3.py:142 takes it, no host
round trip inside a step.  Needs an MI355X (the package has no CPU fallback)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as lsp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 151
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
c = 0.7


class BurgersDrift(object):
    """The reference's callback protocol, on NumPy arrays or device tensors."""

    def __init__(self, grid):
        self.grid = grid

    def hamiltonian(self, t, data, p, schemeData=None):
        x0 = torch.as_tensor(np.asarray(self.grid.xs[0]), device=p[0].device) if torch.is_tensor(p[0]) else np.asarray(self.grid.xs[0])
        return 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + c * x0 * p[1]

    def dissipation(self, t, data, derivMin, derivMax, schemeData, dim):
        a = np.maximum(abs(derivMin[dim]), abs(derivMax[dim]))               # |dH/dp_d| = |p_d| (+ |c x_0| for d = 1), bounded over the range
        if dim != 1:
            return a
        x0 = np.abs(c * np.asarray(self.grid.xs[0]))
        return a + (torch.as_tensor(x0, device=data.device) if torch.is_tensor(data) else x0)


gmin, gmax = -np.ones((3, 1)), np.ones((3, 1))
g = lsp.createGrid(gmin, gmax, n * np.ones((3, 1), dtype=np.int64), None)
system = BurgersDrift(g)
data0 = lsp.shapeSphere(g, np.zeros((3, 1)), 0.5)
def bundle():
    # (a fresh Bundle per run: the library caches what it found out about a schemeData on the Bundle)
    return lsp.Bundle(dict(grid=g, hamFunc=system.hamiltonian, partialFunc=system.dissipation,
                           dissFunc=lsp.artificialDissipationGLF, CoStateCalc=lsp.upwindFirstWENO5))


opts = lsp.odeCFLset(lsp.Bundle(dict(factorCFL=0.8, singleStep='on')))


def run(label):
    sd = bundle()
    y, t = torch.as_tensor(data0.reshape(-1, 1), device="cuda"), 0.0
    t, y, _ = lsp.odeCFL3(lsp.termLaxFriedrichs, [t, 1e9], y, opts, sd)            # warm-up (and, on the fused path, the compilation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t, y, _ = lsp.odeCFL3(lsp.termLaxFriedrichs, [t, 1e9], y, opts, sd)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print("%-44s %7.3f ms per RK3 step   t = %.6f" % (label, ms, t))
    return t, y


os.environ["HJ_TRACE"] = "0"
t_split, y_split = run("as written, HJ_TRACE=0 (split path):")
del os.environ["HJ_TRACE"]
t_traced, y_traced = run("as written, traced by the library (fused):")
print("    the expression the tracer wrote:\n        " + lsp.trace_callbacks(g, system.hamiltonian, system.dissipation).source.replace("\n", "\n        "))

# the same H / alpha once more, as a device expression: x[d] node, p[d] costate, par[k] parameters, dmin[d] / dmax[d] the costate range
reg = lsp.register_native_hamiltonian("burgers_drift_example", 3, """
    H = par[0] * x[0] * p[1] + 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    alpha[0] = fmax(fabs(dmin[0]), fabs(dmax[0]));
    alpha[1] = fmax(fabs(dmin[1]), fabs(dmax[1])) + fabs(par[0] * x[0]);
    alpha[2] = fmax(fabs(dmin[2]), fabs(dmax[2]));
""", nparams=1)
reg.attach(system, params=lambda s: [c])          # the SAME object, the SAME schemeData: selected by callable identity

t_fused, y_fused = run("hand-written device expression attached (fused):")
for name, tt, yy in (("traced", t_traced, y_traced), ("hand-written", t_fused, y_fused)):
    print("same trajectory (%s): |t - t_split| = %.1e, max |y - y_split| = %.1e" % (name, abs(tt - t_split), float((yy - y_split).abs().max())))
