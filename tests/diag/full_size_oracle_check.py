#!/usr/bin/env python3
"""BASELINE C4 (513^3 Dubins, fp64) and C5 (129^4 pendulum, fp32) against the CPU oracle at FULL size, once (minutes of one host core:
too slow for the suite, which checks these sizes through size-independent properties).  C4: one odeCFL3 step through the drop-in
API against oracle.ode_cfl_3; C5: one termLaxFriedrichs evaluation against oracle.term_lax_friedrichs (fp32 product, fp64 oracle).
"range": a run-time Hamiltonian whose alpha reads the costate range, 201^3 fp64 (the size of the bench's range leg: the two-pairs-per-thread
run-time kernels), one odeCFL3 step under artificialDissipationGLF / LLF / LLLF against the oracle, WENO5_ASSHIPPED and the intended WENO5.
Test infrastructure: imports oracle/.  usage: tests/diag/full_size_oracle_check.py [c4|c5|both|range]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import levelsetpy_amd as L
from oracle import hj_oracle as O

what = sys.argv[1] if len(sys.argv) > 1 else "both"


def mem_available_gb():
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable:"):
            return int(ln.split()[1]) / 1e6
    return 0.0


# the NumPy oracle keeps a few dozen whole-grid temporaries alive: refuse to start on a host that could run out of memory
need = {"c4": 80.0, "c5": 160.0}
for leg in ("c4", "c5"):
    if what in (leg, "both") and mem_available_gb() < need[leg]:
        print("%s: %.0f GB of host memory available, %.0f wanted: leg skipped" % (leg, mem_available_gb(), need[leg]))
        what = {"both": "c5" if leg == "c4" else "c4", leg: "none"}.get(what, what)
if what in ("c4", "both"):
    n = int(os.environ.get("N", "513"))
    lo, hi = [-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n)]
    g = L.createGrid(np.array([lo]).T, np.array([hi]).T, n * np.ones((3, 1), dtype=np.int64), 2)
    og = O.Grid(lo, hi, [n, n, n], [2])
    d0 = O.shape_cylinder(og, 2, None, .5)
    sysn = L.DubinsVehicleRel(g, 1, 1)
    sd = L.Bundle(dict(grid=g, hamFunc=sysn.hamiltonian, partialFunc=sysn.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], torch.as_tensor(d0.reshape(-1, 1), device="cuda"), op, sd)
    got = y.cpu().numpy()
    print("C4 %d^3: product step done (t = %.6e); oracle running on one core ..." % (n, t), flush=True)
    osys = O.DubinsRel(og, 1, 1)
    calls = [0]
    def term(tt, yy):
        calls[0] += 1
        t0 = time.time()
        r = O.term_lax_friedrichs(og, osys, "WENO5_ASSHIPPED", tt, yy)
        print("   oracle stage %d: %.0f s" % (calls[0], time.time() - t0), flush=True)
        return r
    t0 = time.time()
    to, yo = O.ode_cfl_3(term, [0., 10.], d0.reshape(-1, 1), 0.8, single_step=True)
    diff = float(np.abs(got - yo).max())
    print("C4 %d^3 one odeCFL3 step vs oracle: max |diff| = %.3e (max |y| %.3f), t %r / %r, oracle %.0f s" % (n, diff, float(np.abs(yo).max()), t, to, time.time() - t0), flush=True)
    assert diff <= 1e-11 and abs(t - to) <= 1e-14
    del got, yo, y, d0
if what in ("c5", "both"):
    q = int(os.environ.get("Q", "129"))
    lo, hi = [-np.pi] * 4, [np.pi * (1 - 2 / q)] * 4
    g = L.createGrid(np.array([lo]).T, np.array([hi]).T, q * np.ones((4, 1), dtype=np.int64), [0, 1, 2, 3], low_mem=True)
    og = O.Grid(lo, hi, [q] * 4, [0, 1, 2, 3])
    d0 = O.shape_sphere(og, None, .5)
    s4 = L.DoublePendulum4D(g, 1.0)
    sd = L.Bundle(dict(grid=g, hamFunc=s4.hamiltonian, partialFunc=s4.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
    yd, sb, _ = L.termLaxFriedrichs(0., torch.as_tensor(d0.astype(np.float32).reshape(-1, 1), device="cuda"), sd)
    got = yd.cpu().numpy().astype(np.float64)
    print("C5 %d^4 fp32: product term done (stepBound %.6e); oracle (fp64) running on one core ..." % (q, sb), flush=True)
    t0 = time.time()
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), "WENO5_ASSHIPPED", 0., d0.astype(np.float32).astype(np.float64).reshape(-1, 1))
    scale = float(np.abs(yo).max())
    diff = float(np.abs(got - yo).max())
    print("C5 %d^4 termLaxFriedrichs (fp32) vs oracle (fp64): max |diff| = %.3e = %.2e of max |ydot| %.3f; stepBound rel. diff %.2e; oracle %.0f s" %
          (q, diff, diff / scale, scale, abs(sb - sbo) / sbo, time.time() - t0), flush=True)
    assert diff <= 1e-4 * scale and abs(sb - sbo) <= 1e-5 * sbo
if what == "range":
    class Drift(object):            # H = |p|^2/2 + c p0 p1: alpha_0 = max|p0| + |c| max|p1|, alpha_1 = max|p1| + |c| max|p0| over the ranges handed in --
        def __init__(self, grid, c):    # alpha_i reads ANOTHER dimension's range, so GLF / LLF / LLLF are three different schemes; scalars or arrays
            self.grid, self.c = grid, c

        def hamiltonian(self, t, data, p, sd=None):
            return 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + self.c * p[0] * p[1]

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            am = lambda d: np.maximum(np.abs(dmin[d]), np.abs(dmax[d]))  # noqa: E731
            return am(dim) + (abs(self.c) * am(1 - dim) if dim < 2 else 0.0)
    src = "H = par[0] * p[0] * p[1];\n" + "".join(
        "H += 0.5 * p[%d] * p[%d];  alpha[%d] = fmax(fabs(dmin[%d]), fabs(dmax[%d]));\n" % (d, d, d, d, d) for d in range(3)) + \
        "alpha[0] += fabs(par[0]) * fmax(fabs(dmin[1]), fabs(dmax[1]));\nalpha[1] += fabs(par[0]) * fmax(fabs(dmin[0]), fabs(dmax[0]));\n"
    n = int(os.environ.get("N", "201"))
    lo, hi = [-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n]
    g = L.createGrid(np.array([lo]).T, np.array([hi]).T, n * np.ones((3, 1), dtype=np.int64), 2)
    og = O.Grid(lo, hi, [n, n, n], [2])
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    reg = L.register_native_hamiltonian("coupled_burgers_3d", 3, src, nparams=1)
    sysn = Drift(g, 0.6)
    reg.attach(sysn, params=lambda o: [o.c])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    kinds = (("glf", L.artificialDissipationGLF), ("llf", L.artificialDissipationLLF), ("lllf", L.artificialDissipationLLLF))
    for scheme, deriv in (("WENO5_ASSHIPPED", L.upwindFirstWENO5), ("WENO5", L.upwindFirstWENO5Intended)):
        for kind, diss in kinds:
            sd = L.Bundle(dict(grid=g, hamFunc=sysn.hamiltonian, partialFunc=sysn.dissipation, dissFunc=diss, CoStateCalc=deriv))
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], torch.as_tensor(d0.reshape(-1, 1), device="cuda"), op, sd)
            dgs = g.__dict__["_hj_device"]
            dgx = dgs[next(iter(dgs))] if isinstance(dgs, dict) else dgs
            kern = dgx.lib.hj_last_kernel(dgx.ctx).decode()
            t0 = time.time()
            osys = Drift(og, 0.6)
            to, yo = O.ode_cfl_3(lambda tt, yy: O.term_lax_friedrichs(og, osys, scheme, tt, yy, diss=kind), [0., 10.], d0.reshape(-1, 1), 0.8, single_step=True)
            diff = float(np.abs(y.cpu().numpy() - yo).max())
            print("range-alpha %d^3 %-16s %-4s one odeCFL3 step vs oracle: max |diff| = %.3e (max |y| %.3f), t %.16e / %.16e, oracle %.0f s %s" % (
                n, scheme, kind, diff, float(np.abs(yo).max()), t, to, time.time() - t0, kern), flush=True)
            assert diff <= 1e-11 and abs(t - to) <= 1e-13 * to
print("full-size oracle check: ok")
