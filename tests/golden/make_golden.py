#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the UNMODIFIED
reference (robotsorcerer/LevelSetPy, mounted read-only at /root/reference).

Build-container only: the reference never travels.  Run as

    python tests/golden/make_golden.py

The reference's `import cupy` resolves to the NumPy-backed stand-in in
oracle/_harness/cupy (our code), with CuPy's out-of-bounds wrap emulated for
upwindFirstWENO5.  Everything written here is DATA: seeded
inputs and the outputs the reference produced for them.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_harness"))
import ref_loader  # noqa: E402

ref_loader.load()

from LevelSetPy.Utilities import Bundle, expand  # noqa: E402
from LevelSetPy.Grids import createGrid  # noqa: E402
from LevelSetPy.InitialConditions import shapeCylinder, shapeSphere  # noqa: E402
from LevelSetPy.DynamicalSystems import DubinsVehicleRel, DoubleIntegrator  # noqa: E402
from LevelSetPy.BoundaryCondition import addGhostExtrapolate, addGhostPeriodic  # noqa: E402
from LevelSetPy.SpatialDerivative import (upwindFirstWENO5, upwindFirstENO3,  # noqa: E402
                                          upwindFirstENO2)
from LevelSetPy.SpatialDerivative.ENO3aHelper import upwindFirstENO3aHelper  # noqa: E402
from LevelSetPy.ExplicitIntegration import (odeCFL2, odeCFL3, odeCFLset,  # noqa: E402
                                            termLaxFriedrichs, termRestrictUpdate,
                                            artificialDissipationGLF)
from LevelSetPy.ValueFuncs import HJIPDE_solve  # noqa: E402

A = np.asarray
SCHEMES = {"ENO2": upwindFirstENO2, "ENO3": upwindFirstENO3, "WENO5_ASSHIPPED": upwindFirstWENO5}


def col(v):
    return np.asarray(v, dtype=np.float64).reshape(-1, 1)


def dubins_grid(n):
    n = np.atleast_1d(n)
    if n.size == 1:
        n = np.repeat(n, 3)
    gmin = col([-.75, -1.25, -np.pi])
    gmax = col([3.25, 1.25, np.pi * (1 - 2 / n[2])])
    return createGrid(gmin, gmax, n.reshape(-1, 1).astype(np.int64), 2), gmin, gmax, n


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print("wrote %-22s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ (1) ghost cells
def gen_ghost():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((8, 9, 10))
    # force the edge cases the sign-aware extrapolation has: zero and sign changes at edges
    x[0, 0, :] = 0.0
    x[-1, 1, :] = 0.0
    x[:, 0, 2] = 0.0
    x[:, -1, 3] = np.abs(x[:, -1, 3])
    x[3, :, 0] = -np.abs(x[3, :, 0])
    x[4, :, -1] = 0.0
    out = {"x": x}
    for dim in range(3):
        for w in (1, 2, 3):
            out["per_d%d_w%d" % (dim, w)] = A(addGhostPeriodic(x, dim, w, None))
            out["ext_d%d_w%d_tz0" % (dim, w)] = A(addGhostExtrapolate(x, dim, w, None))
            out["ext_d%d_w%d_tz1" % (dim, w)] = A(
                addGhostExtrapolate(x, dim, w, Bundle(dict(towardZero=True))))
    x32 = x.astype(np.float32)
    out["ext_f32in_d1_w2"] = A(addGhostExtrapolate(x32, 1, 2, None))
    save("ghost.npz", **out)


# ------------------------------------------------------------------ (2,3) derivatives
def deriv_case(tag, g, data, out, helper=True):
    out[tag + "_data"] = data
    out[tag + "_dx"] = A(g.dx).ravel()
    out[tag + "_bc"] = A([1 if b is addGhostPeriodic else 0 for b in g.bdry])
    for dim in range(g.dim):
        if helper:
            dL, dR, DD = upwindFirstENO3aHelper(g, data, dim, False, False)
            for k in range(3):
                out["%s_helper_dL%d_d%d" % (tag, k, dim)] = A(dL[k])
                out["%s_helper_dR%d_d%d" % (tag, k, dim)] = A(dR[k])
            out["%s_helper_D1_d%d" % (tag, dim)] = A(DD.D1)
            out["%s_helper_D2_d%d" % (tag, dim)] = A(DD.D2)
            out["%s_helper_D3_d%d" % (tag, dim)] = A(DD.D3)
        for name, fn in SCHEMES.items():
            L, R = fn(g, data, dim)
            out["%s_%s_L_d%d" % (tag, name, dim)] = A(L)
            out["%s_%s_R_d%d" % (tag, name, dim)] = A(R)


def gen_deriv():
    rng = np.random.default_rng(0)
    out = {}
    # 2-D, 16x12, axis 1 periodic
    g2 = createGrid(col([-1, -np.pi]), col([1, np.pi * (1 - 2 / 12)]),
                    A([[16], [12]], dtype=np.int64), 1)
    d2 = shapeSphere(g2, np.zeros((2, 1)), .5) + 0.3 * rng.standard_normal(g2.shape)
    deriv_case("g2", g2, d2, out)
    out["g2_min"], out["g2_max"] = A(g2.min).ravel(), A(g2.max).ravel()
    # 3-D Dubins grid 9x10x11 (axis 2 periodic), SDF + noise
    g3, gmin, gmax, n = dubins_grid([9, 10, 11])
    d3 = shapeCylinder(g3, 2, np.zeros((3, 1)), .5) + 0.05 * rng.standard_normal(g3.shape)
    deriv_case("g3", g3, d3, out)
    out["g3_min"], out["g3_max"] = gmin.ravel(), gmax.ravel()
    # 3-D smooth data (no noise): WENO/ENO on the signed-distance cylinder
    d3s = shapeCylinder(g3, 2, np.zeros((3, 1)), .5)
    deriv_case("g3s", g3, d3s, out, helper=False)
    # 4-D 5x6x7x8, all axes periodic (bdry set by hand: createGrid only takes one pdDim)
    n4 = A([[5], [6], [7], [8]], dtype=np.int64)
    g4 = createGrid(col([-1, -2, -1, -2]), col([1, 2, 1, 2]), n4, None)
    for i in range(4):
        g4.bdry[i] = addGhostPeriodic
    d4 = rng.standard_normal(g4.shape)
    deriv_case("g4", g4, d4, out, helper=False)
    out["g4_min"], out["g4_max"] = A(g4.min).ravel(), A(g4.max).ravel()
    save("deriv.npz", **out)


# ------------------------------------------------------------------ (4,5) GLF + LF term
def gen_term():
    rng = np.random.default_rng(2)
    out = {}
    g3, gmin, gmax, n = dubins_grid([13, 12, 11])
    out["dub_min"], out["dub_max"], out["dub_N"] = gmin.ravel(), gmax.ravel(), n
    d3 = shapeCylinder(g3, 2, np.zeros((3, 1)), .5) + 0.02 * rng.standard_normal(g3.shape)
    out["dub_data"] = d3
    for (ub, wb) in ((1, 1), (5, 5), (2, 3)):
        sys_ = DubinsVehicleRel(g3, ub, wb)
        for name, fn in SCHEMES.items():
            sd = Bundle(dict(grid=g3, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                             dissFunc=artificialDissipationGLF, CoStateCalc=fn))
            y = expand(d3.flatten(), 1)
            yd, sb, _ = termLaxFriedrichs(0.3, y, sd)
            out["dub_u%d_w%d_%s_ydot" % (ub, wb, name)] = A(yd)
            out["dub_u%d_w%d_%s_sb" % (ub, wb, name)] = A(sb)
    # GLF alone on random left/right derivative fields
    sys_ = DubinsVehicleRel(g3, 1, 1)
    dL = [rng.standard_normal(g3.shape) for _ in range(3)]
    dR = [rng.standard_normal(g3.shape) for _ in range(3)]
    sd = Bundle(dict(grid=g3, partialFunc=sys_.dissipation))
    diss, sb = artificialDissipationGLF(0., d3, dL, dR, sd)
    for i in range(3):
        out["glf_dL%d" % i], out["glf_dR%d" % i] = dL[i], dR[i]
    out["glf_diss"], out["glf_sb"] = A(diss), A(sb)
    # double integrator 24x20, both axes extrapolate
    g2 = createGrid(col([-1, -1]), col([1, 1]), A([[24], [20]], dtype=np.int64), None)
    d2 = shapeSphere(g2, np.zeros((2, 1)), .25) + 0.02 * rng.standard_normal(g2.shape)
    out["di_min"], out["di_max"], out["di_N"] = A(g2.min).ravel(), A(g2.max).ravel(), A([24, 20])
    out["di_data"] = d2
    for ub in (1, 2.5):
        sys2 = DoubleIntegrator(g2, ub)
        for name, fn in SCHEMES.items():
            sd = Bundle(dict(grid=g2, hamFunc=sys2.hamiltonian, partialFunc=sys2.dissipation,
                             dissFunc=artificialDissipationGLF, CoStateCalc=fn))
            yd, sb, _ = termLaxFriedrichs(0., expand(d2.flatten(), 1), sd)
            out["di_u%s_%s_ydot" % (ub, name)] = A(yd)
            out["di_u%s_%s_sb" % (ub, name)] = A(sb)
    save("term.npz", **out)


# ------------------------------------------------------------------ (6) integrators
def gen_ode():
    out = {}
    g3, gmin, gmax, n = dubins_grid([21, 21, 21])
    out["dub_min"], out["dub_max"], out["dub_N"] = gmin.ravel(), gmax.ravel(), n
    d3 = shapeCylinder(g3, 2, np.zeros((3, 1)), .5)
    out["dub_data"] = d3
    sys_ = DubinsVehicleRel(g3, 1, 1)
    for name, fn in SCHEMES.items():
        sd = Bundle(dict(grid=g3, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                         dissFunc=artificialDissipationGLF, CoStateCalc=fn))
        op = odeCFLset(Bundle(dict(factorCFL=.8, singleStep='on')))
        # RK3 single steps, (N,1) state as HJIPDE_solve passes it
        y = expand(d3.flatten(), 1)
        t = 0.
        for k in range(5):
            t, y, _ = odeCFL3(termLaxFriedrichs, [t, 10.], y, op, sd)
            if k in (0, 4):
                out["rk3_%s_t%d" % (name, k + 1)] = A(float(t))
                out["rk3_%s_y%d" % (name, k + 1)] = A(y)
        # RK2 to a final time (multi-step loop incl. a truncated last step), factorCFL .95
        op2 = odeCFLset(Bundle(dict(factorCFL=.95, singleStep='off')))
        t, y, _ = odeCFL2(termLaxFriedrichs, [0., 0.02], expand(d3.flatten(), 1), op2, sd)
        out["rk2_%s_t" % name], out["rk2_%s_y" % name] = A(float(t)), A(y)
        # notebook pattern: odeCFL2(termRestrictUpdate, ..., y (N,), positive=0)
        inner = Bundle(dict(grid=g3, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                            dissFunc=artificialDissipationGLF, CoStateCalc=fn))
        sdr = Bundle(dict(innerFunc=termLaxFriedrichs, innerData=inner, positive=0))
        t, y, _ = odeCFL2(termRestrictUpdate, [0., 0.02], d3.flatten(), op2, sdr)
        out["rk2r_%s_t" % name], out["rk2r_%s_y" % name] = A(float(t)), A(y)
    # the same with 1e-2 noise on the initial data: no exact |D2|/|D3| ties, so ENO stencil choices
    # do not hinge on rounding (the exactly symmetric cylinder above is all ties)
    d3n = d3 + 0.01 * np.random.default_rng(4).standard_normal(d3.shape)
    out["dubn_data"] = d3n
    for name, fn in SCHEMES.items():
        sd = Bundle(dict(grid=g3, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                         dissFunc=artificialDissipationGLF, CoStateCalc=fn))
        op = odeCFLset(Bundle(dict(factorCFL=.8, singleStep='on')))
        y = expand(d3n.flatten(), 1)
        t = 0.
        for k in range(5):
            t, y, _ = odeCFL3(termLaxFriedrichs, [t, 10.], y, op, sd)
        out["rk3n_%s_t5" % name], out["rk3n_%s_y5" % name] = A(float(t)), A(y)
    # double integrator RK3
    g2 = createGrid(col([-1, -1]), col([1, 1]), A([[32], [32]], dtype=np.int64), None)
    d2 = shapeSphere(g2, np.zeros((2, 1)), .25)
    sys2 = DoubleIntegrator(g2, 1)
    out["di_data"] = d2
    sd = Bundle(dict(grid=g2, hamFunc=sys2.hamiltonian, partialFunc=sys2.dissipation,
                     dissFunc=artificialDissipationGLF, CoStateCalc=upwindFirstENO3))
    op = odeCFLset(Bundle(dict(factorCFL=.8, singleStep='on')))
    y = expand(d2.flatten(), 1)
    t = 0.
    for k in range(5):
        t, y, _ = odeCFL3(termLaxFriedrichs, [t, 10.], y, op, sd)
    out["di_rk3_ENO3_t5"], out["di_rk3_ENO3_y5"] = A(float(t)), A(y)
    d2n = d2 + 0.01 * np.random.default_rng(5).standard_normal(d2.shape)
    out["din_data"] = d2n
    y = expand(d2n.flatten(), 1)
    t = 0.
    for k in range(5):
        t, y, _ = odeCFL3(termLaxFriedrichs, [t, 10.], y, op, sd)
    out["din_rk3_ENO3_t5"], out["din_rk3_ENO3_y5"] = A(float(t)), A(y)
    save("ode.npz", **out)


# ------------------------------------------------------------------ (7) HJIPDE_solve
def gen_hjipde():
    out = {}
    g3, gmin, gmax, n = dubins_grid([21, 21, 21])
    d3 = shapeCylinder(g3, 2, np.zeros((3, 1)), .5)
    sys_ = DubinsVehicleRel(g3, 1, 1)
    tau = np.array([0., .05, .1])
    for comp in ("minVOverTime", "maxVOverTime", "set"):  # minVWithV0 crashes as shipped (hji_solver.py:578)
        sd = Bundle(dict(grid=g3, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                         CoStateCalc=upwindFirstWENO5, uMode='min', dMode='max'))
        extra = Bundle(dict(keepLast=True, quiet=True, visualize=False))
        data, tau_out, _ = HJIPDE_solve(d3.copy(), tau.copy(), sd, comp, extra)
        out["hj_%s_data" % comp] = A(data)
        out["hj_%s_tau" % comp] = A(tau_out)
    out["hj_data0"], out["hj_tau"] = d3, tau
    out["dub_min"], out["dub_max"], out["dub_N"] = gmin.ravel(), gmax.ravel(), n
    save("hjipde.npz", **out)


# ------------------------------------------------------------------ (8) known answers
def gen_known():
    ka = {}
    g3, gmin, gmax, n = dubins_grid(51)
    d3 = shapeCylinder(g3, 2, np.zeros((3, 1)), .5)
    sys_ = DubinsVehicleRel(g3, 1, 1)
    for name, fn in SCHEMES.items():
        sd = Bundle(dict(grid=g3, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                         dissFunc=artificialDissipationGLF, CoStateCalc=fn))
        op = odeCFLset(Bundle(dict(factorCFL=.8, singleStep='on')))
        y = expand(d3.flatten(), 1)
        yd, sb, _ = termLaxFriedrichs(0., y, sd)
        rec = {"stepBound0": float(sb), "ydot0_l2": float(np.linalg.norm(yd))}
        t = 0.
        for k in range(5):
            t, y, _ = odeCFL3(termLaxFriedrichs, [t, 10.], y, op, sd)
            if k == 0:
                rec["t1"] = float(t)
        rec.update(t5=float(t), sum5=float(y.sum()), l2_5=float(np.linalg.norm(y)),
                   min5=float(y.min()), max5=float(y.max()))
        ka["dubins51_" + name] = rec
    g2 = createGrid(col([-1, -1]), col([1, 1]), A([[128], [128]], dtype=np.int64), None)
    d2 = shapeSphere(g2, np.zeros((2, 1)), .25)
    sys2 = DoubleIntegrator(g2, 1)
    sd = Bundle(dict(grid=g2, hamFunc=sys2.hamiltonian, partialFunc=sys2.dissipation,
                     dissFunc=artificialDissipationGLF, CoStateCalc=upwindFirstENO3))
    op = odeCFLset(Bundle(dict(factorCFL=.8, singleStep='on')))
    y = expand(d2.flatten(), 1)
    yd, sb, _ = termLaxFriedrichs(0., y, sd)
    rec = {"stepBound0": float(sb), "ydot0_l2": float(np.linalg.norm(yd))}
    t = 0.
    for k in range(5):
        t, y, _ = odeCFL3(termLaxFriedrichs, [t, 10.], y, op, sd)
    rec.update(t5=float(t), sum5=float(y.sum()), l2_5=float(np.linalg.norm(y)),
               min5=float(y.min()), max5=float(y.max()))
    ka["dint128_ENO3"] = rec
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, sort_keys=True)
    print("wrote known_answers.json")


# ------------------------------------------------------------------ (9) round 2: what else the shipped code can run
def gen_extra():
    """(a) upwindFirstENO3aHelper with approx4 (fourth candidate); (b) artificialDissipationLLF for the one case
    the shipped function runs: a partialFunc that returns a 0-d array for EVERY dimension, so that
    `(1 / stepBoundInv).get().item()` (diss_local_laxfried.py:121) is applied to a scalar.  The alpha of
    dimension i is built from the GLOBAL costate range of another dimension (derivMin/derivMax are scalars
    there, :84-99) and from the per-node range of dimension i reduced to its extrema (:108-109), so the
    fixture pins what ranges the function hands to partialFunc."""
    import cupy as cp
    from LevelSetPy.ExplicitIntegration.Dissipation.diss_local_laxfried import artificialDissipationLLF
    rng = np.random.default_rng(7)
    out = {}
    g3, gmin, gmax, n = dubins_grid([9, 10, 11])
    d3 = shapeCylinder(g3, 2, np.zeros((3, 1)), .5) + 0.05 * rng.standard_normal(g3.shape)
    out["g3_min"], out["g3_max"], out["g3_data"] = gmin.ravel(), gmax.ravel(), d3
    for dim in range(3):
        dL, dR, DD = upwindFirstENO3aHelper(g3, d3, dim, True, False)
        assert len(dL) == 4
        out["g3_helper4_dL3_d%d" % dim] = A(dL[3])
        out["g3_helper4_dR3_d%d" % dim] = A(dR[3])
    dL = [rng.standard_normal(g3.shape) for _ in range(3)]
    dR = [rng.standard_normal(g3.shape) for _ in range(3)]
    seen = []

    def partial(t, data, derivMin, derivMax, schemeData, dim):
        j = (dim + 1) % 3
        glob = max(abs(float(A(derivMin[j]))), abs(float(A(derivMax[j]))))       # scalars: the global range of dim j
        loc = max(abs(float(A(derivMin[dim]).min())), abs(float(A(derivMax[dim]).max())))   # arrays: per-node range of dim
        seen.append((dim, tuple(np.ndim(A(derivMin[k])) for k in range(3))))
        return cp.asarray(0.25 * (dim + 1) + 0.5 * glob + 0.125 * loc)

    sd = Bundle(dict(grid=g3, partialFunc=partial))
    diss, sb = artificialDissipationLLF(0., d3, dL, dR, sd)
    for i in range(3):
        out["llf_dL%d" % i], out["llf_dR%d" % i] = dL[i], dR[i]
    out["llf_diss"], out["llf_sb"] = A(diss), A(sb)
    out["llf_range_ndims"] = A([list(s[1]) for s in seen])
    save("extra.npz", **out)


if __name__ == "__main__":
    gen_ghost()
    gen_deriv()
    gen_term()
    gen_ode()
    gen_hjipde()
    gen_known()
    gen_extra()
