#!/usr/bin/env python3
"""Randomised parity run of the fused HIP path against the CPU oracle (test infrastructure: imports oracle/; not collected by pytest).
    python tests/fuzz_parity.py [seconds] [seed] [share of range-dependent run-time Hamiltonian cases, default 0.2]
Every case: a random system (Dubins 3-D, double integrator 2-D, double pendulum 4-D fp32), random odd / even / prime extents (below,
at and above the tile sizes), random periodic axes, a random scheme, a random KERNEL forced through the environment knobs the library
reads when a context is created (tiled one-cell, pair, 4-D compile-time tile k, direct), random noisy initial data; two odeCFL3 /
odeCFL2 / odeCFL1 steps (or one termLaxFriedrichs with termRestrictUpdate) against the oracle at the tolerances of the suite
(ENO bit for bit on the native path; WENO5 1e-11; fp32 1e-4 of the fp64 oracle).  Prints one line per case and a summary;
exit code 1 on the first mismatch (with the case's seed, so that it can be replayed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HJ_DIRECT_BELOW", "0")
import torch  # noqa: E402
import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd.context import device_grid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
RANGE_SHARE = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2        # share of the cases that run a range-dependent run-time Hamiltonian
DERIV = {"ENO2": L.upwindFirstENO2, "ENO3": L.upwindFirstENO3, "WENO5_ASSHIPPED": L.upwindFirstWENO5, "WENO5": L.upwindFirstWENO5Intended}
KNOBS = ("HJ_PAIR", "HJ_FORCE_DIRECT", "HJ_TILE4_SEL", "HJ_PAIR4", "HJ_FLAT4", "HJ_MIN_CHUNK", "HJ_TILE_CELLS", "HJ_XP")


def mk(gmin, gmax, N, pd):
    g = L.createGrid(np.asarray(gmin, dtype=np.float64).reshape(-1, 1), np.asarray(gmax, dtype=np.float64).reshape(-1, 1),
                     np.asarray(N, dtype=np.int64).reshape(-1, 1), pd if pd else None)
    return g, O.Grid(gmin, gmax, [int(n) for n in N], list(pd) if pd else [])


_REG = {}


def case_range(rng, k):
    """A run-time Hamiltonian whose alpha reads the costate range (tests/test_gpu_round5.py's BurgersDrift), 2-D / 3-D / 4-D, through
    odeCFLn single steps (range pass + bound kernel with deltaT on the device + fused stages) against the oracle's general GLF protocol."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_round5 import BurgersDriftLocal, _burgers_src, CoupledBurgers, _coupled_src
    dim = int(rng.integers(2, 5))
    # half of the cases: alpha_0 / alpha_1 read the range of the OTHER dimension (the case in which LLF and LLLF differ)
    coupled = bool(rng.random() < 0.5)
    BurgersDrift, par0 = (CoupledBurgers, 0.6) if coupled else (BurgersDriftLocal, 0.7)
    # the dissipation variant: global range (GLF), per-node range in dimension i (LLF), per-node everywhere (LLLF)
    dk = str(rng.choice(["glf", "llf", "lllf"]))
    dfn = {"glf": L.artificialDissipationGLF, "llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[dk]
    N = [int(rng.integers(8, {2: 90, 3: 30, 4: 14}[dim])) for _ in range(dim)]
    pd = [d for d in range(dim) if rng.random() < 0.3]
    gmin, gmax = [-1.0] * dim, [1.0 - (2.0 / N[d] if d in pd else 0.0) for d in range(dim)]
    scheme = str(rng.choice(["ENO2", "ENO3", "WENO5_ASSHIPPED", "WENO5"]))
    g, og = mk(gmin, gmax, N, pd)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[dim - 1]) + 0.02 * rng.standard_normal(N)
    rk = (dim, coupled)
    if rk not in _REG:
        _REG[rk] = L.register_native_hamiltonian(("coupled_burgers_%dd" if coupled else "burgers_drift_%dd") % dim, dim,
                                                 (_coupled_src if coupled else _burgers_src)(dim), nparams=1)
    sys_ = _REG[rk](g, [par0], hamiltonian=lambda s, t, data, p, sd: BurgersDrift(g, par0).hamiltonian(t, data, p, sd),
                    dissipation=lambda s, t, data, dmin, dmax, sd, dm: BurgersDrift(g, par0).dissipation(t, data, dmin, dmax, sd, dm))
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=dfn, CoStateCalc=DERIV[scheme]))
    order = int(rng.integers(1, 4))
    ode = {1: L.odeCFL1, 2: L.odeCFL2, 3: L.odeCFL3}[order]
    oode = {1: O.ode_cfl_1, 2: O.ode_cfl_2, 3: O.ode_cfl_3}[order]
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    yo, t, to = d0.reshape(-1, 1), 0., 0.
    for _ in range(2):
        t, y, _ = ode(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        to, yo = oode(lambda tt, yy: O.term_lax_friedrichs(og, BurgersDrift(og, par0), scheme, tt, yy, diss=dk), [to, 10.], yo, 0.8, single_step=True)
    dg = device_grid(g, "float64")
    used = dg.lib.hj_last_kernel(dg.ctx).decode()
    got = y.cpu().numpy()
    scale = max(1.0, float(np.abs(yo).max()))
    err = float(np.abs(got - yo).max()) / scale
    if scheme.startswith("ENO"):
        bad = np.abs(got - yo) > 1e-11 * scale
        ok = float(bad.mean()) <= 2e-3 and err <= 1e-3 and abs(t - to) <= 1e-12 * to
    else:
        ok = err <= 1e-11 and abs(t - to) <= 1e-12 * to
    print("%4d range  N=%-18s pd=%-12s %-16s float64 order %d %-4s  kernel %-22s err %.2e %s" % (
        k, "x".join(map(str, N)), pd, scheme, order, dk, used, err, "ok" if ok else "MISMATCH"), flush=True)
    return ok, "range-%s%s:%s" % (dk, "-coupled" if coupled else "", used)


def case(rng, k):
    for kn in KNOBS:
        os.environ.pop(kn, None)
    if rng.random() < RANGE_SHARE:
        return case_range(rng, k)
    which = rng.choice(["dubins", "dint", "pend"], p=[0.5, 0.25, 0.25])
    scheme = str(rng.choice(list(DERIV)))
    if which == "dubins":
        N = [int(rng.integers(7, 48)) for _ in range(3)]
        pd = [d for d in range(3) if rng.random() < (0.8 if d == 2 else 0.2)]
        gmin, gmax = [-.75, -1.25, -np.pi], [3.25, 1.25, np.pi]
        mkp, mko = (lambda g: L.DubinsVehicleRel(g, 1, 1)), (lambda og: O.DubinsRel(og, 1, 1))
        dtype = "float32" if rng.random() < 0.3 else "float64"         # (round 5, late: single precision on the 2-D / 3-D systems too)
    elif which == "dint":
        N = [int(rng.integers(9, 160)) for _ in range(2)]
        pd = [d for d in range(2) if rng.random() < 0.2]
        gmin, gmax = [-1., -1.], [1., 1.]
        mkp, mko = (lambda g: L.DoubleIntegrator(g, 1)), (lambda og: O.DoubleIntegrator(og, 1))
        dtype = "float32" if rng.random() < 0.3 else "float64"
    else:
        N = [int(rng.integers(7, 15)), int(rng.integers(7, 20)), int(rng.integers(7, 20)), int(rng.choice([rng.integers(8, 40), rng.integers(34, 140)]))]
        pd = [0, 1, 2, 3] if rng.random() < 0.6 else [d for d in range(4) if rng.random() < 0.5]
        gmin, gmax = [-np.pi, -8., -np.pi, -8.], [np.pi, 8., np.pi, 8.]
        mkp, mko = (lambda g: L.DoublePendulum4D(g, 1.0)), (lambda og: O.DoublePendulum4D(og, 1.0))
        dtype = "float32" if rng.random() < 0.7 else "float64"
    gmax = [gmax[d] - (gmax[d] - gmin[d]) / N[d] if d in pd else gmax[d] for d in range(len(N))]
    # the kernel, forced
    kern = str(rng.choice(["default", "pair", "single", "direct", "tile4", "split", "flat4", "xp"]))
    if kern == "pair":
        os.environ["HJ_PAIR"] = "2"
        os.environ["HJ_FLAT4"] = "0"
    elif kern == "flat4":                   # round 6: the 4-D full-row kernel wherever the last axis fits (other grids: whatever "pair" runs)
        os.environ["HJ_PAIR"] = "2"
        os.environ["HJ_FLAT4"] = "2"
    elif kern == "xp":                      # round 6: the transposed march wherever a launch has that form (3-D Dubins)
        os.environ["HJ_PAIR"] = "2"
        os.environ["HJ_XP"] = "2"
    elif kern == "single":
        os.environ["HJ_PAIR"] = "0"
    elif kern == "direct":
        os.environ["HJ_FORCE_DIRECT"] = "1"
    elif kern == "tile4":
        os.environ["HJ_PAIR"] = "2"
        os.environ["HJ_TILE4_SEL"] = str(int(rng.integers(0, 3)))
        os.environ["HJ_FLAT4"] = "0"
    if rng.random() < 0.3:
        os.environ["HJ_MIN_CHUNK"] = str(int(rng.integers(1, 6)))
    g, og = mk(gmin, gmax, N, pd)
    d0 = O.shape_sphere(og, None, 0.45 * min(b - a for a, b in zip(gmin, gmax)) / 2) + 0.05 * rng.standard_normal(N)
    sysp, syso = mkp(g), mko(og)
    if kern == "split" and dtype == "float64":
        # foreign callables (here: lambdas around the system's own methods -- not recognised as native, by design): the split path --
        # derivative kernels, the callbacks on device arrays, the dissipation kernel -- through the generic integrator loop
        sd = L.Bundle(dict(grid=g, hamFunc=lambda *a: sysp.hamiltonian(*a), partialFunc=lambda *a: sysp.dissipation(*a),
                           dissFunc=L.artificialDissipationGLF, CoStateCalc=DERIV[scheme]))
    else:
        sd = L.Bundle(dict(grid=g, hamFunc=sysp.hamiltonian, partialFunc=sysp.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=DERIV[scheme]))
    order = int(rng.integers(1, 4))
    restrict = bool(rng.random() < 0.25)
    ode = {1: L.odeCFL1, 2: L.odeCFL2, 3: L.odeCFL3}[order]
    oode = {1: O.ode_cfl_1, 2: O.ode_cfl_2, 3: O.ode_cfl_3}[order]
    tdt = torch.float64 if dtype == "float64" else torch.float32
    y = torch.as_tensor(d0.reshape(-1) if restrict else d0.reshape(-1, 1), device="cuda").to(tdt)     # (termRestrictUpdate: an (N,) vector)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    if restrict:
        sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sd, positive=int(rng.integers(0, 2))))
        func, data = L.termRestrictUpdate, sdr

        def oterm(tt, yy):
            yd, sb = O.term_lax_friedrichs(og, syso, scheme, tt, yy)
            return (np.maximum(yd, 0) if sdr.positive else np.minimum(yd, 0)), sb
    else:
        func, data = L.termLaxFriedrichs, sd

        def oterm(tt, yy):
            return O.term_lax_friedrichs(og, syso, scheme, tt, yy)
    yo, t, to = d0.reshape(-1, 1), 0., 0.
    for _ in range(2):
        t, y, _ = ode(func, [t, 10.], y, op, data)
        to, yo = oode(oterm, [to, 10.], yo, 0.8, single_step=True)
    dg = device_grid(g, dtype)
    used = dg.lib.hj_last_kernel(dg.ctx).decode()
    if kern == "split" and dtype == "float64":
        used = "split"
    got = y.double().cpu().numpy().reshape(-1, 1)
    scale = max(1.0, float(np.abs(yo).max()))
    err = float(np.abs(got - yo).max()) / scale
    if dtype == "float32" and scheme.startswith("ENO"):
        # an fp32 product against the fp64 oracle: stencil selections flip where the candidates tie within fp32 rounding (noisy data):
        # masked, as SURVEY 8(c) allows -- at most 1 % of the cells beyond the fp32 tolerance, none by more than a few per cent
        bad = np.abs(got - yo) > 3e-4 * scale
        ok = float(bad.mean()) <= 0.01 and err <= 0.05 and abs(t - to) <= 1e-5 * to
    elif dtype == "float32":
        ok = err <= 3e-4 and abs(t - to) <= 1e-5 * to
    elif scheme.startswith("ENO") and which != "pend" and used != "split":
        ok = np.array_equal(got, yo) and t == to          # the reference's systems on the fused path: bit for bit
    elif scheme.startswith("ENO"):
        # the build-defined 4-D system evaluates its drift in another order than the oracle's NumPy expression (last-bit differences in
        # H): the ENO selections can flip where candidates tie within rounding -- masked as in the suite's multi-step ENO comparisons
        bad = np.abs(got - yo) > 1e-11 * scale
        ok = float(bad.mean()) <= 2e-3 and err <= 1e-3 and abs(t - to) <= 1e-13 * to
    else:
        ok = err <= 1e-11 and abs(t - to) <= 1e-13 * to
    print("%4d %-6s N=%-18s pd=%-12s %-16s %-7s order %d%s kernel %-22s err %.2e %s" % (
        k, which, "x".join(map(str, N)), pd, scheme, dtype, order, " clamp" if restrict else "      ", used, err, "ok" if ok else "MISMATCH"), flush=True)
    return ok, used


t_end = time.time() + budget
k, used_all = 0, {}
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + k)
    ok, used = case(rng, k)
    used_all[used] = used_all.get(used, 0) + 1
    if not ok:
        print("FAILED: replay with  python tests/fuzz_parity.py 1 %d" % (seed0 + k))
        sys.exit(1)
    k += 1
print("fuzz: %d cases ok in %.0f s; kernels: %s" % (k, budget, used_all))
