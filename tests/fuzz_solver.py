#!/usr/bin/env python3
"""Randomised HJIPDE_solve runs (test infrastructure; not collected): the device path -- a tau interval as ONE native call with the
post-step operators fused into the last RK stage (hji_solver.py + hj_rk_integrate) -- against the same solve forced through the
step-by-step Python loop (HJ_HJIPDE_STEPWISE=1: odeCFL3 singleStep calls and array operators, the reference's structure), which
itself is what the suite pins to the oracle.  BITWISE: both paths run the same kernels on the same states.
    python tests/fuzz_solver.py [seconds] [seed]
Random: system (Dubins 3-D / double integrator 2-D), extents, compMethod (none, minVOverTime, maxVOverTime, minVWithV0, maxVWithV0,
minVWithL, maxVWithL, minWithZero), obstacles and targets (static), tau vectors, keepLast or store-all, NumPy or tensor input,
derivative scheme."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HJ_DIRECT_BELOW", "0")
import torch  # noqa: E402
import levelsetpy_amd as L  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 999
DERIV = {"ENO2": L.upwindFirstENO2, "ENO3": L.upwindFirstENO3, "WENO5_ASSHIPPED": L.upwindFirstWENO5, "WENO5": L.upwindFirstWENO5Intended}


def case(rng, k):
    three = rng.random() < 0.6
    if three:
        N = [int(rng.integers(9, 30)) for _ in range(3)]
        gmin, gmax, pd = [-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / N[2])], 2
        mksys = lambda g: L.DubinsVehicleRel(g, 1, 1)  # noqa: E731
    else:
        N = [int(rng.integers(12, 90)) for _ in range(2)]
        gmin, gmax, pd = [-1., -1.], [1., 1.], None
        mksys = lambda g: L.DoubleIntegrator(g, 1)  # noqa: E731
    g = L.createGrid(np.array(gmin).reshape(-1, 1), np.array(gmax).reshape(-1, 1), np.array(N, dtype=np.int64).reshape(-1, 1), pd)
    xs = [np.asarray(v) for v in g.xs]
    r0 = 0.45 * min(b - a for a, b in zip(gmin, gmax)) / 2
    if three:
        d0 = np.sqrt((xs[0] - 1.0) ** 2 + xs[1] ** 2) - r0 * 0.6 + 0.02 * rng.standard_normal(N)
    else:
        d0 = np.sqrt(xs[0] ** 2 + xs[1] ** 2) - r0 + 0.02 * rng.standard_normal(N)
    comp = str(rng.choice(["none", "minVOverTime", "maxVOverTime", "minVWithV0", "maxVWithV0", "minVWithL", "maxVWithL", "minWithZero"]))
    scheme = str(rng.choice(list(DERIV)))
    extra = dict(quiet=True)
    if rng.random() < 0.6:
        extra["keepLast"] = True
    if rng.random() < 0.35:
        extra["obstacleFunction"] = np.sqrt((xs[0] + 0.2) ** 2 + (xs[1] - 0.3) ** 2) - 0.25 * r0
    if comp.endswith("WithL") or rng.random() < 0.2:
        extra["targetFunction"] = d0 + 0.05 * np.cos(3 * xs[0])
    nt = int(rng.integers(2, 5))
    tau = np.concatenate([[0.0], np.cumsum(rng.uniform(0.004, 0.03, nt - 1))])
    as_tensor = rng.random() < 0.5
    outs = []
    for stepwise in ("0", "1"):
        os.environ["HJ_HJIPDE_STEPWISE"] = stepwise
        sysn = mksys(g)
        sd = L.Bundle(dict(grid=g, hamFunc=sysn.hamiltonian, partialFunc=sysn.dissipation, CoStateCalc=DERIV[scheme]))
        ex = L.Bundle({k2: (torch.as_tensor(v, device="cuda") if (as_tensor and isinstance(v, np.ndarray)) else v) for k2, v in extra.items()})
        din = torch.as_tensor(d0, device="cuda") if as_tensor else d0
        data, tau_out, _ = L.HJIPDE_solve(din, tau, sd, None if comp == "none" else comp, ex)
        a = data.detach().cpu().numpy() if hasattr(data, "detach") else np.asarray(data)
        outs.append((a, np.asarray(tau_out)))
    os.environ.pop("HJ_HJIPDE_STEPWISE", None)
    ok = outs[0][0].shape == outs[1][0].shape and np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    worst = float(np.abs(outs[0][0] - outs[1][0]).max()) if outs[0][0].shape == outs[1][0].shape else float("nan")
    print("%4d N=%-10s %-13s %-16s tau %d %s%s%s%s max|diff| %.1e %s" % (
        k, "x".join(map(str, N)), comp, scheme, nt, "keepLast " if extra.get("keepLast") else "storeAll ", "obst " if "obstacleFunction" in extra else "",
        "targ " if "targetFunction" in extra else "", "tensor" if as_tensor else "numpy", worst, "ok" if ok else "MISMATCH"), flush=True)
    return ok


t_end = time.time() + budget
k = 0
while time.time() < t_end:
    if not case(np.random.default_rng(seed0 + k), k):
        print("FAILED: replay with  python tests/fuzz_solver.py 1 %d" % (seed0 + k))
        sys.exit(1)
    k += 1
print("solver fuzz: %d cases ok in %.0f s" % (k, budget))
