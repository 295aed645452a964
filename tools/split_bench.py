#!/usr/bin/env python3
"""Foreign-callback Dubins (hamFunc / partialFunc the package knows nothing about) at 201^3 on device tensors:
ms per odeCFL3 step with the split path's device kernels (hj_lf_split_begin / _end, hj_rk_combine) and with
HJ_SPLIT_KERNELS=0 (stock torch elementwise launches, round-1 behaviour), next to the fused native path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 201
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T,
                 n * np.ones((3, 1), dtype=np.int64), 2)
d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
x1, x2 = (torch.as_tensor(np.ascontiguousarray(np.broadcast_to(np.asarray(g.xs[i]), g.shape)), device="cuda") for i in (0, 1))
c3 = torch.as_tensor(np.ascontiguousarray(np.broadcast_to(np.cos(np.asarray(g.xs[2])), g.shape)), device="cuda")
s3 = torch.as_tensor(np.ascontiguousarray(np.broadcast_to(np.sin(np.asarray(g.xs[2])), g.shape)), device="cuda")
a0 = (1 - c3).abs() + x2.abs()
a1 = s3.abs() + x1.abs()


def ham(t, data, p, sd):
    return p[0] * (1 - c3) - p[1] * s3 - (p[0] * x2 - p[1] * x1 - p[2]).abs() + p[2].abs()


def part(t, data, dmin, dmax, sd, dim):
    return [a0, a1, 2.0][dim]


sd = L.Bundle(dict(grid=g, hamFunc=ham, partialFunc=part, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
sdn = L.Bundle(dict(grid=g, hamFunc=L.DubinsVehicleRel(g, 1, 1).hamiltonian, partialFunc=None, dissFunc=L.artificialDissipationGLF,
                    CoStateCalc=L.upwindFirstWENO5))
sysn = L.DubinsVehicleRel(g, 1, 1)
sdn = L.Bundle(dict(grid=g, hamFunc=sysn.hamiltonian, partialFunc=sysn.dissipation, dissFunc=L.artificialDissipationGLF,
                    CoStateCalc=L.upwindFirstWENO5))
op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
for name, s in (("foreign callbacks (split path)", sd), ("native (fused path)", sdn)):
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    t = 0.
    for _ in range(3):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, s)
    torch.cuda.synchronize()
    k = 10
    t0 = time.perf_counter()
    for _ in range(k):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, s)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / k
    print("%-34s HJ_SPLIT_KERNELS=%s  n=%d  %.3f ms per odeCFL3 step  (%.3e cell-substeps/s)" %
          (name, os.environ.get("HJ_SPLIT_KERNELS", "1"), n, ms, n ** 3 * 3 / (ms * 1e-3)), flush=True)
