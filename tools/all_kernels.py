#!/usr/bin/env python3
"""Runs every kernel family of the library a few times at BASELINE sizes so that ONE `rocprofv3 --kernel-trace --stats` pass sees them
all (tools/kernel_table.py turns the stats into the per-kernel roofline table of DESIGN.md 4.4):
  201^3 fp64 Dubins: fused step with each scheme (pair kernel, seam / pre-pass kernels of the intended WENO5), split path with foreign
  callbacks (upwind_kernel, lf_split_end_kernel, rk_combine_kernel), termNormal / termReinit / termConvection (term_kernel),
  post-step min / NaN guard (HJIPDE_solve), the static step bound (alpha_bound_kernel); 51^3: direct kernel; 4096^2 ENO3 (C3);
  129^4 fp32 (C5).
usage: all_kernels.py [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 201
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2)
d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
op1 = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
sysn = L.DubinsVehicleRel(g, 1, 1)
y0 = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
# ---- fused path, every scheme
for deriv in (L.upwindFirstWENO5, L.upwindFirstWENO5Intended, L.upwindFirstENO3, L.upwindFirstENO2):
    sd = L.Bundle(dict(grid=g, hamFunc=sysn.hamiltonian, partialFunc=sysn.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=deriv))
    t, y = 0., y0
    for _ in range(reps):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op1, sd)
torch.cuda.synchronize()
# ---- split path: callbacks the package knows nothing about
x1, x2 = (torch.as_tensor(np.ascontiguousarray(np.broadcast_to(np.asarray(g.xs[i]), g.shape)), device="cuda") for i in (0, 1))
c3 = torch.as_tensor(np.ascontiguousarray(np.broadcast_to(np.cos(np.asarray(g.xs[2])), g.shape)), device="cuda")
s3 = torch.as_tensor(np.ascontiguousarray(np.broadcast_to(np.sin(np.asarray(g.xs[2])), g.shape)), device="cuda")
a0, a1 = (1 - c3).abs() + x2.abs(), s3.abs() + x1.abs()
ham = lambda t, data, p, sd: p[0] * (1 - c3) - p[1] * s3 - (p[0] * x2 - p[1] * x1 - p[2]).abs() + p[2].abs()   # noqa: E731
part = lambda t, data, dmin, dmax, sd, dim: [a0, a1, 2.0][dim]   # noqa: E731
sdf = L.Bundle(dict(grid=g, hamFunc=ham, partialFunc=part, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
t, y = 0., y0
for _ in range(max(2, reps // 3)):
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op1, sdf)
torch.cuda.synchronize()
# ---- the other terms
phi = y0.reshape(g.shape)
speed = 0.5 + 0.3 * torch.cos(phi)
vel = [0.7, -0.4 + 0.5 * torch.sin(phi), -0.2]
for fn, extra in ((L.termNormal, dict(speed=speed)), (L.termReinit, dict(initial=phi, subcell_fix_order=1)), (L.termConvection, dict(velocity=vel))):
    sdt = L.Bundle(dict(grid=g, derivFunc=L.upwindFirstWENO5, **extra))
    for _ in range(reps):
        fn(0., y0, sdt)
torch.cuda.synchronize()
# ---- HJIPDE_solve: post-step minimum with the target, NaN guard, static bound
sds = L.Bundle(dict(grid=g, hamFunc=sysn.hamiltonian, partialFunc=sysn.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5,
                    accuracy='veryHigh'))
try:
    L.HJIPDE_solve(torch.as_tensor(d0, device="cuda"), np.linspace(0, 0.02, 3), sds, 'minVWithTarget', L.Bundle(dict(targetFunction=d0, quiet=True, keepLast=1)))
except Exception as e:  # noqa: BLE001
    print("HJIPDE_solve leg skipped:", repr(e)[:200])
torch.cuda.synchronize()
# ---- 51^3: the direct kernel
g5 = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / 51)]]).T, 51 * np.ones((3, 1), dtype=np.int64), 2)
s5 = L.DubinsVehicleRel(g5, 1, 1)
sd5 = L.Bundle(dict(grid=g5, hamFunc=s5.hamiltonian, partialFunc=s5.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
t, y = 0., torch.as_tensor(L.shapeCylinder(g5, 2, np.zeros((3, 1)), .5).reshape(-1, 1), device="cuda")
for _ in range(3 * reps):
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op1, sd5)
torch.cuda.synchronize()
# ---- C3 and C5
m = 4096
g2 = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), m * np.ones((2, 1), dtype=np.int64), None, low_mem=True)
s2 = L.DoubleIntegrator(g2, 1.)
sd2 = L.Bundle(dict(grid=g2, hamFunc=s2.hamiltonian, partialFunc=s2.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO3))
xx = torch.linspace(-1, 1, m, dtype=torch.float64, device="cuda")
t, y = 0., (torch.sqrt(xx.reshape(m, 1) ** 2 + xx.reshape(1, m) ** 2) - 0.4).reshape(-1, 1)
for _ in range(reps):
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op1, sd2)
torch.cuda.synchronize()
del y
q = 129
g4 = L.createGrid(-np.pi * np.ones((4, 1)), np.pi * (1 - 2 / q) * np.ones((4, 1)), q * np.ones((4, 1), dtype=np.int64), [0, 1, 2, 3], low_mem=True)
s4 = L.DoublePendulum4D(g4, 1.0)
sd4 = L.Bundle(dict(grid=g4, hamFunc=s4.hamiltonian, partialFunc=s4.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstWENO5))
ax = torch.linspace(-np.pi, np.pi * (1 - 2 / q), q, dtype=torch.float32, device="cuda")
t, y = 0., (torch.sqrt(ax.reshape(q, 1, 1, 1) ** 2 + ax.reshape(1, q, 1, 1) ** 2 + ax.reshape(1, 1, q, 1) ** 2 + ax.reshape(1, 1, 1, q) ** 2) - 0.5).reshape(-1, 1)
for _ in range(max(2, reps // 2)):
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op1, sd4)
torch.cuda.synchronize()
print("all_kernels: done")
