import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid

def run(n, flag, reps=1):
    g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n[2])]]).T,
                     np.array(n, dtype=np.int64).reshape(-1, 1), 2)
    rng = np.random.default_rng(1)
    data = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5) + 0.02 * rng.standard_normal(n)
    os.environ["HJ_PAIR"] = flag
    dg = DeviceGrid(g, "float64"); dg.bind_stream()
    y = dg.to_device(data)
    outs = []
    for _ in range(reps):
        out = torch.full(n, float("nan"), dtype=torch.float64, device="cuda")
        _ffi.check(dg.lib.hj_rk_substep(dg.ctx, _ffi.ENO2, _ffi.HAM_DUBINS_REL, _ffi.darr([1., 1., 1., 2.]), 0., _ffi.STAGE_EULER, 1e-3, 0,
                                        dg.ptr(y), None, dg.ptr(out), 3, 0, n[0]))
        dg.sync()
        outs.append(out.cpu().numpy())
    return outs

os.environ["HJ_DEBUG"] = "1"
for n in [(59, 60, 44), (54, 49, 63), (51, 51, 51), (21, 21, 21), (40, 64, 64)]:
    ref = run(n, "0")[0]
    got = run(n, "2", reps=3)
    for k, o in enumerate(got):
        d = np.abs(ref - o)
        bad = np.argwhere(~(d == 0))
        print(n, "rep", k, "differing:", len(bad), "nan:", int(np.isnan(o).sum()))
        if len(bad):
            print("   planes", np.unique(bad[:, 0]), "rows", np.unique(bad[:, 1])[:12], "cols", np.unique(bad[:, 2])[:24])
