import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid
n = int(os.environ.get("N", "201"))
g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)
d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
dg = DeviceGrid(g); dg.bind_stream()
y = dg.to_device(d0); a = dg.empty()
par = _ffi.darr([1., 1., 1., 2.])
def sub(scheme, src, y0, out, p0=0, p1=n, stage=1):
    _ffi.check(dg.lib.hj_rk_substep(dg.ctx, scheme, 0, par, 0., stage, 2e-3, 0, dg.ptr(src), dg.ptr(y0) if y0 is not None else None, dg.ptr(out), 1, p0, p1))
sub(3, y, None, a)
for scheme, name in ((3, "asshipped"), (2, "weno5"), (1, "eno3")):
    e1 = torch.empty(3, dtype=torch.float64, device="cuda"); e2 = torch.empty_like(e1)
    _ffi.check(dg.lib.hj_max_d1sq(dg.ctx, dg.ptr(a), dg.ptr(e1))); _ffi.check(dg.lib.hj_max_d1sq(dg.ctx, dg.ptr(a), dg.ptr(e2)))
    outs = []
    for trial in range(4):
        o = torch.zeros_like(a); sub(scheme, a, y, o, stage=2); dg.sync(); outs.append(o)
    d = max(float((outs[0]-outs[k]).abs().max()) for k in (1, 2, 3))
    nd = int(((outs[0]-outs[1]).abs() > 0).sum())
    print("%s cfg NT=%s R=%s: repeated-launch max diff %.3e (%d cells), eps equal %s" % (name, os.environ.get("HJ_NT"), os.environ.get("HJ_R"), d, nd, bool(torch.equal(e1, e2))), flush=True)
    if scheme == 2:
        _ffi.check(dg.lib.hj_ctx_set_weno_eps_source(dg.ctx, dg.ptr(e1)))
        outs = []
        for trial in range(3):
            o = torch.zeros_like(a); sub(scheme, a, y, o, stage=2); dg.sync(); outs.append(o)
        print("   weno5 with a fixed eps source: max diff %.3e" % max(float((outs[0]-outs[k]).abs().max()) for k in (1, 2)))
        _ffi.check(dg.lib.hj_ctx_set_weno_eps_source(dg.ctx, None))
