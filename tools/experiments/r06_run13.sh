#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "cooperative" > gpurun_out/r06_t13.log 2>&1
rc=$?
tail -8 gpurun_out/r06_t13.log
[ $rc -ne 0 ] && exit $rc
cat > /tmp/coop_time.py <<'PY'
import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.context import DeviceGrid
for n in (31, 41, 51, 64):
    g = L.createGrid(np.array([[-.75, -1.25, -np.pi]]).T, np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T, n * np.ones((3, 1), dtype=np.int64), 2)
    d0 = torch.as_tensor(np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)), device="cuda")
    for coop in ("0", "1"):
        os.environ["HJ_COOP"] = coop
        dg = DeviceGrid(g, "float64"); dg.bind_stream()
        cur, nxt, w1 = d0.clone(), torch.empty_like(d0), torch.empty_like(d0)
        tout, dtout = C.c_double(), C.c_double()
        par = _ffi.darr([1., 1., 1., 2.]); sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
        t = 0.
        def one():
            global cur, nxt, t
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, par, t, 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(nxt), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur; t = float(tout.value)
        for _ in range(2000): one()
        best = 1e9
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2000): one()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 2000)
        print("n=%d HJ_COOP=%s  %.2f us per RK3 step  (%s)" % (n, coop, best * 1e6, dg.lib.hj_last_kernel(dg.ctx).decode()), flush=True)
PY
python /tmp/coop_time.py 2>&1 | grep "^n=" | tee gpurun_out/r06_coop_time.log
