#!/usr/bin/env python3
"""alpha_bound_kernel launch-shape sweep (child process per shape: the knobs are read once): wall time of hj_range_alpha_max at 201^3
(memset + kernel + 40-byte copy + synchronisation; the shapes differ in the kernel only)."""
import os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import levelsetpy_amd as L
    from levelsetpy_amd import _ffi
    from levelsetpy_amd.context import device_grid
    import bench
    g = bench.dubins_grid(L, 201, 201)
    reg = L.register_native_hamiltonian("bench_range", 3, bench.RANGE_SRC, nparams=1)
    dg = device_grid(g)
    dg.bind_stream()
    y = bench.device_sdf(torch, g, 0.5, ignore=(2,))
    keys = torch.zeros(8, dtype=torch.int64, device="cuda")
    par = _ffi.darr([0.7])
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    _ffi.check(dg.lib.hj_range_pass(dg.ctx, sid, reg.ham_id, par, C.c_void_p(y.data_ptr()), C.c_void_p(keys.data_ptr())))
    _ffi.check(dg.lib.hj_ctx_set_range_source(dg.ctx, C.c_void_p(keys.data_ptr())))
    am = (C.c_double * 4)()
    for _ in range(20):
        _ffi.check(dg.lib.hj_range_alpha_max(dg.ctx, reg.ham_id, par, am))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        _ffi.check(dg.lib.hj_range_alpha_max(dg.ctx, reg.ham_id, par, am))
    torch.cuda.synchronize()
    print("blocks %5s threads %5s: %.1f us per call, alpha max %s" % (os.environ.get("HJ_ALPHA_BLOCKS"), os.environ.get("HJ_ALPHA_THREADS"),
                                                                   1e6 * (time.perf_counter() - t0) / n, [round(am[d], 6) for d in range(3)]), flush=True)
    sys.exit(0)
for thr in (64, 256, 1024):
    for blk in (64, 128, 256, 512, 1024):
        env = dict(os.environ, HJ_ALPHA_BLOCKS=str(blk), HJ_ALPHA_THREADS=str(thr))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=120)
