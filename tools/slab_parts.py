#!/usr/bin/env python3
"""Latency of the pieces of a slab substep on one GPU: RCCL self halo exchange, edge kernels."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper, HALO
n = int(sys.argv[1]) if len(sys.argv) > 1 else 201
g = L.createGrid(np.array([[-2., -1.25, -np.pi]]).T, np.array([[2. * (1 - 2 / n), 1.25, np.pi * (1 - 2 / n)]]).T,
                 n * np.ones((3, 1), dtype=np.int64), [0, 2], low_mem=True)
slab = SlabDecomposition(n, 1, 0, True, self_exchange=True)
nat = NativeSlabStepper(g, slab, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.],
                        [float(v) for v in np.asarray(g.dx).ravel()])
lib, ctx = nat.dg.lib, nat.dg.ctx
cur, nxt = nat.buf["cur"], nat.buf["nxt"]
cur.normal_()
def timeit(name, fn, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); print("%-40s %.1f us" % (name, 1e6 * (time.perf_counter() - t0) / k), flush=True)
ip = lambda b: C.c_void_p(b[HALO:].data_ptr())
timeit("hj_halo_exchange (4 x %.2f MB self)" % (3 * n * n * 8 / 1e6), lambda: _ffi.check(lib.hj_halo_exchange(ctx, ip(cur))))
par = _ffi.darr([1., 1., 1., 2.])
def sub(p0, p1):
    _ffi.check(lib.hj_rk_substep(ctx, 3, 0, par, 0., 1, 1e-3, 0, ip(cur), None, ip(nxt), 1, p0, p1))
timeit("edge kernel planes [0,3)", lambda: sub(0, 3))
timeit("edge kernel planes [n-3,n)", lambda: sub(n - 3, n))
timeit("interior planes [3,n-3)", lambda: sub(3, n - 3))
timeit("all planes", lambda: sub(0, n))
dist.destroy_process_group()
