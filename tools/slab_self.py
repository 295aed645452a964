#!/usr/bin/env python3
"""One GPU, world_size 1, backend nccl (RCCL): periodic axis 0 closed through a self send/recv.
Checks the RCCL halo path of levelsetpy_amd.dist against the in-kernel periodic wrap and times it."""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scheme = sys.argv[2] if len(sys.argv) > 2 else "WENO5_ASSHIPPED"
steps = 20
gmin = np.array([[-2., -1.25, -np.pi]]).T
gmax = np.array([[2. * (1 - 2 / n), 1.25, np.pi * (1 - 2 / n)]]).T
g = L.createGrid(gmin, gmax, n * np.ones((3, 1), dtype=np.int64), [0, 2], low_mem=True)
d0 = np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)) + 0.1 * np.sin(3 * np.asarray(g.xs[0]))
for self_ex in (False, True):
    slab = SlabDecomposition(n, 1, 0, True, self_exchange=self_ex)
    be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS[scheme], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.])
    integ = SlabIntegrator(slab, be, [float(v) for v in np.asarray(g.dx).ravel()], 3, 0.8, needs_eps=(scheme == "WENO5"))
    integ.set_state(torch.as_tensor(d0, device="cuda"))
    t = 0.
    for _ in range(3):
        t, _ = integ.step(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        t, _ = integ.step(t)
    enq = time.perf_counter() - t0
    torch.cuda.synchronize(); sec = time.perf_counter() - t0
    res = integ.state().clone()
    print("   CPU enqueue time %.3f ms/step" % (1e3 * enq / steps))
    print("self_exchange=%s: %.3f ms/step (%.3e cell-substeps/s), halo_lo=%s" % (self_ex, 1e3 * sec / steps, n ** 3 * 3 * steps / sec, slab.halo_lo), flush=True)
    if self_ex:
        print("max |rccl-ring - in-kernel wrap| = %.3e" % float((res - ref).abs().max()))
    else:
        ref = res
# native steppers (ncclSend/ncclRecv inside the C library), same self ring: per-substep 3-plane exchange
# and the deep-halo variant (one 9-plane exchange per step)
from levelsetpy_amd.dist import NativeSlabStepper
steps = 100
for deep in (False, True):
    slab = SlabDecomposition(n, 1, 0, True, self_exchange=True)
    nat = NativeSlabStepper(g, slab, _ffi.SCHEME_IDS[scheme], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.],
                            [float(v) for v in np.asarray(g.dx).ravel()], deep=deep)
    nat.set_state(torch.as_tensor(d0, device="cuda"))
    t = 0.
    for _ in range(23):
        t, _ = nat.step(t)
    chk = nat.state().clone()
    for _ in range(10):
        t, _ = nat.step(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        t, _ = nat.step(t)
    enq = time.perf_counter() - t0
    torch.cuda.synchronize(); sec = time.perf_counter() - t0
    print("native self ring deep=%s: %.3f ms/step (%.3e cell-substeps/s), CPU enqueue %.3f ms/step" % (deep, 1e3 * sec / steps, n ** 3 * 3 * steps / sec, 1e3 * enq / steps))
    print("   max |native-ring - in-kernel wrap| after 23 steps = %.3e" % float((chk - ref).abs().max()))
    nat.close()
dist.destroy_process_group()
