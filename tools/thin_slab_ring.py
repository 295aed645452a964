#!/usr/bin/env python3
"""One rank's slab of the 513^3 grid at N = 2, 4, 8 as a PERIODIC ring through a real RCCL self send/recv on one GPU
(a (513/N) x 513 x 513 grid, axis 0 periodic): both native schedules with their real launches, streams, events and
RCCL calls -- only the link is missing (the 'exchange' is a device-local copy).
"plain" = the same slab grid without the ring (axis 0 wraps inside the kernel, no streams / exchange): the ceiling for any schedule.
usage: thin_slab_ring.py [n] [worlds, e.g. 8 or 2,4,8] [schedules: deep,sub,plain] [C4|C5]
C5: the slabs of the 129^4 fp32 double-pendulum grid (all axes periodic) instead -- n = 129: 65-, 33- and 17-plane slabs of 129^3-cell planes."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
steps = 30
worlds = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 8]
scheds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["sub", "deep"]
wl = sys.argv[4] if len(sys.argv) > 4 else "C4"
for world in worlds:
    n0 = (n + world - 1) // world
    if wl == "C5":
        gmin = np.array([[-np.pi, -8, -np.pi, -8]]).T
        gmax = np.array([[np.pi * (1 - 2 / n0), 8 * (1 - 2 / n), np.pi * (1 - 2 / n), 8 * (1 - 2 / n)]]).T
        g = L.createGrid(gmin, gmax, np.array([[n0], [n], [n], [n]], dtype=np.int64), [0, 1, 2, 3], low_mem=True)
        xs = [torch.as_tensor(np.asarray(v).ravel(), device="cuda", dtype=torch.float32) for v in g.vs]
        shp = lambda d: [-1 if k == d else 1 for k in range(4)]  # noqa: E731
        d0 = (sum((xs[d] ** 2).reshape(shp(d)) for d in range(4)).sqrt() - 0.5).contiguous()
        ham, par, dtype, esz_step = _ffi.HAM_DOUBLE_PENDULUM, [1., 0., 0., 0.], "float32", 32
    else:
        g = L.createGrid(np.array([[-2., -1.25, -np.pi]]).T, np.array([[2. * (1 - 2 / n0), 1.25, np.pi * (1 - 2 / n)]]).T,
                         np.array([[n0], [n], [n]], dtype=np.int64), [0, 2], low_mem=True)
        d0 = torch.as_tensor(np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)), device="cuda")
        ham, par, dtype, esz_step = _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.], "float64", 64
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    for sched in scheds:
        deep = sched == "deep"
        if deep and n0 < 18:
            print("N=%d: %3d-plane slab deep         -- too thin for the deep-halo stepper" % (world, n0), flush=True)
            continue
        slab = SlabDecomposition(n0, 1, 0, True, self_exchange=sched != "plain")
        st = NativeSlabStepper(g, slab, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], ham, par, dxs, dtype, deep=deep)
        st.set_state(d0)
        t = 0.0
        for _ in range(5):
            t, _ = st.step(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            t, _ = st.step(t)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / steps
        cells = n0 * n ** (3 if wl == "C5" else 2)
        print("%s N=%d: %3d-plane slab %-12s %.3f ms/step  frac %.3f   (ideal = undivided/N)" %
              (wl, world, n0, {"deep": "deep", "sub": "per-substep", "plain": "plain (no ring)"}[sched], ms, cells * esz_step / (ms * 1e-3) / 8e12), flush=True)
        st.close()
dist.destroy_process_group()
