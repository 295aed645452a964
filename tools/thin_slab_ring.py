#!/usr/bin/env python3
"""One rank's slab of the 513^3 grid at N = 2, 4, 8 as a PERIODIC ring through a real RCCL self send/recv on one GPU
(a (513/N) x 513 x 513 grid, axis 0 periodic): both native schedules with their real launches, streams, events and
RCCL calls -- only the link is missing (the 'exchange' is a device-local copy).
"plain" = the same slab grid without the ring (axis 0 wraps inside the kernel, no streams / exchange): the ceiling for any schedule.
usage: thin_slab_ring.py [n] [worlds, e.g. 8 or 2,4,8] [schedules: deep,sub,plain]"""
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
for world in worlds:
    n0 = (n + world - 1) // world
    g = L.createGrid(np.array([[-2., -1.25, -np.pi]]).T, np.array([[2. * (1 - 2 / n0), 1.25, np.pi * (1 - 2 / n)]]).T,
                     np.array([[n0], [n], [n]], dtype=np.int64), [0, 2], low_mem=True)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    d0 = torch.as_tensor(np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)), device="cuda")
    for sched in scheds:
        deep = sched == "deep"
        slab = SlabDecomposition(n0, 1, 0, True, self_exchange=sched != "plain")
        st = NativeSlabStepper(g, slab, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.], dxs, deep=deep)
        st.set_state(d0)
        t = 0.0
        for _ in range(5):
            t, _ = st.step(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            t, _ = st.step(t)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / steps
        cells = n0 * n * n
        print("N=%d: %3d-plane slab %-12s %.3f ms/step  frac %.3f   (ideal = undivided/N)" %
              (world, n0, {"deep": "deep", "sub": "per-substep", "plain": "plain (no ring)"}[sched], ms, cells * 64 / (ms * 1e-3) / 8e12), flush=True)
        st.close()
dist.destroy_process_group()
