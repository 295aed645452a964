#!/usr/bin/env python3
"""Experiment: capture two slab RK3 steps (kernels + RCCL halo send/recv) in one HIP graph."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import levelsetpy_amd as L
from levelsetpy_amd import _ffi
from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 201
g = L.createGrid(np.array([[-2., -1.25, -np.pi]]).T, np.array([[2. * (1 - 2 / n), 1.25, np.pi * (1 - 2 / n)]]).T,
                 n * np.ones((3, 1), dtype=np.int64), [0, 2], low_mem=True)
d0 = np.asarray(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)) + 0.1 * np.sin(3 * np.asarray(g.xs[0]))
slab = SlabDecomposition(n, 1, 0, True, self_exchange=True)
be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.])
integ = SlabIntegrator(slab, be, [float(v) for v in np.asarray(g.dx).ravel()], 3, 0.8)
integ.set_state(torch.as_tensor(d0, device="cuda"))
t = 0.
for _ in range(4):
    t, _ = integ.step(t)
torch.cuda.synchronize()
ref_state = integ.state().clone()
# eager reference for 4 more steps
for _ in range(4):
    t, _ = integ.step(t)
torch.cuda.synchronize()
eager = integ.state().clone()
integ.set_state(ref_state)
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    be.dg.bind_stream()
    with torch.cuda.graph(gr, stream=s):
        be.dg.bind_stream()
        integ.step(0.); integ.step(0.)
torch.cuda.current_stream().wait_stream(s)
be.dg.bind_stream()
integ.set_state(ref_state)
gr.replay(); gr.replay()
torch.cuda.synchronize()
print("graph vs eager after 4 steps: max diff %.3e" % float((integ.state() - eager).abs().max()))
steps = 40
t0 = time.perf_counter()
for _ in range(steps // 2):
    gr.replay()
enq = time.perf_counter() - t0
torch.cuda.synchronize(); sec = time.perf_counter() - t0
print("graph replay: %.3f ms/step (%.3e cell-substeps/s), CPU enqueue %.3f ms/step" % (1e3 * sec / steps, n ** 3 * 3 * steps / sec, 1e3 * enq / steps))
dist.destroy_process_group()
