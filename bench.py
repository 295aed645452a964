#!/usr/bin/env python3
"""Headline benchmark: grid-cell RK-substep updates / second of the HJI hot path
(odeCFL3 -> termLaxFriedrichs -> upwindFirstWENO5 -> artificialDissipationGLF -> ghost cells)
on the Dubins-relative 3-D problem, fp64 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one odeCFL3 time step = 3 fused RK substeps over the whole grid.  Inputs are
resident in HBM before the timed region.  N > 1 (launched by torch.distributed.run): the grid is
slab-decomposed along axis 0, every rank owns an n^3 slab of an (N*n) x n x n grid ("weak"
scaling) and exchanges 3 ghost planes with its neighbours per substep over RCCL.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (algorithmic bytes / measured kernel
time vs the 8 TB/s HBM peak) and "cpu_baseline" (the NumPy oracle timed on this host's cores on a
bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the slab stepper uses three streams (+ RCCL's): give each its own hardware queue, otherwise streams
# that share a queue serialise (must be set before HIP initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

SPINUP_STEPS = int(os.environ.get("HJ_BENCH_SPINUP", "300"))   # untimed clock ramp before the W warm-up steps
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_SUBSTEP = {"float64": 64.0 / 3.0, "float32": 32.0 / 3.0}   # 8 words / RK3 step


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=201, help="grid points per axis (per rank along axis 0)")
    ap.add_argument("--scheme", default="WENO5_ASSHIPPED", choices=["WENO5", "WENO5_ASSHIPPED", "ENO3", "ENO2"],
                    help="WENO5_ASSHIPPED = what the reference's upwindFirstWENO5 computes (parity-pinned; "
                         "headline); WENO5 = the intended nonlinear scheme (reported in 'also')")
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=101, help="grid size of the CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=16)
    ap.add_argument("--extra-schemes", default="WENO5",
                    help="comma list of further schemes timed after the headline (reported in 'also')")
    return ap.parse_args()


def dubins_grid(L, n0, n):
    """C2 geometry; n0 planes along axis 0 (dx kept, max extended for slabs)."""
    dx0 = 4.0 / (n - 1)
    gmin = np.array([[-.75, -1.25, -np.pi]]).T
    gmax = np.array([[-.75 + dx0 * (n0 - 1), 1.25, np.pi * (1 - 2 / n)]]).T
    N = np.array([[n0], [n], [n]], dtype=np.int64)
    return L.createGrid(gmin, gmax, N, 2, low_mem=True), gmin, gmax


def time_steps(torch, dg, lib, sid, ham, par, bufs, steps, warmup, world):
    import torch.distributed as dist
    cur, nxt, w0, w1 = bufs
    tout, dtout = C.c_double(), C.c_double()
    parv = (C.c_double * 4)(*par)
    t = 0.0

    def one(cur, nxt, t):
        # three arrays are enough for RK3: the first stage buffer doubles as the output (stage 3
        # reads w1 and y0=cur only), which keeps the 201^3 working set (195 MB) inside the 256 MB
        # Infinity Cache
        rc = lib.hj_rk_step(dg.ctx, 3, sid, ham, parv, t, 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt),
                            dg.ptr(nxt if w0 is None else w0), dg.ptr(w1), C.byref(tout), C.byref(dtout))
        if rc != 0:
            raise RuntimeError(lib.hj_last_error().decode())
        return nxt, cur, float(tout.value)

    # device spin-up (untimed, reported as config.spinup_steps): the GPU's clocks need a few tens of
    # milliseconds of load to reach their steady state; without it a short --steps run measures the ramp
    for _ in range(SPINUP_STEPS):
        cur, nxt, t = one(cur, nxt, t)
    for _ in range(warmup):
        cur, nxt, t = one(cur, nxt, t)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        cur, nxt, t = one(cur, nxt, t)
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    return wall, dev_ms, cur, t


def cpu_baseline(n, steps, scheme):
    """The NumPy oracle (a port of the reference's array path) on one core of this host."""
    from oracle import hj_oracle as O
    og = O.Grid([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n)], [n] * 3, [2])
    osys = O.DubinsRel(og, 1, 1)
    y = O.shape_cylinder(og, 2, None, .5).reshape(-1, 1)
    term = lambda tt, yy: O.term_lax_friedrichs(og, osys, scheme, tt, yy)  # noqa: E731
    t = 0.
    t0 = time.perf_counter()
    for _ in range(steps):
        t, y = O.ode_cfl_3(term, [t, 10.], y, 0.8, single_step=True)
    sec = time.perf_counter() - t0
    return {"value": n ** 3 * 3 * steps / sec, "unit": "cell-substeps/s", "cores": 1, "kind": "port",
            "sample": "%d RK3 steps of Dubins-relative %d^3 %s+GLF fp64 with oracle/hj_oracle.py (NumPy, "
                      "single-threaded) in %.1f s; host has %d cores" % (steps, n, scheme, sec, os.cpu_count())}


def measured_traffic(n, scheme, dtype, world):
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, FETCH_SIZE
    doubled per MI355X_MICROARCH.md's gfx950 correction); None when no pass exists for this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tab = json.load(f)
        rec = tab.get("%d/%s/%s" % (n, scheme, dtype))
        return rec["bytes_per_launch"] if (rec and world == 1) else None
    except Exception:  # noqa: BLE001
        return None


def main():
    a = parse()
    # stdout carries exactly one JSON line: libraries that print banners to fd 1 (RCCL's version header at
    # communicator creation, for one) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import levelsetpy_amd as L
    from levelsetpy_amd import _ffi
    from levelsetpy_amd.context import DeviceGrid

    n = a.n
    if world > 1 or os.environ.get("HJ_BENCH_FORCE_SLAB"):
        from levelsetpy_amd import dist as hjdist
        if world == 1:      # rehearsal of the N > 1 leg on one GPU (a single slab, no neighbours)
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        result = hjdist.bench_slab(a, rank, world)
        wall, dev_ms, cells, sched = result["wall"], result["dev_ms"], result["cells"], result["parallelism"]
        also = {"slab_check_max_abs_diff": result.get("slab_check_max_abs_diff")}
    else:
        g, gmin, gmax = dubins_grid(L, n, n)
        dg = DeviceGrid(g, a.dtype)
        dg.bind_stream()
        lib = dg.lib
        d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
        par = [1.0, 1.0, 1.0, 2.0]
        ham = _ffi.HAM_DUBINS_REL

        def run(scheme):
            bufs = [dg.to_device(d0).clone(), dg.empty(), (dg.empty() if os.environ.get("HJ_BENCH_4BUF") else None), dg.empty()]
            return time_steps(torch, dg, lib, _ffi.SCHEME_IDS[scheme], ham, par, bufs, a.steps, a.warmup, 1)

        wall, dev_ms, cur, t_end = run(a.scheme)
        assert bool(torch.isfinite(cur).all()), "non-finite state after the timed steps"
        cells = n ** 3
        sched = "single"
        also = {}
        for s in [x for x in a.extra_schemes.split(",") if x and x != a.scheme]:
            w2, d2, _, _ = run(s)
            also[s] = {"value": cells * 3 * a.steps / w2, "ms_per_step": 1e3 * w2 / a.steps,
                       "roofline_frac": cells * 3 * a.steps / w2 * BYTES_PER_SUBSTEP[a.dtype] / 1e9 / HBM_PEAK_GBS}

    if world > 1:
        import torch.distributed as dist
        tw = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    value = cells * world * 3 * a.steps / wall
    per_gpu = value / world
    bps = BYTES_PER_SUBSTEP[a.dtype]
    # dominant kernel = the fused substep; its mean launch duration from HIP events over the timed
    # region (3 launches per step, back to back on the ctx stream)
    kern_ms = dev_ms / (a.steps * 3)
    achieved = cells * bps / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": "grid-cell RK-substep updates/sec, Dubins-3D HJI %d^3 fp64" % n if a.dtype == "float64"
                  else "grid-cell RK-substep updates/sec, Dubins-3D HJI %d^3 fp32" % n,
        "value": value, "unit": "cell-substeps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * wall / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if a.dtype == "float64" else "f32", "data": "synthetic",
        "config": {"workload": "Dubins-relative (air3D) 3-D HJI, %s x %d x %d grid, %s + GLF, odeCFL3 "
                               "(factorCFL 0.8), cylinder r=0.5 initial data" %
                               (("%d" % n) if world == 1 else ("%dx%d" % (world, n)), n, n, a.scheme),
                   "scheme": a.scheme, "parallelism": sched, "substeps_per_step": 3,
                   "spinup_steps": SPINUP_STEPS},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(n, a.scheme, a.dtype, world),
                     "kernel": "fused_substep_kernel", "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": cells * bps},
        "per_gpu_value": per_gpu,
    }
    if also:
        out["also"] = also
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_n, a.cpu_steps, a.scheme)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or os.environ.get("HJ_BENCH_FORCE_SLAB"):
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
