#!/usr/bin/env python3
"""Headline benchmark: grid-cell RK-substep updates / second of the HJI hot path
(odeCFL3 -> termLaxFriedrichs -> upwindFirstWENO5 -> artificialDissipationGLF -> ghost cells)
on the Dubins-relative 3-D problem, fp64 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one odeCFL3 time step = 3 RK substeps over the whole grid; inputs are resident in HBM
before the timed region; a window of exactly K steps is timed between synchronisations, R times (--repeats,
default 51), and `value` is the MEDIAN window (inter-quartile range and min/max in "repeats").

N = 1 (default): BASELINE C2, the 201^3 grid.  The same run also times, each with its own spin-up and
reported under "also" with its own roofline fraction: the intended WENO5 arithmetic at 201^3, the
single-GPU 513^3 grid (C4's N = 1 point), C3 (double integrator 4096^2, ENO3; "C3 fast": the opt-in lean ENO arithmetic) and C5 (double
pendulum 129^4 fp32, all axes periodic) -- C3 / C5 / 513^3 / WENO5 with their own PMC traffic passes --, the headline with the CFL reduction kept
in every launch, the drop-in API legs, a run-time Hamiltonian, the split path, and a Hamiltonian whose alpha reads the costate range (fused
against split, plus the local Lax-Friedrichs variants).
`--gpus N --plan-only`: no GPU is touched -- every rank's slab, stepper, launches, halo bytes and predicted ms/step of the N-rank leg.
N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE C4 -- the 513^3 grid
slab-decomposed along axis 0 over the N ranks (65/64-plane slabs at N = 8), halo planes exchanged with
ncclSend/ncclRecv over RCCL, STRONG scaling (`--global-n 513`, the default for N > 1; every N
integrates the same grid, so N = 1 with `--global-n 513` is the single-domain 513^3 number).
`--global-n 0` selects the weak-scaling leg instead (every rank owns a 201^3 slab of an (N*201) x 201 x 201
grid).  Before it is timed, a decomposition has to reproduce the single-domain result on the hardware
it runs on ("slab_check_max_abs_diff").

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (algorithmic bytes / measured kernel time vs
the 8 TB/s HBM peak; "traffic" = fabric bytes per launch from two rocprofv3 --pmc child passes of this very
run, taken before the timed legs) and "cpu_baseline" (the NumPy oracle timed on this host's cores on a bounded sample
of the same workload; N = 1 only).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import subprocess
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
REF_IMPORT = "reference import (BASELINE.md s2, build container, 1 core): 6.6e5 at 51^3, 3.7e5 at 101^3, 1.57e5 at 201^3"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=51,
                    help="the window of EXACTLY K timed steps (synchronised on both sides) is repeated this many times; "
                         "value = the median window, the inter-quartile range is reported beside it")
    ap.add_argument("--n", type=int, default=201, help="grid points per axis of the single-GPU / weak-scaling leg")
    ap.add_argument("--global-n", type=int, default=None,
                    help="strong scaling: an n^3 grid slab-decomposed over the ranks (default 513 when N > 1; "
                         "0 = weak scaling with --n planes per rank)")
    ap.add_argument("--workload", default=None, choices=["C4", "C5"],
                    help="slab leg only: C4 = Dubins 3-D fp64 (default for N > 1), C5 = double pendulum 4-D fp32, all axes "
                         "periodic, ring of slabs (BASELINE configs[4]; --global-n = points per axis, default 129)")
    ap.add_argument("--scheme", default="WENO5_ASSHIPPED", choices=["WENO5", "WENO5_ASSHIPPED", "ENO3", "ENO2"],
                    help="WENO5_ASSHIPPED = what the reference's upwindFirstWENO5 computes (parity-pinned; "
                         "headline); WENO5 = the intended nonlinear scheme (reported in 'also')")
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra workloads reported under 'also'")
    ap.add_argument("--also", default="WENO5,513,C3,C3 fast,C5,CFL,API,RTC,RANGE", help="comma list of the extra workloads to time (API = the 201^3 / "
                    "51^3 workloads through odeCFL3 / HJIPDE_solve, the reference's own call protocol; RTC = the same system as a "
                    "Hamiltonian compiled at run time with hipRTC, and as Python callbacks on the split path; RANGE = a Hamiltonian whose alpha depends "
                    "on the costate range, fused in two launches per stage against the split path)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run")
    ap.add_argument("--plan-only", action="store_true",
                    help="with --gpus N: print (one JSON line; a table on stderr) what every rank of the N-rank slab leg will do -- slab, "
                         "stepper and schedule, the launches of a substep, halo bytes per step, predicted ms/step -- WITHOUT touching a GPU")
    ap.add_argument("--single", default=None, help=argparse.SUPPRESS)     # child passes of live_traffic: the main leg times this `also` workload (C3, C5)
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ CPU baseline
# Workers are started BEFORE the GPU is initialised (a process that holds the GPU must not fork/exec
# on this pool) and sit blocked on stdin until the GPU legs are over, so they do not disturb the timing.
def cpu_worker(spec):
    """Child process: the NumPy oracle (a port of the reference's array path; 18x faster than the reference
    import itself, which materialises 519 fancy-index gathers per step) on n^3 for `steps` RK3 steps.  With a
    fourth field (a path prefix) the initial data and the state after the steps are saved as .npy there: the GPU leg
    integrates the SAME data through the drop-in API and the bench line carries the difference ("parity")."""
    parts = spec.split(":")
    n, steps, scheme = int(parts[0]), int(parts[1]), parts[2]
    save = parts[3] if len(parts) > 3 and parts[3] else None
    from oracle import hj_oracle as O
    og = O.Grid([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n)], [n] * 3, [2])
    osys = O.DubinsRel(og, 1, 1)
    y = O.shape_cylinder(og, 2, None, .5).reshape(-1, 1)
    if save:
        np.save(save + "_y0.npy", y)
    term = lambda tt, yy: O.term_lax_friedrichs(og, osys, scheme, tt, yy)  # noqa: E731
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    if not sys.stdin.readline().startswith("go"):
        return
    t = 0.
    t0 = time.perf_counter()
    for _ in range(steps):
        t, y = O.ode_cfl_3(term, [t, 10.], y, 0.8, single_step=True)
    sec = time.perf_counter() - t0
    if save:
        np.save(save + "_y.npy", y)
    sys.stdout.write(json.dumps({"n": n, "steps": steps, "sec": sec, "finite": bool(np.isfinite(y).all()), "t": float(t),
                                 "state": save}) + "\n")
    sys.stdout.flush()


PARITY_TOL = 1e-11      # HIP fp64 vs the CPU restatement, absolute, O(1) data


def gpu_parity(L, torch, c2, scheme):
    """The `parity` object of the bench line: the oracle's state after its timed steps at the HEADLINE size (saved by the
    cpu_baseline worker) against the product path run from the same initial data through the drop-in API --
    odeCFL3(termLaxFriedrichs, ...) with singleStep='on', device tensor in/out (the reference call being replaced:
    ValueFuncs/hji_solver.py:542).  Raises if the tolerance is exceeded: a fast wrong kernel must not print a value."""
    pre, n, steps = c2["state"], c2["n"], c2["steps"]
    y0, yref = np.load(pre + "_y0.npy"), np.load(pre + "_y.npy")
    for suf in ("_y0.npy", "_y.npy"):
        try:
            os.remove(pre + suf)
        except OSError:
            pass
    g = dubins_grid(L, n, n)
    sysd = L.DubinsVehicleRel(g, 1, 1)
    calc = {"WENO5_ASSHIPPED": L.upwindFirstWENO5, "ENO3": L.upwindFirstENO3, "ENO2": L.upwindFirstENO2}.get(scheme)
    if calc is None:
        calc = L.upwindFirstWENO5Intended
    sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation,
                       dissFunc=L.artificialDissipationGLF, CoStateCalc=calc))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y = torch.from_numpy(y0).to("cuda")
    t = 0.
    for _ in range(steps):
        t, y, _sd = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    kern = L.context.device_grid(g).lib.hj_last_kernel(L.context.device_grid(g).ctx)
    diff = float(np.max(np.abs(y.cpu().numpy() - yref)))
    dt = abs(float(t) - c2["t"])
    ok = diff <= PARITY_TOL and dt <= 1e-14 and c2["finite"]
    out = {"config": "%d^3" % n, "steps": steps, "scheme": scheme, "max_abs_diff": diff, "tol": PARITY_TOL, "t_abs_diff": dt,
           "ok": bool(ok), "kernel": kern.decode() if kern else "?",
           "path": "odeCFL3(termLaxFriedrichs, singleStep='on') on a device tensor vs oracle/hj_oracle.py on the same initial data",
           "max_abs_state": float(np.max(np.abs(yref)))}
    if not ok:
        raise RuntimeError("parity at the headline size FAILED: %r" % (out,))
    return out


class CpuBaseline(object):
    def __init__(self, scheme, n=201):
        self.scheme, self.n = scheme, n
        try:
            self.cores_avail = len(os.sched_getaffinity(0))
        except AttributeError:
            self.cores_avail = os.cpu_count() or 1
        self.pool_size = max(1, min(self.cores_avail, 16))
        import tempfile
        self.state_prefix = os.path.join(tempfile.gettempdir(), "hj_bench_parity_%d" % os.getpid())
        self.specs = [("c2", "%d:2:%s:%s" % (n, scheme, self.state_prefix)), ("c1", "51:10:%s" % scheme)] + \
                     [("all%d" % i, "101:4:%s" % scheme) for i in range(self.pool_size)]
        self.procs = []
        for name, spec in self.specs:
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", spec],
                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT)
            self.procs.append((name, p))

    def _go(self, items):
        for _, p in items:
            assert p.stdout.readline().strip() == "ready"
        t0 = time.perf_counter()
        for _, p in items:
            p.stdin.write("go\n")
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for _, p in items]
        wall = time.perf_counter() - t0
        for _, p in items:
            p.wait()
        return res, wall

    def run(self):
        (c2,), _ = self._go(self.procs[0:1])
        (c1,), _ = self._go(self.procs[1:2])
        allr, wall = self._go(self.procs[2:])
        rate = lambda r: r["n"] ** 3 * 3 * r["steps"] / r["sec"]  # noqa: E731
        units = sum(r["n"] ** 3 * 3 * r["steps"] for r in allr)
        try:
            model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
        except Exception:  # noqa: BLE001
            model = "?"
        self.c2 = c2
        return {"value": rate(c2), "unit": "cell-substeps/s", "cores": 1, "kind": "port",
                "sample": "%d RK3 steps of Dubins-relative %d^3 %s+GLF fp64 (the GPU workload itself) with "
                          "oracle/hj_oracle.py (NumPy restatement pinned to the reference's outputs, single-threaded) "
                          "in %.1f s on one core of %s (%d cores visible, %d in this process's affinity); %s"
                          % (c2["steps"], self.n, self.scheme, c2["sec"], model, os.cpu_count(), self.cores_avail, REF_IMPORT),
                "c1_51cubed": {"value": rate(c1), "cores": 1, "sample": "%d RK3 steps at 51^3 in %.2f s" % (c1["steps"], c1["sec"])},
                "all_cores": {"value": units / wall, "cores": len(allr),
                              "sample": "%d processes, each %d RK3 steps at 101^3 on its own grid, wall %.1f s"
                                        % (len(allr), allr[0]["steps"], wall)}}

    def abort(self):
        for _, p in self.procs:
            if p.poll() is None:
                p.kill()


# ------------------------------------------------------------------------------------------ workloads
def dubins_grid(L, n0, n):
    """C2/C4 geometry; n0 planes along axis 0 (dx kept, max extended for weak-scaling slabs)."""
    dx0 = 4.0 / (n - 1)
    gmin = np.array([[-.75, -1.25, -np.pi]]).T
    gmax = np.array([[-.75 + dx0 * (n0 - 1), 1.25, np.pi * (1 - 2 / n)]]).T
    N = np.array([[n0], [n], [n]], dtype=np.int64)
    return L.createGrid(gmin, gmax, N, 2, low_mem=True)


def device_sdf(torch, g, radius, ignore=(), dtype=None):
    """shapeCylinder / shapeSphere of a low-memory grid built on the device: sqrt(sum_i x_i^2) - r over the
    axes not in `ignore` (cylinder.py:55-59, sphere.py:50-57)."""
    shape = [int(v) for v in np.asarray(g.N).ravel()]
    acc = torch.zeros(shape, dtype=torch.float64, device="cuda")
    for i, v in enumerate(g.vs):
        if i in ignore:
            continue
        view = [1] * len(shape)
        view[i] = -1
        acc += torch.as_tensor(np.asarray(v).ravel(), device="cuda").reshape(view) ** 2
    return (acc.sqrt_() - radius).to(dtype or torch.float64).contiguous()


def workload(L, _ffi, torch, name, scheme, dtype, n):
    """(description, grid, ham id, params, scheme, dtype, initial data on the device)."""
    if name == "dubins":
        g = dubins_grid(L, n, n)
        d0 = device_sdf(torch, g, 0.5, ignore=(2,), dtype=torch.float64 if dtype == "float64" else torch.float32)
        return ("Dubins-relative (air3D) 3-D HJI, %d x %d x %d grid, %s + GLF, odeCFL3 (factorCFL 0.8), cylinder r=0.5 "
                "initial data" % (n, n, n, scheme)), g, _ffi.HAM_DUBINS_REL, [1.0, 1.0, 1.0, 2.0], scheme, dtype, d0
    if name == "C3":
        n = 4096
        g = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), n * np.ones((2, 1), dtype=np.int64), None, low_mem=True)
        return ("double integrator 2-D, 4096 x 4096 grid, ENO3 + GLF, odeCFL3 (factorCFL 0.8), sphere r=0.25 (BASELINE C3)",
                g, _ffi.HAM_DOUBLE_INTEGRATOR, [1.0, 0, 0, 0], "ENO3", "float64", device_sdf(torch, g, 0.25))
    if name == "C3 fast":
        w = list(workload(L, _ffi, torch, "C3", scheme, dtype, n))
        w[0] = w[0].replace("ENO3 + GLF", "ENO3 in the opt-in lean arithmetic (set_eno_mode('fast'): 1e-11 from the reference outside cells whose stencil "
                            "selectors tie within rounding; the default is bit for bit) + GLF")
        w[4] = "ENO3_FAST"
        return tuple(w)
    if name == "C5":
        n = int(os.environ.get("HJ_BENCH_C5_N", "129"))
        gmin = np.array([[-np.pi, -8, -np.pi, -8]]).T
        gmax = np.array([[np.pi * (1 - 2 / n), 8 * (1 - 2 / n), np.pi * (1 - 2 / n), 8 * (1 - 2 / n)]]).T
        g = L.createGrid(gmin, gmax, n * np.ones((4, 1), dtype=np.int64), [0, 1, 2, 3], low_mem=True)
        return ("double pendulum 4-D, %d^4 grid fp32, all axes periodic, WENO5_ASSHIPPED + GLF, odeCFL3, sphere r=0.5 "
                "(BASELINE C5 on one GPU)" % n, g, _ffi.HAM_DOUBLE_PENDULUM, [1.0, 0, 0, 0], "WENO5_ASSHIPPED", "float32",
                device_sdf(torch, g, 0.5, dtype=torch.float32))
    raise ValueError(name)


class StepRunner(object):
    """One odeCFL3 step per call (hj_rk_step: 3 launches, three arrays) on a context of its own.  `env`: environment the context is
    created under (the library reads its knobs in hj_ctx_create); `share`: another runner whose state arrays this one steps too."""

    def __init__(self, torch, _ffi, DeviceGrid, wl, env=None, share=None):
        self.desc, g, self.ham, par, self.scheme, self.dtype, d0 = wl
        old = {}
        for k, v in (env or {}).items():
            old[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            self.dg = DeviceGrid(g, self.dtype)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.dg.bind_stream()
        self.lib = self.dg.lib
        self.sid = _ffi.SCHEME_IDS[self.scheme]
        self.state = share.state if share is not None else {"cur": d0.clone(), "nxt": self.dg.empty(), "w1": self.dg.empty(), "t": 0.0}
        self.tout, self.dtout = C.c_double(), C.c_double()
        self.parv = (C.c_double * 4)(*par)

    def one(self):
        # three arrays are enough for RK3: the first stage buffer doubles as the output (stage 3 reads w1
        # and y0 = cur only), which keeps the 201^3 working set (195 MB) inside the 256 MB Infinity Cache
        st, dg = self.state, self.dg
        rc = self.lib.hj_rk_step(dg.ctx, 3, self.sid, self.ham, self.parv, st["t"], 1e9, 0.8, 1e300, 0, dg.ptr(st["cur"]),
                                 dg.ptr(st["nxt"]), dg.ptr(st["nxt"]), dg.ptr(st["w1"]), C.byref(self.tout), C.byref(self.dtout))
        if rc != 0:
            raise RuntimeError(self.lib.hj_last_error().decode())
        st["cur"], st["nxt"], st["t"] = st["nxt"], st["cur"], float(self.tout.value)

    def window(self, torch, steps):
        """(wall seconds, HIP-event milliseconds) of exactly `steps` steps between synchronisations"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            self.one()
        e1.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, e0.elapsed_time(e1)

    def settle(self, torch, steps, spinup):
        # device spin-up (untimed, reported as config.spinup_steps): the GPU's clocks need a few tens of
        # milliseconds of load to reach their steady state; without it a short --steps run measures the ramp
        for _ in range(spinup):
            self.one()
        # ... and until the clocks have SETTLED: untimed blocks of K steps until three in a row agree within 0.5 % (at most
        # HJ_BENCH_SETTLE_BLOCKS = 60 blocks).  A fixed 300-step spin-up (33 ms at 201^3) was not always enough after the child passes
        # that precede this leg: a run whose 25 windows drifted by 2 % reported the headline 5 % below what the later legs of the
        # SAME run measured with the same kernels (profiles/r04_bench_default.json vs gpurun r04_run12)
        settle, last = 0, []
        for _ in range(int(os.environ.get("HJ_BENCH_SETTLE_BLOCKS", "60"))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.one()
            torch.cuda.synchronize()
            last.append(time.perf_counter() - t0)
            settle += steps
            if len(last) >= 3 and max(last[-3:]) - min(last[-3:]) <= 0.005 * min(last[-3:]):
                break
        return settle

    def result(self, torch, walls, devs, settle):
        assert bool(torch.isfinite(self.state["cur"]).all()), "non-finite state after the timed steps (%s)" % self.desc
        nl, fused = C.c_int(3), C.c_int(0)
        if self.lib.hj_rk_plan(self.dg.ctx, 3, self.sid, self.ham, self.parv, 0, C.byref(nl), C.byref(fused)) != 0:
            raise RuntimeError(self.lib.hj_last_error().decode())
        kern = self.lib.hj_last_kernel(self.dg.ctx)
        return {"desc": self.desc, "cells": self.dg.numel, "dtype": self.dtype, "scheme": self.scheme, "walls": walls, "devs": devs,
                "settle_steps": settle, "launches_per_step": int(nl.value), "stage_fused": int(fused.value),
                "kernel": kern.decode() if kern else "?"}


def time_single(torch, _ffi, DeviceGrid, wl, steps, warmup, repeats, spinup):
    """R repeats of K odeCFL3 steps (hj_rk_step: 3 launches, three arrays) on one GPU.  Returns a dict with
    the wall seconds and HIP-event milliseconds of every repeat."""
    rn = StepRunner(torch, _ffi, DeviceGrid, wl)
    settle = rn.settle(torch, steps, spinup)
    for _ in range(warmup):
        rn.one()
    walls, devs = [], []
    for _ in range(repeats):
        w, d = rn.window(torch, steps)
        walls.append(w)
        devs.append(d)
    return rn.result(torch, walls, devs, settle)


def time_interleaved(torch, _ffi, DeviceGrid, wl, env_b, steps, warmup, repeats, spinup):
    """A/B in ONE run: the default context (A) and a context created under `env_b` (B) step the SAME three arrays, a window of K steps
    each in turn -- same clocks, same cache state, same neighbours on the box.  Returns (result A, result B).  (the
    kept-CFL leg, timed minutes after the headline, read -8.3 % in the driver's run and -2.1 % in a same-box A/B.)"""
    ra = StepRunner(torch, _ffi, DeviceGrid, wl)
    rb = StepRunner(torch, _ffi, DeviceGrid, wl, env=env_b, share=ra)
    settle = ra.settle(torch, steps, spinup)
    for _ in range(warmup):
        ra.one()
        rb.one()
    wa, da, wb, db = [], [], [], []
    for _ in range(repeats):
        w, d = ra.window(torch, steps)
        wa.append(w)
        da.append(d)
        w, d = rb.window(torch, steps)
        wb.append(w)
        db.append(d)
    return ra.result(torch, wa, da, settle), rb.result(torch, wb, db, settle)


def quartiles(xs):
    """(q1, median, q3) by linear interpolation."""
    v = sorted(xs)
    def at(f):
        x = f * (len(v) - 1)
        i = int(x)
        return v[i] if i + 1 >= len(v) else v[i] + (x - i) * (v[i + 1] - v[i])
    return at(0.25), at(0.5), at(0.75)


def time_api(L, torch, n, kind, scheme, steps, warmup, repeats, spinup):
    """The same workload through the reference's own call protocol (what a user of LevelSetPy switches to; the raw C loop
    of time_single is the floor these are compared with):
      "odeCFL3"  K calls of odeCFL3(termLaxFriedrichs, [t, tf], y, options(singleStep='on'), schemeData), device tensor
                 in / tensor out -- the call HJIPDE_solve makes per step (reference ValueFuncs/hji_solver.py:542)
      "solve"    ONE HJIPDE_solve(data0, [0, T], schemeData, 'none', keepLast) call per window with T = (K - 1/2) dt,
                 i.e. exactly K steps (the last one clipped), tensor in / tensor out
    Every window is synchronised on both sides; one window = K RK3 steps."""
    g = dubins_grid(L, n, n)
    sysd = L.DubinsVehicleRel(g, 1, 1)
    calc = {"WENO5_ASSHIPPED": L.upwindFirstWENO5, "ENO3": L.upwindFirstENO3, "ENO2": L.upwindFirstENO2,
            "WENO5": L.upwindFirstWENO5Intended}[scheme]
    sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation,
                       dissFunc=L.artificialDissipationGLF, CoStateCalc=calc))
    d0 = device_sdf(torch, g, 0.5, ignore=(2,))
    cells = d0.numel()
    st = {"y": d0.reshape(-1, 1), "t": 0.0}
    if kind in ("odeCFL3", "numpy"):
        op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
        if kind == "numpy":
            # the reference's own driver loop: a NumPy array in, whatever comes back fed into the next call
            st["y"] = d0.reshape(-1, 1).cpu().numpy()

        def window(k):
            for _ in range(k):
                st["t"], st["y"], _sd = L.odeCFL3(L.termLaxFriedrichs, [st["t"], 1e9], st["y"], op, sd)
    else:
        dxs = [float(v) for v in np.asarray(g.dx).ravel()]
        # static stepBound of the Dubins system (dubins_relative.py:106-111, artificial_diss_glf.py:107-109), v = w = 1
        x1 = max(abs(float(np.asarray(g.vs[0]).ravel()[0])), abs(float(np.asarray(g.vs[0]).ravel()[-1])))
        x2 = max(abs(float(np.asarray(g.vs[1]).ravel()[0])), abs(float(np.asarray(g.vs[1]).ravel()[-1])))
        c3 = np.cos(np.asarray(g.vs[2]).ravel())
        s3 = np.sin(np.asarray(g.vs[2]).ravel())
        a0 = float(np.max(np.abs(1.0 - c3))) + x2
        a1 = float(np.max(np.abs(s3))) + x1
        dt = 0.8 / (a0 / dxs[0] + a1 / dxs[1] + 2.0 / dxs[2])
        ex = L.Bundle(dict(keepLast=True, quiet=True))
        st["y"] = d0

        def window(k):
            data, _tau, _e = L.HJIPDE_solve(st["y"], [0.0, (k - 0.5) * dt], sd, 'none', ex)
            st["y"] = data
    window(max(1, min(spinup, 60)))
    for _ in range(max(1, warmup // max(1, steps)) + 1):
        window(steps)
    walls = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        window(steps)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    yfin = st["y"]
    if kind == "numpy":
        kind_out = type(yfin).__name__
        assert bool(np.isfinite(np.asarray(yfin)).all())
    else:
        assert bool(torch.isfinite(yfin).all())
    med = statistics.median(walls)
    q = quartiles(walls)
    bps = BYTES_PER_SUBSTEP["float64"]
    val = cells * 3 * steps / med
    return {"value": val, "ms_per_step": 1e3 * med / steps, "steps": steps, "dtype": "f64",
            "roofline_frac": val * bps / 1e9 / HBM_PEAK_GBS,
            "repeats": {"n": len(walls), "iqr_over_median": (q[2] - q[0]) / med,
                        "value_min": cells * 3 * steps / max(walls), "value_max": cells * 3 * steps / min(walls)},
            "workload": "Dubins-relative %d^3 %s + GLF through %s" % (n, scheme, {
                "odeCFL3": "odeCFL3(termLaxFriedrichs, ..., singleStep='on') calls, device tensor in / out",
                "numpy": "odeCFL3(termLaxFriedrichs, ..., singleStep='on') calls, a NumPy array in, every result fed back into the next call "
                         "(the reference's driver loop; results are lazy.HostView handles that stay in HBM until somebody looks)",
                "solve": "one HJIPDE_solve(keepLast) call per window of K steps, device tensor in / out"}[kind])}


DUBINS_REL_COL = "col[0] = cos(x[2]); col[1] = sin(x[2]);"      # once per grid column, outside the march
DUBINS_REL_SRC = """
    const T c3 = col[0], s3 = col[1];
    H = p[0] * (par[0] - par[1] * c3) - p[1] * (par[1] * s3) - par[2] * fabs(p[0] * x[1] - p[1] * x[0] - p[2]) + par[2] * fabs(p[2]);
    alpha[0] = fabs(par[0] - par[1] * c3) + fabs(par[2] * x[1]);
    alpha[1] = fabs(par[1] * s3) + fabs(par[2] * x[0]);
    alpha[2] = par[3];
"""


def time_runtime_ham(L, torch, _ffi, DeviceGrid, a, s_head):
    """The headline system written by a USER: (1) as a device expression registered at run time (hj_ham_register: the fused
    pair kernel compiled with hipRTC, one pair per thread in 256-thread workgroups) through the same raw C loop as the
    headline; (2) as plain Python callbacks on device tensors (the split path every foreign hamFunc / partialFunc takes:
    derivative kernels -> callbacks -> dissipation kernel -> hj_rk_combine) through odeCFL3."""
    out = {}
    reg = L.register_native_hamiltonian("bench_dubins_rel", 3, DUBINS_REL_SRC, nparams=4, column_src=DUBINS_REL_COL, ncol=2)
    wl = list(workload(L, _ffi, torch, "dubins", a.scheme, a.dtype, a.n))
    wl[0] = wl[0].replace("Dubins-relative", "Dubins-relative as a run-time (hipRTC) Hamiltonian")
    wl[2] = reg.ham_id
    t0 = time.perf_counter()
    r = time_single(torch, _ffi, DeviceGrid, tuple(wl), a.steps, a.warmup, min(15, a.repeats), SPINUP_STEPS)
    s = summarize(r, a.steps)
    out["%d^3 run-time Hamiltonian (hipRTC)" % a.n] = {
        "workload": r["desc"], "value": s["value"], "ms_per_step": s["ms_per_step"], "roofline_frac": s["frac"], "repeats": s["repeats"],
        "kernel": r["kernel"], "vs_builtin": s["value"] / s_head["value"], "leg_wall_s_incl_compile": time.perf_counter() - t0}
    # the same formulas as FOREIGN Python callbacks on device tensors (lambdas: nothing the library could recognise) through odeCFL3
    # single-step calls: (a) HJ_TRACE=0 -- the split path every foreign hamFunc / partialFunc took until round 6; (b) default -- the
    # callbacks are traced into a device expression (levelsetpy_amd/trace_ham.py), compiled with hipRTC, checked against the callbacks on
    # the first data, and run fused
    g = dubins_grid(L, a.n, a.n)
    sysd = L.DubinsVehicleRel(g, 1, 1)
    calc = {"WENO5_ASSHIPPED": L.upwindFirstWENO5, "ENO3": L.upwindFirstENO3, "ENO2": L.upwindFirstENO2, "WENO5": L.upwindFirstWENO5Intended}[a.scheme]
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    legs, finals = {}, {}
    for kind in ("split", "traced"):
        prev = os.environ.get("HJ_TRACE")
        if kind == "split":
            os.environ["HJ_TRACE"] = "0"
        try:
            sd = L.Bundle(dict(grid=g, hamFunc=lambda t, d, p, sd_: sysd.hamiltonian(t, d, p, sd_),
                               partialFunc=lambda t, d, lo, hi, sd_, dim: sysd.dissipation(t, d, lo, hi, sd_, dim),
                               dissFunc=L.artificialDissipationGLF, CoStateCalc=calc))
            y, t = device_sdf(torch, g, 0.5, ignore=(2,)).reshape(-1, 1), 0.0
            t0 = time.perf_counter()
            for _ in range(3):
                t, y, _sd = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            finals[kind] = (t, y.clone())
            k = max(3, min(10, a.steps)) if kind == "split" else a.steps
            walls = []
            for _ in range(3 if kind == "split" else min(9, a.repeats)):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k):
                    t, y, _sd = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t1)
            med = statistics.median(walls)
            cells = y.numel()
            from levelsetpy_amd.context import device_grid as _dgrid
            dgk = _dgrid(g)
            legs[kind] = {"value": cells * 3 * k / med, "ms_per_step": 1e3 * med / k, "steps": k, "first_three_steps_s_incl_trace_compile_check": first,
                          "kernel": dgk.lib.hj_last_kernel(dgk.ctx).decode()}
        finally:
            if prev is None:
                os.environ.pop("HJ_TRACE", None)
            else:
                os.environ["HJ_TRACE"] = prev
    dmax = float((finals["traced"][1] - finals["split"][1]).abs().max())
    assert abs(finals["traced"][0] - finals["split"][0]) <= 1e-12 and dmax <= 1e-10, (finals["traced"][0], finals["split"][0], dmax)
    out["%d^3 foreign Python callbacks (split path)" % a.n] = dict(
        legs["split"], workload="the same system as Python hamFunc / partialFunc callbacks (lambdas) on device tensors through odeCFL3 single steps, HJ_TRACE=0: "
        "derivative kernels -> callbacks -> dissipation kernel", vs_run_time_hamiltonian=legs["split"]["value"] / s["value"])
    out["%d^3 foreign Python callbacks (traced)" % a.n] = dict(
        legs["traced"], workload="the same lambdas, default: traced into a device expression, compiled with hipRTC, checked against the callbacks, run fused "
        "(odeCFL3 single-step calls: one Python call per step)", vs_split_path=legs["traced"]["value"] / legs["split"]["value"],
        vs_run_time_hamiltonian=legs["traced"]["value"] / s["value"], max_abs_diff_vs_split_after_3_steps=dmax)
    return out


RANGE_SRC = """
    H = par[0] * x[0] * p[1] + 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    alpha[0] = fmax(fabs(dmin[0]), fabs(dmax[0]));
    alpha[1] = fmax(fabs(dmin[1]), fabs(dmax[1])) + fabs(par[0] * x[0]);
    alpha[2] = fmax(fabs(dmin[2]), fabs(dmax[2]));
"""


class _RangeSystem(object):
    """H = |p|^2 / 2 + c x_0 p_1 with the partials bounded over the costate RANGE the reference hands to partialFunc
    (artificial_diss_glf.py:80-99) -- a system whose alpha depends on derivMin / derivMax: the general protocol."""

    def __init__(self, grid, c, torch):
        self.grid, self.c, self.torch = grid, c, torch
        self._x0 = None

    def _x0t(self, like):
        if self._x0 is None:
            v = np.asarray(self.grid.vs[0]).ravel()
            self._x0 = self.torch.as_tensor(v, device=like.device, dtype=like.dtype).reshape(-1, 1, 1)
        return self._x0

    def hamiltonian(self, t, data, p, sd=None):
        return self.c * self._x0t(p[0]) * p[1] + 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2])

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        a = max(abs(float(dmin[dim])), abs(float(dmax[dim])))
        if dim != 1:
            return a
        return (a + (self.c * self._x0t(data)).abs()).expand(data.shape).contiguous()


def time_range_ham(L, torch, a):
    """A Hamiltonian whose alpha depends on the costate range, through odeCFL3 single-step calls (device tensors): as Python
    callbacks on the split path (before registration), then registered as a device expression: range pass + fused substep per
    stage, deltaT from the first stage's reduced bound (one host read per step)."""
    g = dubins_grid(L, a.n, a.n)
    calc = {"WENO5_ASSHIPPED": L.upwindFirstWENO5, "ENO3": L.upwindFirstENO3, "ENO2": L.upwindFirstENO2, "WENO5": L.upwindFirstWENO5Intended}[a.scheme]
    sysd = _RangeSystem(g, 0.7, torch)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    res = {}
    ys = {}
    # split: HJ_TRACE=0, the callbacks as they are; traced (round 6): the same callbacks, traced by the library (their float() / max() of the
    # range are rewritten for the trace: trace_ham._rewritten); fused: the hand-written expression RANGE_SRC attached to the object
    for kind in ("split", "traced", "fused"):
        if kind == "fused":
            L.register_native_hamiltonian("bench_range", 3, RANGE_SRC, nparams=1).attach(sysd, params=lambda o: [o.c])
        prev_trace = os.environ.get("HJ_TRACE")
        if kind == "split":
            os.environ["HJ_TRACE"] = "0"
        sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=calc))
        y, t = device_sdf(torch, g, 0.5, ignore=(2,)).reshape(-1, 1), 0.0
        try:
            for _ in range(3):
                t, y, _sd = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
        finally:
            if prev_trace is None:
                os.environ.pop("HJ_TRACE", None)
            else:
                os.environ["HJ_TRACE"] = prev_trace
        ys[kind] = (t, y.clone())
        k = max(3, min(10, a.steps)) if kind == "split" else a.steps
        walls = []
        for _ in range(3 if kind == "split" else min(9, a.repeats)):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k):
                t, y, _sd = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t1)
        assert bool(torch.isfinite(y).all())
        med = statistics.median(walls)
        from levelsetpy_amd.context import device_grid as _dgrid
        dgk = _dgrid(g)
        res[kind] = {"value": y.numel() * 3 * k / med, "ms_per_step": 1e3 * med / k, "steps": k, "kernel": dgk.lib.hj_last_kernel(dgk.ctx).decode()}
    local = {}
    for name, fn in (("LLF", L.artificialDissipationLLF), ("LLLF", L.artificialDissipationLLLF)):
        try:
            sd = L.Bundle(dict(grid=g, hamFunc=sysd.hamiltonian, partialFunc=sysd.dissipation, dissFunc=fn, CoStateCalc=calc))
            y, t = device_sdf(torch, g, 0.5, ignore=(2,)).reshape(-1, 1), 0.0
            for _ in range(3):
                t, y, _sd = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                t, y, _sd = L.odeCFL3(L.termLaxFriedrichs, [t, 1e9], y, op, sd)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(y).all())
            local[name] = 1e3 * (time.perf_counter() - t1) / a.steps
        except Exception as e:  # noqa: BLE001
            local[name] = repr(e)
    # the two paths computed the same three steps (1e-11: different operation order in the range reduction and the callbacks)
    dmax = float((ys["fused"][1] - ys["split"][1]).abs().max())
    assert abs(ys["fused"][0] - ys["split"][0]) <= 1e-12 and dmax <= 1e-10, (ys["fused"][0], ys["split"][0], dmax)
    bps = BYTES_PER_SUBSTEP["float64"]
    return {"%d^3 data-dependent alpha (range pass + fused substep)" % a.n: {
        "workload": "H = |p|^2/2 + c x0 p1, alpha_d = max(|derivMin_d|, |derivMax_d|) (+ |c x0|), %s + GLF: a Hamiltonian registered at run time "
                    "whose alpha reads the costate range (artificial_diss_glf.py:80-99), odeCFL3 singleStep calls on device tensors; "
                    "two launches per stage + the bound kernel; deltaT is formed on the device and read by the first stage from device memory, "
                    "the host polls the same value from page-locked memory (no stream synchronisation inside a step)" % a.scheme,
        "value": res["fused"]["value"], "ms_per_step": res["fused"]["ms_per_step"], "steps": res["fused"]["steps"],
        "roofline_frac": res["fused"]["value"] * bps / 1e9 / HBM_PEAK_GBS,
        "split_path_ms_per_step": res["split"]["ms_per_step"], "vs_split_path": res["fused"]["value"] / res["split"]["value"],
        "fused_vs_split_max_abs_diff_after_3_steps": dmax,
        # the SAME Python callbacks traced by the library instead of the hand-written expression (round 6)
        "traced_callbacks": dict(res["traced"], vs_split_path=res["traced"]["value"] / res["split"]["value"],
                                 max_abs_diff_vs_split_after_3_steps=float((ys["traced"][1] - ys["split"][1]).abs().max())),
        # the same system under the LOCAL Lax-Friedrichs variants (per-node costate ranges inside the fused kernel, round 5): fused only
        "local_variants_fused_ms_per_step": local}}


def summarize(r, steps):
    """value from the MEDIAN repeat, spread, and the roofline numbers of the step's launches."""
    med = statistics.median(r["walls"])
    k = r["walls"].index(sorted(r["walls"])[len(r["walls"]) // 2])
    per_s = lambda w: r["cells"] * 3 * steps / w  # noqa: E731
    bps = BYTES_PER_SUBSTEP[r["dtype"]]
    dev_step_ms = r["devs"][k] / steps
    achieved = r["cells"] * 3 * bps / (dev_step_ms * 1e-3) / 1e9
    return {"value": per_s(med), "ms_per_step": 1e3 * med / steps,
            "repeats": {"n": len(r["walls"]), "value_min": per_s(max(r["walls"])), "value_max": per_s(min(r["walls"])),
                        "value_q1": per_s(quartiles(r["walls"])[2]), "value_q3": per_s(quartiles(r["walls"])[0]),
                        "spread": (max(r["walls"]) - min(r["walls"])) / med,
                        "iqr_over_median": (quartiles(r["walls"])[2] - quartiles(r["walls"])[0]) / med,
                        "windows_ms": [round(1e3 * w, 4) for w in r["walls"]]},
            "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "dev_step_ms": dev_step_ms}


def roofline_obj(r, s, tr, step_bytes, ach=None):
    """The `roofline` object of one workload.  achieved = algorithmic bytes of an RK3 step / the time of the step's
    launches, with that time taken as the LARGER of (a) the HIP-event time of the step on the launch stream, median
    repeat, and (b) the sum of the kernels' durations in the rocprofv3 kernel-trace child pass of this run (VERDICT r02:
    the two differed by 3-4 %; the fraction must follow from the profile evidence too).  `bound` is "hbm": the fraction is of
    the 8 TB/s HBM peak (the metric's roofline) for every workload; `memory_regime` says where the bytes are served from -- the
    working set of a step (three arrays) against the 256 MiB Infinity Cache: below it the launches stream from the cache / fabric;
    the HBM-resident companion of the headline is also["513^3 ..."]."""
    nl = r["launches_per_step"]
    ev_ms = s["dev_step_ms"]
    rp = (tr or {}).get("rocprof") if tr else None
    rp_ms = rp["step_kernels_ms"] if rp else None
    step_ms = max(ev_ms, rp_ms) if rp_ms else ev_ms
    achieved = step_bytes / (step_ms * 1e-3) / 1e9
    working_set = 3 * r["cells"] * (8 if r["dtype"] == "float64" else 4)
    regime = "hbm" if working_set > 256 * 2 ** 20 else "infinity_cache"
    # what a plain streaming kernel reaches on this box in the step's access mix (one 1R:1W launch + two 2R:1W launches per
    # RK3 step: bytes-weighted harmonic mix of the copy and triad rates), in the memory regime of this working set
    ach_gbs = None
    if ach and ach.get(regime):
        c, t3 = ach[regime]["copy"] * 1e3, ach[regime]["triad"] * 1e3
        ach_gbs = 8.0 / (2.0 / c + 6.0 / t3) if (nl == 3 and c > 0 and t3 > 0) else t3
    # (`bound` names the roofline the fraction is taken of -- the 8 TB/s HBM peak on algorithmic bytes, for every workload --; where
    #  the bytes are actually served from is `memory_regime`)
    return {"bound": "hbm", "memory_regime": regime,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "achievable": ach_gbs, "frac_of_achievable": (achieved / ach_gbs) if ach_gbs else None,
            "achievable_source": (ach["source"] + "; regime: " + regime) if ach_gbs else None,
            # measured fabric bytes per launch / per step (null unless counters were collected on these kernel sources)
            "traffic": tr["bytes_per_launch"] if tr else None,
            "traffic_per_step": tr["bytes_per_step"] if tr else None,
            "traffic_source": tr["source"] if tr else None,
            "kernel": "%s x %d launches = one RK3 step%s" % (r["kernel"], nl, " (stages 1+2 in one launch)" if r.get("stage_fused") else ""),
            "kernel_ms": step_ms / nl, "step_ms": step_ms,
            "kernel_ms_hip_events": ev_ms / nl, "step_ms_hip_events": ev_ms,
            "kernel_ms_rocprof": rp["kernel_ms"] if rp else None, "step_ms_rocprof": rp_ms,
            "working_set_bytes": working_set,
            "algorithmic_bytes_per_launch": step_bytes / nl, "algorithmic_bytes_per_step": step_bytes}


# fp64 vector peak of MI355X: 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 FMA lanes per clock x 2 flop x 2.4 GHz
# (MI355X_MICROARCH.md: vector FP32 157.3 TF, fp64 at half rate) = 39.3e12 fp64 lane-operations per second
VALU_F64_LANE_OPS = 39.3e12
# VALU instructions per cell and plane (= per cell-substep) of the intended WENO5 kernel's plane loop (Dubins, 3-D): 293 in round 5
# (268 of them fp64; static count on the built library, tools/isa_mix.py -> profiles/r05_isa_mix.txt; 335 by SQ_INSTS_VALU in
# round 3, before the smoothness terms were shared).  All counted as fp64-rate ones: an upper bound on the fraction
WENO5_VALU_OPS_PER_CELL = float(os.environ.get("HJ_WENO5_VALU_OPS", "293"))


def valu_ceiling(cell_substeps_per_s):
    achieved = cell_substeps_per_s * WENO5_VALU_OPS_PER_CELL
    return {"bound": "valu-fp64", "ops_per_cell_substep": WENO5_VALU_OPS_PER_CELL, "achieved": achieved / 1e12,
            "peak": VALU_F64_LANE_OPS / 1e12, "unit": "Tera fp64 lane-ops/s", "frac": achieved / VALU_F64_LANE_OPS,
            "source": "VALU instructions per cell of the kernel's plane loop (static count on the built library, profiles/r05_isa_mix.txt) x "
                      "measured rate; peak = 78.6 TFLOP/s fp64 vector / 2 flop per FMA lane-op"}


# C5 (4-D, fp32): 57 VALU instructions per cell and plane in round 5's fused_pair4_kernel (static count, profiles/r05_isa_mix.txt; the
# round-4 kernel retired 130 by SQ_INSTS_VALU) against 10.67 algorithmic bytes = 5.3 per byte; the chip's balance is 78.6e12 / 8e12 = 9.8:
# the workload is back on the memory side of the ridge -- its vector ceiling (1.4e12 cell-substeps/s) is above its HBM ceiling (7.5e11),
# and what bounds the kernel is the texture-address path.  fp32 vector peak 157.3 TFLOP/s = 78.6e12 lane-operations/s
# (a packed v_pk_fma_f32 counts as one instruction here)
VALU_F32_LANE_OPS = 78.6e12
C5_VALU_OPS_PER_CELL = float(os.environ.get("HJ_C5_VALU_OPS", "57"))


# C3 (4096^2, ENO3 in the reference's operation order, fp64): 132 VALU instructions per cell and plane in the plane loop of
# fused_pair_kernel<double, HamDoubleIntegrator, ENO3, 256, 1, 2, 2, 2> (tools/isa_mix.py on the built library, profiles/r05_isa_mix.txt) against 21.33 algorithmic bytes = 6.3 per byte; balance 39.3e12 / 8e12 = 4.9:
# beyond the ridge as well -- vector ceiling 2.9e11 cell-substeps/s, HBM ceiling 3.75e11
C3_VALU_OPS_PER_CELL = float(os.environ.get("HJ_C3_VALU_OPS", "132"))


def valu_ceiling_c3(cell_substeps_per_s):
    achieved = cell_substeps_per_s * C3_VALU_OPS_PER_CELL
    return {"bound": "valu-fp64", "ops_per_cell_substep": C3_VALU_OPS_PER_CELL, "achieved": achieved / 1e12,
            "peak": VALU_F64_LANE_OPS / 1e12, "unit": "Tera fp64 lane-ops/s", "frac": achieved / VALU_F64_LANE_OPS,
            "ceiling_cell_substeps_per_s": VALU_F64_LANE_OPS / C3_VALU_OPS_PER_CELL,
            "source": "VALU instructions per cell of the kernel's plane loop (static count on the built library, profiles/r05_isa_mix.txt) x "
                      "measured rate; peak = 78.6 TFLOP/s fp64 vector / 2 flop per FMA lane-op"}


def valu_ceiling_c5(cell_substeps_per_s):
    achieved = cell_substeps_per_s * C5_VALU_OPS_PER_CELL
    return {"bound": "valu-fp32", "ops_per_cell_substep": C5_VALU_OPS_PER_CELL, "achieved": achieved / 1e12,
            "peak": VALU_F32_LANE_OPS / 1e12, "unit": "Tera fp32 lane-ops/s", "frac": achieved / VALU_F32_LANE_OPS,
            "ceiling_cell_substeps_per_s": VALU_F32_LANE_OPS / C5_VALU_OPS_PER_CELL,
            "source": "VALU instructions per cell of the kernel's plane loop (static count on the built library, profiles/r05_isa_mix.txt) x measured "
                      "rate; peak = 157.3 TFLOP/s fp32 vector / 2 flop per FMA lane-op; since round 5 this workload's vector ceiling is ABOVE "
                      "its HBM ceiling: the HBM fraction is the one that grades it"}


def source_hash():
    """Hash of every kernel / host source of the library (csrc/*.h, *.hip)."""
    import glob
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "levelsetpy_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


TRACE_STEPS = 400      # steps of the kernel-trace child pass that are looked at (after its spin-up)
N_SUBSTEP_KERNELS = 8      # the first N entries of STEP_KERNELS are the substep kernels themselves (the rest: helpers of the intended WENO5)
STEP_KERNELS = ("fused_pair_kernel", "fused_substep_kernel", "fused12_pair_kernel", "fused12_kernel", "direct_substep_kernel", "fused_pair4_kernel", "fused_flat4_kernel", "coop_rk_kernel",
                "max_d1sq_kernel", "partials_to_values_kernel", "keys_to_values_kernel", "eps_seam_kernel")


# spin-up of the kernel-trace child pass, in units of the timed leg's fixed spin-up: 12 (round 5; was 4: one box of the round needed ~1500 steps
# to reach its clocks -- the timed leg's first five windows read 23.0 ms, the rest 21.5 -- and the pass's last 400 of 1602 steps were still
# 6.6 % slow, which the roofline (the LARGER of the two times) then inherited)
TRACE_SPIN = int(os.environ.get("HJ_BENCH_TRACE_SPIN", "12"))


def live_traffic(a, n=None, scheme=None, single=None):
    """roofline.traffic and roofline.kernel_ms_rocprof measured IN THIS RUN: three child passes of this script under
    rocprofv3 -- `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (they do not fit one pass, MI355X_MICROARCH.md) and a plain
    `--kernel-trace` pass for the kernel durations (counter collection serialises and slows the dispatches, so the
    durations come from a pass of their own) -- each NSTEP RK3 steps of the workload, started BEFORE this process touches
    the GPU (a process that holds the GPU must not fork/exec on this pool) and finished before the timed legs begin.
    EVERY kernel an RK3 step launches is counted (the substep kernels, the stage-fused one, the epsilon pre-pass of the
    intended WENO5): bytes_per_step = (2*FETCH_SIZE + WRITE_SIZE) KiB summed over them / steps (FETCH_SIZE doubled: the
    guide's gfx950 correction for wide coalesced reads), bytes_per_launch = that / launches of the substep kernels.
    Returns None when rocprofv3 is missing or a pass fails; the caller then falls back to profiles/traffic.json."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    scheme = scheme or a.scheme
    vals, dur, t0 = {}, None, time.perf_counter()
    # single = "C3" / "C5": the counter passes of an `also` workload (round 5: also[...].roofline.traffic measured in the run too); no
    # duration pass for them -- their roofline is priced on the HIP-event time of the timed leg
    for ctr in (("FETCH_SIZE", "WRITE_SIZE") if single else ("FETCH_SIZE", "WRITE_SIZE", None)):
        # counter passes: short (every dispatch is serialised and slow under --pmc; byte counts do not depend on the
        # clocks); the duration pass: the timed leg's own spin-up, so that its kernels run at settled clocks
        # (60 steps of spin-up in the counter passes: on grids of >= 40 M cells the library spends up to 9 x 6 = 54 launches per
        # (scheme, stage class) choosing a tile shape, and the stage-1 class sees ONE launch per step -- the counted steps must
        # run the settled shape: ADVICE r03)
        # (the duration pass spins up TRACE_SPIN x longer than the timed leg's fixed part: it cannot use the leg's settle loop -- its rows
        # are dealt to steps by position -- and has to reach the same clocks)
        # (and it reports the MEDIAN step of its last 400: the mean of a 2 ms tail was hit by a transient once -- 42.4 us per
        # launch where the timed leg and a whole-run trace of the same box both read 38.6-39.0)
        spin, warm, steps = (60, 1, 4) if ctr else (TRACE_SPIN * SPINUP_STEPS, 2, TRACE_STEPS)
        if single:
            spin = 12          # (16.8 M / 277 M cells: below the 40 M cells of the tile-shape tuner or with compile-time tiles)
        nstep = spin + warm + steps
        d = tempfile.mkdtemp(prefix="hj_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp", HJ_BENCH_SPINUP=str(spin), HJ_BENCH_SETTLE_BLOCKS="0")   # (fixed step count: the rows are dealt to steps by position)
        cmd = [exe] + (["--pmc", ctr] if ctr else []) + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-also", "--no-live-traffic",
               "--steps", str(steps), "--warmup", str(warm), "--repeats", "1", "--n", str(n or a.n), "--scheme", scheme,
               "--dtype", a.dtype] + (["--single", single] if single else [])
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180, check=True)
            if ctr:
                rows = []
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    with open(f) as fh:
                        for r in csv.DictReader(fh):
                            k = r["Kernel_Name"]
                            if r["Counter_Name"] == ctr and any(x in k for x in STEP_KERNELS):
                                rows.append((int(r.get("Dispatch_Id", len(rows))), float(r["Counter_Value"]),
                                             any(x in k for x in STEP_KERNELS[:N_SUBSTEP_KERNELS])))
                rows.sort()
                per_step = len(rows) // nstep if rows else 0
                if not per_step:
                    return None
                tail = rows[-per_step * steps:]              # the last `steps` steps: settled tile shape
                nsub = sum(1 for _, _, is_sub in tail if is_sub)
                if not nsub:
                    return None
                vals[ctr] = (sum(v for _, v, _ in tail) / steps, nsub / steps)
            else:
                # kernel-trace only: mean duration of the substep kernels over the LAST `steps` steps (clocks settled)
                rows = []
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    with open(f) as fh:
                        for r in csv.DictReader(fh):
                            if any(x in r["Kernel_Name"] for x in STEP_KERNELS):
                                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                                             any(x in r["Kernel_Name"] for x in STEP_KERNELS[:N_SUBSTEP_KERNELS])))
                rows.sort()
                per_step = len(rows) // nstep if rows else 0
                if per_step:
                    tail = rows[-per_step * steps:]
                    # per step: all its kernels, and its substep kernels alone; the median step speaks for the pass
                    by_step = [tail[i * per_step:(i + 1) * per_step] for i in range(steps)]
                    all_ms = sorted(1e-6 * sum(e - b for b, e, _ in st) for st in by_step)
                    sub_ms = sorted(1e-6 * sum(e - b for b, e, is_sub in st if is_sub) / max(1, sum(1 for _, _, is_sub in st if is_sub)) for st in by_step)
                    dur = {"kernel_ms": sub_ms[len(sub_ms) // 2],                        # substep-kernel dispatch, median step
                           "step_kernels_ms": all_ms[len(all_ms) // 2],                  # all kernels of a step, median step
                           "step_kernels_ms_q1_q3": [all_ms[len(all_ms) // 4], all_ms[(3 * len(all_ms)) // 4]],
                           "dispatches_per_step": per_step}
        except Exception:  # noqa: BLE001
            if ctr:
                return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    per_step = (2 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024
    nsub = vals["FETCH_SIZE"][1]
    return {"bytes_per_launch": per_step / nsub, "bytes_per_step": per_step, "substep_launches_per_step": nsub,
            "fetch_kib_per_step": vals["FETCH_SIZE"][0], "write_kib_per_step": vals["WRITE_SIZE"][0],
            "rocprof": dur,
            "source": ("rocprofv3 child passes of this run (--pmc FETCH_SIZE and --pmc WRITE_SIZE: the last 4 of 17 RK3 steps; every kernel of the "
                       "step counted; %.0f s)" % (time.perf_counter() - t0)) if single else
                      "rocprofv3 child passes of this run (--pmc FETCH_SIZE and --pmc WRITE_SIZE: the last 4 of 65 RK3 steps, --kernel-trace: "
                      "the median step of the last %d of %d; every kernel of the step counted; %.0f s)" % (TRACE_STEPS, TRACE_SPIN * SPINUP_STEPS + 2 + TRACE_STEPS, time.perf_counter() - t0)}


def achievable_rates():
    """Streaming rates this box achieves in the substep's access mixes (tools/ubench/bw2 --quick, a child process run BEFORE
    this one touches the GPU): TB/s of a 16-byte-per-lane copy (1R:1W) and triad (2R:1W, the mix of RK stages 2 and 3) on
    1 GiB arrays (HBM) and on 64 MiB arrays (three of them resident in the 256 MiB Infinity Cache: the regime of the 201^3
    headline, 65 MB per array).  None if the binary is not built (build() compiles it)."""
    exe = os.path.join(ROOT, "tools", "ubench", "bw2")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=120, cwd="/tmp")
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        return {"hbm": d["hbm_1GiB"], "infinity_cache": d["infinity_cache_64MiB"], "unit": "TB/s",
                "source": "tools/ubench/bw2 --quick in this run: best of the grid-stride copy / triad shapes incl. the non-temporal "
                          "one-element-per-thread shape (round 5: 6.4-6.5 TB/s copy on 1 GiB arrays; the MI355X guide quotes 6.29), 16 B per "
                          "lane, 1 GiB arrays (hbm) and 64 MiB arrays (infinity_cache)"}
    except Exception:  # noqa: BLE001
        return None


def measured_traffic(n, scheme, dtype):
    """HBM bytes per RK3 step from the committed rocprofv3 PMC passes (profiles/traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs by tools/profile_round.sh, FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 correction).  Counters cannot be read from inside this process, so the
    figure is only reported when the pass was taken on THIS kernel source (hash of csrc/ recorded with it);
    otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tab = json.load(f)
        rec = tab.get("%d/%s/%s" % (n, scheme, dtype))
        if rec and rec.get("source_hash") == source_hash():
            return rec
    except Exception:  # noqa: BLE001
        pass
    return None


def error_json(a, world, msg):
    """The ONE line a failed run prints instead of a result: no `value`, an `error`, non-zero exit code follows."""
    return json.dumps({"metric": "grid-cell RK-substep updates/sec, Dubins-3D HJI, %d MI355X" % world, "value": None,
                       "unit": "cell-substeps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                       "higher_is_better": True, "error": msg})


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a, argv, script=None, visible_devices=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks from HERE, one process per
    GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment, exactly what
    torch.distributed.run would set), BEFORE this process makes any GPU call -- it never does: it only waits, relays rank 0's
    single JSON line and returns the worst exit code.  A rank that dies takes the others down (their exact PIDs), and when
    rank 0 printed no line of its own an `error` line is printed instead.  Returns the exit code."""
    n = a.gpus
    if visible_devices is None:
        import torch                                    # device_count() does not initialise HIP on this image
        visible_devices = torch.cuda.device_count()
    if not os.environ.get("HJ_BENCH_ONE_DEVICE") and visible_devices < n:
        print(error_json(a, n, "--gpus %d asked for, %d GPU(s) visible: refusing to report a %d-GPU figure from fewer "
                               "devices (HJ_BENCH_ONE_DEVICE=1 rehearses all ranks on GPU 0)" % (n, visible_devices, n)))
        return 2
    env0 = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(free_port()), HJ_BENCH_LAUNCHED="1")
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, cwd=os.getcwd(), text=True,
                                      stdout=subprocess.PIPE if r == 0 else 2))   # other ranks' stdout -> fd 2
    limit = float(os.environ.get("HJ_BENCH_WATCHDOG_S", "900")) + 120.0
    t0, rcs, out0 = time.time(), [None] * n, []
    import threading
    rd = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    rd.start()
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = "rank %d exited with code %d" % (r, rcs[r])
        if failed is None and time.time() - t0 > limit:
            failed = "the %d ranks did not finish within %.0f s" % (n, limit)
        if failed is not None:
            t1 = time.time()
            # give the surviving ranks a moment to report by themselves (collective time-outs), then end exactly the
            # processes started here
            while time.time() - t1 < 20 and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for r, p in enumerate(procs):
                rcs[r] = p.wait()
            break
        time.sleep(0.05)
    rd.join(timeout=10)
    lines = [ln.strip() for ln in out0 if ln.strip().startswith("{")]
    rc = max((abs(c) for c in rcs if c), default=0)
    if lines and (rc == 0 or '"error"' in lines[-1]):
        print(lines[-1])
    else:
        print(error_json(a, n, failed or "rank 0 printed no result line"))
        rc = rc or 1
    sys.stdout.flush()
    return min(rc, 255)


def plan_only(a):
    """--gpus N --plan-only: the N-rank slab leg on paper (levelsetpy_amd.dist.plan_slab_run); no GPU, no process group."""
    from levelsetpy_amd import dist as hjdist
    wname = a.workload or "C4"
    gn = (513 if wname == "C4" else 129) if a.global_n is None else a.global_n
    plan = hjdist.plan_slab_run(a, a.gpus, global_n=gn, workload=wname)
    w = sys.stderr.write
    w("%s %s, %s %s on %d ranks (planes per rank: %s)\n" % (plan["workload"], plan["grid"], plan["scheme"], plan["dtype"], a.gpus,
                                                          "/".join(str(c) for c in plan["planes_per_rank"])))
    for e in plan["ranks"]:
        w("rank %d: planes [%d, %d) = %d, neighbours lo %s hi %s, %.1f MB per array; %s\n" % (
            e["rank"], e["planes"][0], e["planes"][1], e["n_local"], e["lo"], e["hi"], e["slab_bytes_per_array"] / 1e6, e["stepper"]))

        def line(tag, pl):
            if pl:
                w("    %-9s %s: %d workgroups x %d threads = %d tiles (%s) x %d chunks of %d planes, %d per CU -> %d round(s), LDS %d B\n" % (
                    tag, pl["kernel"], pl["workgroups"], pl["threads"], pl["tiles"], "x".join(str(v) for v in pl["tile"]), pl["chunks"],
                    pl["chunk_planes"], pl["workgroups_per_cu"], pl["rounds"], pl["lds_bytes"]))
        if "launches_per_substep" in e:
            line("interior", e["launches_per_substep"]["interior"])
            line("edges", e["launches_per_substep"]["edges"])
        else:
            for st in e["launches_per_step"]:
                line("stage %d in" % st["stage"], st["interior"])
                line("stage %d ed" % st["stage"], st["edges"])
        w("    halo bytes sent per step %.2f MB (%.3f ms per neighbour at the xGMI peak); self-ring prediction %s ms/step\n" % (
            e["halo_bytes_sent_per_step"] / 1e6, e["link_ms_per_step_at_peak"], e["self_ring_ms_per_step"]))
    pr = plan["predicted"]
    w("predicted: compute %s ms/step (slowest rank, self ring), link %.3f ms/step at peak -> %s cell-substeps/s\n" % (
        pr["ms_per_step_compute_self_ring"], pr["ms_per_step_link_at_peak"],
        ("%.3e" % pr["value_cell_substeps_per_s"]) if pr["value_cell_substeps_per_s"] else "n/a"))
    print(json.dumps(plan))


def main():
    a = parse()
    if a.cpu_worker:
        cpu_worker(a.cpu_worker)
        return
    if a.plan_only:
        plan_only(a)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HJ_BENCH_ONE_DEVICE"):      # rehearsal: every rank on GPU 0 (several processes sharing one card)
        local = 0
    if world != a.gpus:
        # a figure labelled N GPUs must come from N ranks: refuse anything else (torchrun with a different
        # --nproc-per-node, a stale WORLD_SIZE in the environment)
        if rank == 0:
            print(error_json(a, a.gpus, "--gpus %d but WORLD_SIZE=%d: the launcher started a different number of ranks" % (a.gpus, world)))
        sys.exit(2)
    slab_leg = world > 1 or bool(os.environ.get("HJ_BENCH_FORCE_SLAB")) or (a.global_n not in (None, 0)) or a.workload is not None
    cpu = None
    if rank == 0 and world == 1 and not slab_leg and not a.no_cpu_baseline:
        cpu = CpuBaseline(a.scheme, a.n)     # before the GPU is touched; idle until the GPU legs are done
    a.live, a.live_also = None, {}
    a.achievable = None
    # (quick runs -- --no-also -- and runs that are themselves being profiled skip the passes)
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if rank == 0 and world == 1 and not slab_leg and not a.no_also and not profiled:
        a.achievable = achievable_rates()    # a child process, before this one touches the GPU
    if rank == 0 and world == 1 and not slab_leg and not a.no_live_traffic and not a.no_also and not profiled:
        a.live = live_traffic(a)             # two rocprofv3 child passes, also before this process touches the GPU
        # the same for the digit entries of --also (513^3: the point where the arrays do not fit the Infinity Cache)
        a.live_also = {x: live_traffic(a, int(x)) for x in a.also.split(",") if x.isdigit()} if a.live else {}
        if a.live and "WENO5" in a.also.split(",") and a.scheme != "WENO5":
            a.live_also["WENO5"] = live_traffic(a, scheme="WENO5")
        for x in ("C3", "C3 fast", "C5"):
            if a.live and x in a.also.split(","):
                a.live_also[x] = live_traffic(a, single=x)
    # stdout carries exactly one JSON line: libraries that print banners to fd 1 (RCCL's version header at
    # communicator creation, for one) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # Watchdog of the multi-rank leg (VERDICT r02 / ADVICE r02): the native path's first real N > 1 run goes through
    # ncclCommInitRank and grouped send/recv that have only ever run as a one-rank self ring.  A rank that sits in a
    # collective the others never reach would hang the driver's scaling run; after HJ_BENCH_WATCHDOG_S seconds
    # (default 900) every rank exits non-zero and rank 0 prints ONE JSON line that carries an "error" instead of a value.
    dog = None
    if slab_leg:
        import threading
        limit = float(os.environ.get("HJ_BENCH_WATCHDOG_S", "900"))

        def bark():
            sys.stderr.write("[bench] rank %d: no result after %.0f s -- a rank is stuck (communicator set-up or a "
                             "collective); aborting\n" % (rank, limit))
            part = getattr(a, "_partial", None)
            if part is not None:
                # the main figure was complete; only the companion leg (the other scaling point, run after it) is stuck
                if rank == 0:
                    part["also"]["companion_leg_error"] = "watchdog: the companion leg did not finish within %.0f s" % limit
                    os.write(json_fd, (json.dumps(part) + "\n").encode())
                os._exit(0)
            if rank == 0:
                err = {"metric": "grid-cell RK-substep updates/sec, Dubins-3D HJI, slab-decomposed over %d MI355X" % world,
                       "value": None, "unit": "cell-substeps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                       "higher_is_better": True, "error": "watchdog: the %d-rank slab leg did not finish within %.0f s" % (world, limit)}
                os.write(json_fd, (json.dumps(err) + "\n").encode())
            os._exit(3)
        dog = threading.Timer(limit, bark)
        dog.daemon = True
        dog.start()

        def arm_companion():
            # the companion leg gets its own, shorter deadline: it must not hold the finished main figure back for long
            t2 = threading.Timer(float(os.environ.get("HJ_BENCH_COMPANION_S", "240")), bark)
            t2.daemon = True
            t2.start()
            a._dog2 = t2
        a._arm_companion = arm_companion
    try:
        out = run(a, rank, world, local, slab_leg, cpu)
    except BaseException as e:
        if cpu is not None:
            cpu.abort()
        if slab_leg and rank == 0 and not isinstance(e, (SystemExit, KeyboardInterrupt)):
            err = {"metric": "grid-cell RK-substep updates/sec, Dubins-3D HJI, slab-decomposed over %d MI355X" % world,
                   "value": None, "unit": "cell-substeps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                   "higher_is_better": True, "error": repr(e)}
            os.write(json_fd, (json.dumps(err) + "\n").encode())
        raise
    finally:
        if dog is not None:
            dog.cancel()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


def run(a, rank, world, local, slab_leg, cpu):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        import datetime
        import torch.distributed as dist
        # collectives that a peer never joins raise after this instead of blocking for torch's default 10 minutes x N
        tmo = datetime.timedelta(seconds=float(os.environ.get("HJ_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        backend = os.environ.get("HJ_BENCH_BACKEND", "nccl")       # "gloo": rehearsal of the multi-rank leg on ONE card
        if backend == "nccl":                                      # (RCCL refuses two ranks on one device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    import levelsetpy_amd as L
    from levelsetpy_amd import _ffi
    from levelsetpy_amd.context import DeviceGrid

    bps = BYTES_PER_SUBSTEP[a.dtype]
    dt_tag = "f64" if a.dtype == "float64" else "f32"
    if slab_leg:
        from levelsetpy_amd import dist as hjdist
        import torch.distributed as dist
        if world == 1:      # rehearsal of the N > 1 leg on one GPU (a single slab, no neighbours)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        wname = a.workload or "C4"
        gn = (513 if wname == "C4" else 129) if a.global_n is None else a.global_n
        res = hjdist.bench_slab(a, rank, world, global_n=gn, workload=wname)
        if res.get("nranks") != world:
            raise RuntimeError("the slab transport spans %r ranks, --gpus says %d" % (res.get("nranks"), world))
        wl = res["workload"]
        bps = BYTES_PER_SUBSTEP[wl["dtype"]]
        dt_tag = "f64" if wl["dtype"] == "float64" else "f32"
        walls = res["walls"]
        tw = torch.tensor(walls, dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)           # per repeat: the slowest rank
        walls = [float(v) for v in tw.cpu()]
        med = statistics.median(walls)
        k = walls.index(sorted(walls)[len(walls) // 2])
        total_cells = res["total_cells"]
        value = total_cells * 3 * a.steps / med
        dev_step_ms = res["devs"][k] / a.steps
        # roofline of the step's launches on THIS rank (rank 0): its own cells over its own device time
        achieved = res["local_cells"] * 3 * bps / (dev_step_ms * 1e-3) / 1e9
        strong = gn > 0 or wname != "C4"
        if wname == "C4":
            grid_txt = ("%d x %d x %d" % (gn, gn, gn)) if strong else ("%dx%d x %d x %d" % (world, a.n, a.n, a.n))
            what = "Dubins-3D HJI %s %s" % ("%d^3" % gn if strong else "%d^3 per GPU" % a.n, "fp64" if dt_tag == "f64" else "fp32")
        else:
            grid_txt, what = wl["grid_txt"], "double-pendulum 4-D HJI %s fp32 periodic" % wl["grid_txt"]
        q = quartiles(walls)
        out = {
            "metric": "grid-cell RK-substep updates/sec, %s, slab-decomposed over %d MI355X" % (what, world),
            "value": value, "unit": "cell-substeps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * med / a.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": dt_tag, "data": "synthetic",
            "config": {"workload": "%s; %s grid slab-decomposed over %d along axis 0 (%s planes per rank)"
                                   % (wl["desc"], grid_txt, world, res["planes"]),
                       "scheme": wl["scheme"], "parallelism": res["parallelism"], "substeps_per_step": 3,
                       "spinup_steps": SPINUP_STEPS, "rccl_nranks": res.get("nranks"), "world_size": world,
                       "launched_by": "bench.py itself (one child process per rank)" if os.environ.get("HJ_BENCH_LAUNCHED")
                                      else "an external launcher (torch.distributed.run)" if world > 1 else "single process"},
            "repeats": {"n": len(walls), "value_min": total_cells * 3 * a.steps / max(walls),
                        "value_max": total_cells * 3 * a.steps / min(walls), "spread": (max(walls) - min(walls)) / med,
                        "iqr_over_median": (q[2] - q[0]) / med},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "traffic_note": "counters are not collected in the slab leg (rocprofv3 --pmc serialises the "
                                         "streams the overlap depends on); the undivided grid's traffic is in the N=1 line",
                         "kernel": "%s (rank 0's launches of one RK3 step on its slab: %s)" % (res["kernel"], res["launches"]),
                         "kernel_ms": dev_step_ms, "algorithmic_bytes_per_launch": res["local_cells"] * 3 * bps},
            "per_gpu_value": value / world,
            "also": {"slab_check_max_abs_diff": res.get("slab_check_max_abs_diff"), "stepper": res.get("transport"),
                     # the exchange alone and the launches alone on this rank (and the slowest / fastest rank): link or kernel?
                     "time_of_a_step": res.get("diagnostics")},
        }
        # The companion point of the scaling curve, same ranks, same communicator set-up: BASELINE's metric names 201^3 and
        # configs[3] names 513^3, so the strong-scaling line also carries the WEAK-scaling figure (a 201-plane slab of an
        # (N*201) x 201 x 201 grid per rank: per-GPU work fixed) and vice versa.  Any failure of this extra leg is reported in
        # place of its number; it cannot take the main figure down (it runs after it) unless a rank hangs (watchdog).
        if world > 1 and wname == "C4" and not a.no_also:
            a._partial = out          # from here on a stuck rank costs the companion figure only (watchdog in main())
            if hasattr(a, "_arm_companion"):
                a._arm_companion()
            try:
                import copy as _copy
                a2 = _copy.copy(a)
                a2.steps, a2.repeats = max(10, a.steps // 2), min(9, a.repeats)
                other_gn = 0 if strong else 513
                r2 = hjdist.bench_slab(a2, rank, world, global_n=other_gn, workload="C4")
                tw2 = torch.tensor(r2["walls"], dtype=torch.float64, device="cuda")
                dist.all_reduce(tw2, op=dist.ReduceOp.MAX)
                med2 = statistics.median([float(v) for v in tw2.cpu()])
                v2 = r2["total_cells"] * 3 * a2.steps / med2
                key = ("weak scaling: %d^3 per GPU" % a.n if other_gn == 0 else "strong scaling: 513^3") + " over %d MI355X" % world
                out["also"][key] = {"value": v2, "per_gpu_value": v2 / world, "ms_per_step": 1e3 * med2 / a2.steps, "steps": a2.steps,
                                    "scaling": "weak" if other_gn == 0 else "strong", "planes_per_rank": r2["planes"],
                                    "parallelism": r2["parallelism"], "slab_check_max_abs_diff": r2.get("slab_check_max_abs_diff"),
                                    "roofline_frac_per_gpu": (v2 / world) * bps / 1e9 / HBM_PEAK_GBS}
            except Exception as e:  # noqa: BLE001
                out["also"]["companion_leg_error"] = repr(e)
            if getattr(a, "_dog2", None) is not None:
                a._dog2.cancel()
            a._partial = None
        # The OTHER native stepper on the same slabs (per-substep exchange <-> one deep-halo exchange per step), same protocol: which of
        # the two is faster depends on the link, and the choice above was made on a one-GPU self ring.  `value` is the faster one, the
        # other stays in `also`.  A failure here costs this extra figure only.
        alt = res.get("alternate_transport")
        if alt and not a.no_also and os.environ.get("HJ_BENCH_SLAB_ALT", "1") != "0":
            a._partial = out
            if hasattr(a, "_arm_companion"):
                a._arm_companion()
            try:
                import copy as _copy
                a3 = _copy.copy(a)
                a3.repeats = min(9, a.repeats)
                os.environ["HJ_BENCH_SPINUP"] = str(min(60, SPINUP_STEPS))
                try:
                    r3 = hjdist.bench_slab(a3, rank, world, global_n=gn, workload=wname, transport=alt, diagnostics=False)
                finally:
                    os.environ["HJ_BENCH_SPINUP"] = str(SPINUP_STEPS)
                tw3 = torch.tensor(r3["walls"], dtype=torch.float64, device="cuda")
                dist.all_reduce(tw3, op=dist.ReduceOp.MAX)
                med3 = statistics.median([float(v) for v in tw3.cpu()])
                v3 = r3["total_cells"] * 3 * a.steps / med3
                first = {"stepper": res.get("transport"), "value": out["value"], "ms_per_step": out["ms_per_step"], "parallelism": res["parallelism"]}
                second = {"stepper": alt, "value": v3, "ms_per_step": 1e3 * med3 / a.steps, "parallelism": r3["parallelism"],
                          "slab_check_max_abs_diff": r3.get("slab_check_max_abs_diff"), "repeats": a3.repeats}
                if v3 > out["value"]:
                    dev3 = r3["devs"][0] / a.steps
                    out["value"], out["ms_per_step"], out["per_gpu_value"] = v3, 1e3 * med3 / a.steps, v3 / world
                    out["config"]["parallelism"] = r3["parallelism"]
                    out["roofline"]["achieved"] = r3["local_cells"] * 3 * bps / (dev3 * 1e-3) / 1e9
                    out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
                    out["roofline"]["kernel_ms"] = dev3
                    out["also"]["stepper"] = alt
                    out["also"]["the other stepper (slower here)"] = first
                else:
                    out["also"]["the other stepper (slower here)"] = second
            except Exception as e:  # noqa: BLE001
                out["also"]["other_stepper_error"] = repr(e)
            if getattr(a, "_dog2", None) is not None:
                a._dog2.cancel()
            a._partial = None
        dist.destroy_process_group()
        return out

    # ---------------------------------------------------------------- single GPU: BASELINE C2 (+ also)
    wl = workload(L, _ffi, torch, a.single, None, None, 0) if a.single else workload(L, _ffi, torch, "dubins", a.scheme, a.dtype, a.n)
    r = time_single(torch, _ffi, DeviceGrid, wl, a.steps, a.warmup, a.repeats, SPINUP_STEPS)
    s = summarize(r, a.steps)
    cells = r["cells"]
    tr = getattr(a, "live", None)
    if tr is None:
        tr = measured_traffic(a.n, a.scheme, a.dtype)
        if tr is not None:
            tr = dict(tr, source="profiles/traffic.json (PMC passes of tools/profile_round.sh on these kernel sources)")
    out = {
        "metric": "grid-cell RK-substep updates/sec, Dubins-3D HJI %d^3 %s" % (a.n, "fp64" if a.dtype == "float64" else "fp32"),
        "value": s["value"], "unit": "cell-substeps/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": s["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dt_tag, "data": "synthetic",
        "config": {"workload": r["desc"], "scheme": a.scheme, "parallelism": "single", "substeps_per_step": 3,
                   "launches_per_step": r["launches_per_step"], "spinup_steps": SPINUP_STEPS + r["settle_steps"],
                   "spinup_note": "untimed: %d fixed steps + %d steps in blocks of K until three blocks in a row agreed within 0.5 %% (clock "
                                  "ramp), then the W warm-up steps" % (SPINUP_STEPS, r["settle_steps"]),
                   "cfl_reduction": "static bound, skipped in launch (alpha is data independent: dt = factorCFL * "
                                    "hj_static_step_bound, equal to the reduced bound by test; value with the reduction kept "
                                    "in every launch: config.cfl_reduction_kept_value)"},
        "repeats": s["repeats"],
        # the step's launches as one unit: algorithmic bytes of an RK3 step (8 words per cell) over the HIP-event
        # time of a step's launches, back to back on the ctx stream (median repeat)
        "roofline": roofline_obj(r, s, tr, cells * 3 * bps, a.achievable),
        "per_gpu_value": s["value"],
    }
    # (`frac` prices the DEVICE time of the step's launches -- HIP events / kernel trace --, `value` the wall time of the K-step window;
    #  the same fraction on the wall clock, i.e. what follows from `value` alone:)
    out["roofline"]["frac_from_value"] = s["value"] * bps / 1e9 / HBM_PEAK_GBS
    out["roofline"]["frac_note"] = "frac: algorithmic bytes / device time of a step's launches; frac_from_value: value x 21.33 B (fp64) / 8 TB/s (wall clock)"
    if a.achievable:
        out["achievable_streaming_rates"] = a.achievable
    also = {}
    if not a.no_also:
        for name in [x for x in a.also.split(",") if x]:
            try:
                if name == "CFL":
                    # the headline launches skip the in-kernel CFL reduction (alpha of the native Hamiltonians is data
                    # independent: hj_rk_step takes dt from hj_static_step_bound, identical by test); this is the same
                    # step with every launch reducing its bound anyway (HJ_KEEP_BOUNDS=1, read when a ctx is created)
                    wl2 = workload(L, _ffi, torch, "dubins", a.scheme, a.dtype, a.n)
                    rA, r2 = time_interleaved(torch, _ffi, DeviceGrid, wl2, {"HJ_KEEP_BOUNDS": "1"}, a.steps, a.warmup, min(21, a.repeats), SPINUP_STEPS)
                    sA, s2 = summarize(rA, a.steps), summarize(r2, a.steps)
                    also["%d^3 with the CFL reduction kept in every launch" % a.n] = {
                        "value": s2["value"], "ms_per_step": s2["ms_per_step"], "roofline_frac": s2["frac"],
                        "roofline_frac_from_value": s2["value"] * bps / 1e9 / HBM_PEAK_GBS, "repeats": s2["repeats"],
                        "interleaved_with": {"what": "windows of the default launches on the same arrays, alternating (same clocks, same cache state)",
                                             "value": sA["value"], "ms_per_step": sA["ms_per_step"]},
                        "vs_headline_interleaved": s2["value"] / sA["value"],
                        "vs_headline": s2["value"] / s["value"]}
                    del rA
                    out["config"]["cfl_reduction_kept_value"] = s2["value"]
                    del r2, wl2
                    continue
                if name == "RTC":
                    also.update(time_runtime_ham(L, torch, _ffi, DeviceGrid, a, s))
                    continue
                if name == "RANGE":
                    also.update(time_range_ham(L, torch, a))
                    continue
                if name == "API":
                    rep = min(15, a.repeats)
                    also["%d^3 via odeCFL3 (tensor)" % a.n] = time_api(L, torch, a.n, "odeCFL3", a.scheme, a.steps, a.warmup, rep, SPINUP_STEPS)
                    also["%d^3 via HJIPDE_solve" % a.n] = time_api(L, torch, a.n, "solve", a.scheme, a.steps, a.warmup, rep, SPINUP_STEPS)
                    also["51^3 singleStep"] = time_api(L, torch, 51, "odeCFL3", a.scheme, max(a.steps, 200), a.warmup, rep, SPINUP_STEPS)
                    also["%d^3 NumPy-in loop" % a.n] = time_api(L, torch, a.n, "numpy", a.scheme, a.steps, a.warmup, rep, SPINUP_STEPS)
                    also["%d^3 NumPy-in loop" % a.n]["vs_tensor_in"] = also["%d^3 NumPy-in loop" % a.n]["value"] / also["%d^3 via odeCFL3 (tensor)" % a.n]["value"]
                    for k in ("%d^3 via odeCFL3 (tensor)" % a.n, "%d^3 via HJIPDE_solve" % a.n, "%d^3 NumPy-in loop" % a.n):
                        also[k]["vs_raw_c_loop"] = also[k]["value"] / s["value"]
                    continue
                if name in ("WENO5", "ENO3", "ENO2", "WENO5_ASSHIPPED"):
                    if name == a.scheme:
                        continue
                    wl2, st, key = workload(L, _ffi, torch, "dubins", name, a.dtype, a.n), a.steps, "%d^3 %s" % (a.n, name)
                elif name.isdigit():
                    wl2, st, key = workload(L, _ffi, torch, "dubins", a.scheme, a.dtype, int(name)), max(10, a.steps // 4), "%s^3 %s" % (name, a.scheme)
                else:
                    wl2, st, key = workload(L, _ffi, torch, name, None, None, 0), max(10, a.steps // 4), name
                r2 = time_single(torch, _ffi, DeviceGrid, wl2, st, max(2, a.warmup // 2), min(9, a.repeats), max(20, SPINUP_STEPS // 3))
                s2 = summarize(r2, st)
                lt = getattr(a, "live_also", {}).get(name)
                ro = roofline_obj(r2, s2, lt, r2["cells"] * 3 * BYTES_PER_SUBSTEP[r2["dtype"]], a.achievable)
                also[key] = {"workload": r2["desc"], "dtype": "f64" if r2["dtype"] == "float64" else "f32", "steps": st,
                             "value": s2["value"], "ms_per_step": s2["ms_per_step"], "roofline_frac": ro["frac"],
                             "achieved_GBps": ro["achieved"], "repeats": s2["repeats"], "kernel": r2["kernel"], "roofline": ro}
                if name == "WENO5":
                    # the intended WENO5 is fp64-VALU bound, not HBM bound: its own ceiling
                    also[key]["roofline_valu"] = valu_ceiling(s2["value"])
                if name == "C5":
                    also[key]["roofline_valu"] = valu_ceiling_c5(s2["value"])
                if name == "C3":
                    also[key]["roofline_valu"] = valu_ceiling_c3(s2["value"])
                del r2, wl2
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001 -- an extra workload must not take the headline down
                import traceback
                traceback.print_exc(file=sys.stderr)
                also[name] = {"error": repr(e)}
    if also:
        out["also"] = also
    if cpu is not None:
        out["cpu_baseline"] = cpu.run()
        # the oracle's state at the headline size, already paid for by the baseline leg, checks the product path
        out["parity"] = gpu_parity(L, torch, cpu.c2, a.scheme)
    # LAST in the line (a log tail of a few thousand characters keeps it): one entry per leg -- name, value, fraction of 8 TB/s on the
    # wall clock, ms per step
    summ = ["headline %d^3 %s: %.4g  frac %.3f  %.4f ms" % (a.n, a.scheme, s["value"], out["roofline"]["frac_from_value"], s["ms_per_step"])]
    for k, v in also.items():
        if isinstance(v, dict) and "value" in v and "ms_per_step" in v:
            b = BYTES_PER_SUBSTEP["float32" if v.get("dtype") == "f32" else "float64"]
            summ.append("%s: %.4g  frac %.3f  %.4f ms" % (k[:60], v["value"], v["value"] * b / 1e9 / HBM_PEAK_GBS, v["ms_per_step"]))
        elif isinstance(v, dict) and "error" in v:
            summ.append("%s: ERROR %s" % (k[:60], str(v["error"])[:80]))
    out["summary"] = summ
    return out


if __name__ == "__main__":
    main()
