"""CPU, world_size 2, gloo: the slab decomposition, halo exchange and RK sequencing of
levelsetpy_amd.dist (the N>1 path) with the arithmetic supplied by the CPU oracle.  The same
SlabIntegrator drives the HIP backend on GPUs; here it must reproduce the undivided oracle run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from levelsetpy_amd import _ffi  # noqa: E402
from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HaloExchanger, HALO  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402


class OracleSlabBackend(object):
    """Test double for HipSlabBackend: padded CPU buffers, arithmetic by the NumPy oracle."""
    device = torch.device("cpu")

    def __init__(self, og_global, slab, scheme, make_sys=None):
        self.slab, self.scheme = slab, scheme
        b, e = slab.begin, slab.end
        g = O.Grid(og_global.min, og_global.max, og_global.N.ravel(), None)
        g.bc = list(og_global.bc)
        g.vs = [v.copy() for v in og_global.vs]
        g.vs[0] = g.vs[0][b:e]
        g.xs = np.meshgrid(*g.vs, indexing='ij')
        g.shape = (e - b,) + og_global.shape[1:]
        self.g = g
        self.sys = make_sys(g) if make_sys is not None else O.DubinsRel(g, 1, 1)
        self.n = e - b
        self.eps = None

    def alloc(self):
        return torch.zeros((self.n + 2 * HALO,) + self.g.shape[1:], dtype=torch.float64)

    def _halo(self, buf):
        lo = buf[:HALO].numpy() if self.slab.halo_lo else None
        hi = buf[self.n + HALO:].numpy() if self.slab.halo_hi else None
        return (lo, hi)

    def substep(self, stage, dt, y, y0, out, p0, p1):
        yi = y[HALO:HALO + self.n].numpy()
        ydot, _ = O.term_lax_friedrichs(self.g, self.sys, self.scheme, 0., yi.reshape(-1), self._halo(y),
                                        self.eps, deriv_range=getattr(self, "deriv_range", None), diss=getattr(self, "diss", "glf"))
        ye = yi + dt * ydot.reshape(yi.shape)
        if stage == _ffi.STAGE_EULER:
            o = ye
        else:
            y0i = y0[HALO:HALO + self.n].numpy()
            if stage == _ffi.STAGE_RK3_HALF:
                o = 0.25 * (3 * y0i + ye)
            elif stage == _ffi.STAGE_RK3_FULL:
                o = (1 / 3) * (y0i + 2 * ye)
            else:
                o = 0.5 * (y0i + ye)
        out[HALO + p0:HALO + p1] = torch.from_numpy(np.ascontiguousarray(o[p0:p1]))

    def local_alpha_max(self):
        return [float(np.max(self.sys.dissipation(0, None, None, None, None, d))) for d in range(self.g.dim)]

    # ---- alpha depending on the costate range (SlabIntegrator(dynamic=True)): [max_d ..., -min_d ...] as float64, MAX-reducible
    def range_pass(self, y):
        yi = y[HALO:HALO + self.n].numpy()
        lo, hi = O.costate_range(self.g, self.scheme, yi.reshape(-1), self._halo(y), self.eps)
        return torch.tensor([float(v) for v in hi] + [-float(v) for v in lo], dtype=torch.float64)

    def set_range(self, v):
        D = self.g.dim
        v = [float(x) for x in v]
        self.deriv_range = ([-x for x in v[D:]], v[:D])

    def alpha_max_now(self):
        lo, hi = self.deriv_range
        return [float(np.max(self.sys.dissipation(0, None, lo, hi, None, d))) for d in range(self.g.dim)]

    def set_dissipation(self, kind):
        self.diss = kind

    def local_bound(self, y):
        yi = y[HALO:HALO + self.n].numpy()
        return O.term_lax_friedrichs(self.g, self.sys, self.scheme, 0., yi.reshape(-1), self._halo(y), self.eps,
                                     deriv_range=getattr(self, "deriv_range", None), diss=self.diss)[1]

    def max_d1sq(self, y):
        yi = y[HALO:HALO + self.n].numpy()
        v = [O.max_d1_squared(self.g, yi, d, self._halo(y) if d == 0 else None) for d in range(self.g.dim)]
        return torch.tensor(v, dtype=torch.float64)

    def set_weno_eps(self, v):
        self.eps = [float(x) for x in v]

    def on_comm_stream(self, fn):
        return fn()

    def join_comm(self, reqs, finish):
        finish(reqs)

    def sync(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [("WENO5_ASSHIPPED", False, 3), ("ENO3", True, 3), ("WENO5", False, 2), ("ENO2", True, 1)]
N, NSTEPS = (13, 8, 9), 2


def _data(og):
    return O.shape_cylinder(og, 2, None, .5) + 0.05 * np.random.default_rng(3).standard_normal(og.shape)


def _worker(rank, world, port, q, cases=None, N=N):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for ci, (scheme, periodic0, order) in enumerate(cases or CASES):
            pd = [0, 2] if periodic0 else [2]
            og = O.Grid([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / N[2])], N, pd)
            data = _data(og)
            slab = SlabDecomposition(N[0], world, rank, periodic0)
            be = OracleSlabBackend(og, slab, scheme)
            integ = SlabIntegrator(slab, be, [float(v) for v in og.dx.ravel()], order, 0.8,
                                   needs_eps=(scheme == "WENO5"))
            integ.set_state(torch.from_numpy(np.ascontiguousarray(data[slab.begin:slab.end])))
            t = 0.0
            for _ in range(NSTEPS):
                t, _dt = integ.step(t)
            q.put((ci, rank, slab.begin, slab.end, t, integ.step_bound, integ.state().numpy().copy()))
    finally:
        dist.destroy_process_group()


def _reference(scheme, periodic0, order, N=N):
    pd = [0, 2] if periodic0 else [2]
    og = O.Grid([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / N[2])], N, pd)
    data = _data(og)
    sys_ = O.DubinsRel(og, 1, 1)
    term = lambda t, y: O.term_lax_friedrichs(og, sys_, scheme, t, y)  # noqa: E731
    ode = {1: O.ode_cfl_1, 2: O.ode_cfl_2, 3: O.ode_cfl_3}[order]
    y, t = data.reshape(-1, 1), 0.0
    for _ in range(NSTEPS):
        t, y = ode(term, [t, 10.], y, 0.8, single_step=True)
    return t, y.reshape(og.shape), term(0., data.reshape(-1, 1))[1]


def test_two_rank_slab_runs_equal_single_domain():
    """All CASES in one pair of gloo ranks (one spawn): extrapolated and periodic axis 0 (the ring
    closes rank 1 <-> 0 through the same peer), RK1/2/3, ENO2's width-2 stencil inside the 3-plane
    halo, and true WENO5 with its all-reduced epsilon."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2 * len(CASES))]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for ci, (scheme, periodic0, order) in enumerate(CASES):
        t_ref, y_ref, sb_ref = _reference(scheme, periodic0, order)
        got = np.zeros(N)
        parts = [r for r in res if r[0] == ci]
        assert len(parts) == 2
        for (_ci, _rank, b, e, t, sb, y) in parts:
            got[b:e] = y
            assert abs(t - t_ref) <= 1e-14
            assert abs(sb - sb_ref) <= 1e-14 * sb_ref        # all-reduced alpha maxima
        assert np.max(np.abs(got - y_ref)) <= 1e-12, (scheme, periodic0, order)


CASES8 = [("WENO5_ASSHIPPED", False, 3), ("ENO3", True, 3), ("WENO5", False, 2)]
N8 = (41, 7, 8)       # 41 = 8*5 + 1: slabs of 6, 5, 5, ... planes (BASELINE C4's 513 = 8*64 + 1 pattern)


def test_eight_rank_uneven_slabs_equal_single_domain():
    """World size 8 (the node size BASELINE C4 names) over gloo with uneven slabs: interior ranks with two
    neighbours, end ranks with one, the periodic ring closed 7 <-> 0, all-reduced alpha maxima and WENO
    epsilon over 8 ranks."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, CASES8, N8)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world * len(CASES8))]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for ci, (scheme, periodic0, order) in enumerate(CASES8):
        t_ref, y_ref, sb_ref = _reference(scheme, periodic0, order, N8)
        got = np.full(N8, np.nan)
        parts = [r for r in res if r[0] == ci]
        assert sorted(e - b for (_c, _r, b, e, _t, _s, _y) in parts) == [5] * 7 + [6]
        for (_ci, _rank, b, e, t, sb, y) in parts:
            got[b:e] = y
            assert abs(t - t_ref) <= 1e-14
            assert abs(sb - sb_ref) <= 1e-14 * sb_ref
        assert np.max(np.abs(got - y_ref)) <= 1e-12, (scheme, periodic0, order)


# ---------------------------------------------------------------- BASELINE C5 as decomposed: 4-D, every axis periodic
N4 = (13, 5, 6, 5)
CASES4 = [("WENO5_ASSHIPPED", 3), ("WENO5", 2)]


def _grid4(N=N4):
    gmin = [-np.pi, -8., -np.pi, -8.]
    gmax = [np.pi * (1 - 2 / N[0]), 8 * (1 - 2 / N[1]), np.pi * (1 - 2 / N[2]), 8 * (1 - 2 / N[3])]
    return O.Grid(gmin, gmax, N, [0, 1, 2, 3])


def _data4(og):
    return O.shape_sphere(og, None, .5) + 0.05 * np.random.default_rng(4).standard_normal(og.shape)


def _worker4(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for ci, (scheme, order) in enumerate(CASES4):
            og = _grid4()
            data = _data4(og)
            slab = SlabDecomposition(N4[0], world, rank, True)
            be = OracleSlabBackend(og, slab, scheme, make_sys=lambda g: O.DoublePendulum4D(g, 1.0))
            integ = SlabIntegrator(slab, be, [float(v) for v in og.dx.ravel()], order, 0.8, needs_eps=(scheme == "WENO5"))
            integ.set_state(torch.from_numpy(np.ascontiguousarray(data[slab.begin:slab.end])))
            t = 0.0
            for _ in range(NSTEPS):
                t, _dt = integ.step(t)
            q.put((ci, rank, slab.begin, slab.end, t, integ.step_bound, integ.state().numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_c5_four_d_all_periodic_ring_of_slabs_equals_single_domain(world):
    """The double-pendulum 4-D grid with every axis periodic (BASELINE configs[4]) over `world` gloo ranks: the ring of
    slabs closes rank world-1 <-> 0 (world 2: both neighbours are the same peer), the drift depends on the axis-0
    coordinate of the slab (the per-slab alpha maxima are all-reduced), slabs of 4 / 3 planes at world 4."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world * len(CASES4))]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    og = _grid4()
    data = _data4(og)
    sys_ = O.DoublePendulum4D(og, 1.0)
    for ci, (scheme, order) in enumerate(CASES4):
        term = lambda t, y: O.term_lax_friedrichs(og, sys_, scheme, t, y)  # noqa: E731
        ode = {2: O.ode_cfl_2, 3: O.ode_cfl_3}[order]
        y, t_ref = data.reshape(-1, 1), 0.0
        for _ in range(NSTEPS):
            t_ref, y = ode(term, [t_ref, 10.], y, 0.8, single_step=True)
        sb_ref = term(0., data.reshape(-1, 1))[1]
        got = np.full(N4, np.nan)
        parts = [r for r in res if r[0] == ci]
        assert len(parts) == world
        for (_ci, _rank, b, e, t, sb, ys) in parts:
            got[b:e] = ys
            assert abs(t - t_ref) <= 1e-14
            assert abs(sb - sb_ref) <= 1e-14 * sb_ref
        assert np.max(np.abs(got - y.reshape(N4))) <= 1e-11, (scheme, order, world)


# ---------------------------------------------------------------- alpha depending on the costate range, decomposed
class BurgersDriftOracle(object):
    """H = |p|^2/2 + c x0 p1, alpha_d = max(|derivMin_d|, |derivMax_d|) (+ |c x0| for d = 1): artificial_diss_glf.py:80-99's protocol."""

    def __init__(self, grid, c):
        self.grid, self.c = grid, c

    def hamiltonian(self, t, data, p, sd=None):
        return 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + self.c * self.grid.xs[0] * p[1]

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        a = np.maximum(np.abs(dmin[dim]), np.abs(dmax[dim]))        # scalars (GLF) or the node's own range as arrays (LLF / LLLF)
        if np.ndim(a) == 0:
            a = float(a)
        return a + np.abs(self.c * self.grid.xs[0]) if dim == 1 else a


class CoupledOracle(object):
    """H = |p|^2/2 + c p0 p1: alpha_0 = max|p0| + |c| max|p1|, alpha_1 = max|p1| + |c| max|p0| -- alpha_i reads the range of ANOTHER dimension, so under
    LLF the grid-wide (all-reduced) range of that dimension matters on every rank, and LLF / LLLF differ."""

    def __init__(self, grid, c):
        self.grid, self.c = grid, c

    def hamiltonian(self, t, data, p, sd=None):
        return 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + self.c * p[0] * p[1]

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        am = lambda d: np.maximum(np.abs(dmin[d]), np.abs(dmax[d]))  # noqa: E731
        a = am(dim) + (abs(self.c) * am(1 - dim) if dim < 2 else 0.0)
        return float(a) if np.ndim(a) == 0 else a


SYSTEMS = {"drift": lambda g: BurgersDriftOracle(g, 0.7), "coupled": lambda g: CoupledOracle(g, 0.6)}
CASES_DYN = [("WENO5_ASSHIPPED", False, 3, "glf", "drift"), ("ENO3", True, 2, "glf", "drift"), ("WENO5", False, 3, "glf", "drift"),
             ("WENO5_ASSHIPPED", True, 3, "llf", "drift"), ("ENO2", False, 2, "lllf", "drift"), ("WENO5", False, 3, "llf", "coupled"),
             ("WENO5_ASSHIPPED", False, 3, "llf", "coupled"), ("WENO5_ASSHIPPED", True, 2, "lllf", "coupled"), ("WENO5_ASSHIPPED", False, 2, "glf", "coupled")]


def _worker_dyn(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for ci, (scheme, periodic0, order, kind, sysname) in enumerate(CASES_DYN):
            pd = [0, 2] if periodic0 else [2]
            og = O.Grid([-1., -1., -1.], [1. - (2. / N[0] if periodic0 else 0.), 1., 1. - 2. / N[2]], N, pd)
            data = O.shape_sphere(og, None, .5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
            slab = SlabDecomposition(N[0], world, rank, periodic0)
            be = OracleSlabBackend(og, slab, scheme, make_sys=SYSTEMS[sysname])
            integ = SlabIntegrator(slab, be, [float(v) for v in og.dx.ravel()], order, 0.8, needs_eps=(scheme == "WENO5"), dynamic=True,
                                   diss=kind)
            integ.set_state(torch.from_numpy(np.ascontiguousarray(data[slab.begin:slab.end])))
            t = 0.0
            for _ in range(NSTEPS):
                t, _dt = integ.step(t)
            q.put((ci, rank, slab.begin, slab.end, t, integ.step_bound, integ.state().numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_range_dependent_alpha_over_ranks_equals_single_domain(world):
    """A Hamiltonian whose alpha reads derivMin / derivMax, slab-decomposed: every substep all-reduces the 2*D range values of the
    slabs, the first stage of a step all-reduces max(alpha) for deltaT; the decomposed run equals the undivided oracle run.
    The local variants (diss='llf' / 'lllf': alpha from the node's own range) all-reduce one scalar, max_x sum_i alpha_i / dx_i, instead."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dyn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world * len(CASES_DYN))]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for ci, (scheme, periodic0, order, kind, sysname) in enumerate(CASES_DYN):
        pd = [0, 2] if periodic0 else [2]
        og = O.Grid([-1., -1., -1.], [1. - (2. / N[0] if periodic0 else 0.), 1., 1. - 2. / N[2]], N, pd)
        data = O.shape_sphere(og, None, .5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
        sys_ = SYSTEMS[sysname](og)
        term = lambda t, y: O.term_lax_friedrichs(og, sys_, scheme, t, y, diss=kind)  # noqa: E731
        ode = {2: O.ode_cfl_2, 3: O.ode_cfl_3}[order]
        y, t_ref, sb_last = data.reshape(-1, 1), 0.0, None
        for _ in range(NSTEPS):
            sb_last = term(t_ref, y)[1]                  # the bound of the step's first stage: what fixed its deltaT
            t_ref, y = ode(term, [t_ref, 10.], y, 0.8, single_step=True)
        got = np.full(N, np.nan)
        parts = [r for r in res if r[0] == ci]
        assert len(parts) == world
        for (_ci, _rank, b, e, t, sb, ys) in parts:
            got[b:e] = ys
            assert abs(t - t_ref) <= 1e-13 * t_ref, (t, t_ref)
            assert abs(sb - sb_last) <= 1e-13 * sb_last
        assert np.max(np.abs(got - y.reshape(N))) <= 1e-11, (scheme, periodic0, order, kind, sysname, world)


def test_slab_decomposition_bookkeeping():
    s = [SlabDecomposition(513, 8, r) for r in range(8)]
    assert [x.n_local for x in s] == [65] + [64] * 7
    assert s[0].begin == 0 and s[-1].end == 513 and all(a.end == b.begin for a, b in zip(s, s[1:]))
    assert s[0].lo is None and s[0].hi == 1 and s[7].hi is None and not s[0].halo_lo and s[0].halo_hi
    p = [SlabDecomposition(129, 8, r, True) for r in range(8)]
    assert p[0].lo == 7 and p[7].hi == 0
    one = SlabDecomposition(20, 1, 0, True)
    assert one.lo is None and one.hi is None        # a single rank wraps inside the kernel
    with pytest.raises(ValueError):
        SlabDecomposition(10, 8, 0)
