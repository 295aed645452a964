"""GPU tests added in round 6.

  * TRANSPOSED MARCH (csrc/hj_instx.hip, hj_fusedv.h XP): the substep launched with the march along grid axis 1 and the slab axis 0 as a
    tile axis must give the bits of the axis-0 march -- whole grids, plane windows, every scheme, periodic / extrapolated mixes, the CFL
    keys, fp32 -- and the slab steppers on top of it the bits of the undivided grid (virtual ranks, self ring).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd import _ffi  # noqa: E402
from levelsetpy_amd.context import DeviceGrid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

from test_gpu_parity import mk, SCHEMES  # noqa: E402

XP_NAME = b"fused_pair_kernel (march along axis 1)"
PAR = [1., 1., 1., 2.]


def _ctx(g, monkeypatch, dtype="float64", **env):
    for k in ("HJ_XP", "HJ_PAIR", "HJ_FORCE_DIRECT", "HJ_MIN_CHUNK", "HJ_FULL_ROWS", "HJ_KEEP_BOUNDS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dg = DeviceGrid(g, dtype)
    dg.bind_stream()
    return dg


def _sub(dg, sid, stage, dt, y, y0, out, p0, p1, slot=3):
    _ffi.check(dg.lib.hj_rk_substep(dg.ctx, sid, _ffi.HAM_DUBINS_REL, _ffi.darr(PAR), 0., stage, dt, 0,
                                    dg.ptr(y), dg.ptr(y0) if y0 is not None else None, dg.ptr(out), slot, p0, p1))


def _rk3_by_substeps(dg, sid, y, dt, windows, eps_src=False):
    """three stages through hj_rk_substep, each computed window by window; returns (y_new, stepBound of the first stage's first window)"""
    a, b, c = torch.full_like(y, 7.0), torch.full_like(y, 7.0), torch.full_like(y, 7.0)
    vals = torch.zeros(4, dtype=y.dtype, device="cuda")
    bounds = []
    for stage, src, y0, dst in ((_ffi.STAGE_EULER, y, None, a), (_ffi.STAGE_RK3_HALF, a, y, b), (_ffi.STAGE_RK3_FULL, b, y, c)):
        if eps_src:       # the intended WENO5 as the slab steppers run it: epsilon from a caller-reduced vector
            _ffi.check(dg.lib.hj_max_d1sq(dg.ctx, dg.ptr(src), dg.ptr(vals)))
            _ffi.check(dg.lib.hj_ctx_set_weno_eps_source(dg.ctx, dg.ptr(vals)))
        for (p0, p1) in windows:
            _sub(dg, sid, stage, dt, src, y0, dst, p0, p1, slot=3)
            sb, am = C.c_double(), (C.c_double * 4)()
            _ffi.check(dg.lib.hj_read_step_bound(dg.ctx, 3, C.byref(sb), am))
            bounds.append((sb.value, tuple(am)[:3]))
    dg.sync()
    return c, bounds


GRIDS = [((37, 21, 19), 2), ((30, 64, 130), (0, 2)), ((12, 50, 40), None), ((65, 48, 77), (1, 2))]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", GRIDS)
def test_transposed_march_equals_axis0_march_bitwise(scheme, n, pd, monkeypatch):
    """One RK3 step through hj_rk_substep, whole grid and plane WINDOWS, marched along axis 1 (HJ_XP=2) against the axis-0 march (HJ_XP=0) and
    the direct kernel: array_equal, and the same CFL keys (stepBound, per-dimension max alpha in GRID order)."""
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
    data = O.shape_cylinder(og, 2, None, .5) + 0.02 * np.random.default_rng(23).standard_normal(og.shape)
    y = torch.as_tensor(data, device="cuda")
    sid = _ffi.SCHEME_IDS[scheme]
    eps_src = scheme == "WENO5"
    n0 = n[0]
    whole = [(0, n0)]
    cuts = [(0, 5), (5, n0 - 4), (n0 - 4, n0)]
    res = {}
    for tag, env, windows in (("axis0", dict(HJ_XP="0", HJ_PAIR="2"), whole), ("xp", dict(HJ_XP="2", HJ_PAIR="2"), whole),
                              ("xp windows", dict(HJ_XP="2", HJ_PAIR="2", HJ_MIN_CHUNK="4"), cuts), ("direct", dict(HJ_XP="0", HJ_FORCE_DIRECT="1"), whole)):
        dg = _ctx(g, monkeypatch, **env)
        out, bounds = _rk3_by_substeps(dg, sid, y, 2e-3, windows, eps_src)
        name = dg.lib.hj_last_kernel(dg.ctx)
        if tag.startswith("xp"):
            assert name == XP_NAME, name
        elif tag == "axis0":
            assert name == b"fused_pair_kernel", name
        res[tag] = (out.clone(), bounds)
    ref, bref = res["axis0"]
    for tag in ("xp", "xp windows", "direct"):
        got, b = res[tag]
        assert torch.equal(got, ref), "%s: %d cells differ, max %.3e" % (tag, int((got != ref).sum()), float((got - ref).abs().max()))
    # whole-grid launches: the reduced keys are the same numbers (alpha maxima per GRID dimension, stepBound)
    assert res["xp"][1] == bref, (res["xp"][1], bref)


@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "ENO3"])
def test_transposed_march_fp32_and_hj_rk_step(scheme, monkeypatch):
    """fp32 instantiations, and hj_rk_step / the clamp of termRestrictUpdate / a fused post-step minimum (MODE 0) under HJ_XP=2."""
    n = (41, 36, 50)
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, 2)
    data = O.shape_cylinder(og, 2, None, .5) + 0.02 * np.random.default_rng(5).standard_normal(og.shape)
    sid = _ffi.SCHEME_IDS[scheme]
    for dtype, tdt in (("float32", torch.float32), ("float64", torch.float64)):
        y = torch.as_tensor(data, device="cuda").to(tdt)
        outs = {}
        for tag, env in (("axis0", dict(HJ_XP="0", HJ_PAIR="2")), ("xp", dict(HJ_XP="2", HJ_PAIR="2"))):
            dg = _ctx(g, monkeypatch, dtype, **env)
            _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, 1))
            cur, nxt, w0, w1 = y.clone(), torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
            tout, dtout = C.c_double(), C.c_double()
            t = 0.
            for _ in range(3):
                _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, _ffi.darr(PAR), t, 1e9, 0.8, 1e300, -1, dg.ptr(cur), dg.ptr(nxt),
                                             dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
                cur, nxt = nxt, cur
                t = float(tout.value)
            dg.sync()
            if tag == "xp":
                assert dg.lib.hj_last_kernel(dg.ctx) == XP_NAME
            outs[tag] = (t, cur.clone())
        assert outs["xp"][0] == outs["axis0"][0]
        assert torch.equal(outs["xp"][1], outs["axis0"][1]), float((outs["xp"][1] - outs["axis0"][1]).abs().max())


def _field(g, og, n, seed=5):
    rng = np.random.default_rng(seed)
    return O.shape_cylinder(og, 2, None, .5) + 0.1 * np.sin(3 * og.xs[0]) + 0.01 * rng.standard_normal(n)


def _undivided(dg, sid, order, full, steps, dt_cap):
    cur, nxt, w0, w1 = full.clone(), torch.empty_like(full), torch.empty_like(full), torch.empty_like(full)
    tout, dtout = C.c_double(), C.c_double()
    t = 0.
    for _ in range(steps):
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, sid, _ffi.HAM_DUBINS_REL, _ffi.darr(PAR), t, 1e9, 0.8, dt_cap, 0,
                                     dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
        cur, nxt = nxt, cur
        t = float(tout.value)
    torch.cuda.synchronize()
    return t, float(dtout.value), cur


@pytest.mark.parametrize("scheme,periodic0,world,order", [("WENO5_ASSHIPPED", False, 3, 3), ("ENO3", False, 2, 2), ("WENO5_ASSHIPPED", True, 3, 3),
                                                          ("ENO2", True, 5, 3), ("WENO5_ASSHIPPED", False, 5, 1)])
def test_deep_halo_stepper_on_the_transposed_march_virtual_ranks_bitwise(scheme, periodic0, world, order, monkeypatch):
    """hj_slab_rk_step_deep with `world` virtual ranks whose interior launches march along axis 1 (HJ_XP=2: windows of the slab axis next to
    pad planes, end ranks with ghost cells on one side), against the undivided grid stepped by the AXIS-0 march: bitwise."""
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    n = (61 if world <= 3 else 97, 26, 24)
    pd = (0, 2) if periodic0 else 2
    gmax0 = 2. * (1 - 2 / n[0]) if periodic0 else 2.
    g, og = mk([-2., -1.25, -np.pi], [gmax0, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
    data = _field(g, og, n)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    sid = _ffi.SCHEME_IDS[scheme]
    dg = _ctx(g, monkeypatch, HJ_XP="0")
    monkeypatch.setenv("HJ_XP", "2")
    monkeypatch.setenv("HJ_PAIR", "2")
    steppers = []
    for r in range(world):
        slab = SlabDecomposition(n[0], world, r, periodic0)
        steppers.append(NativeSlabStepper(g, slab, sid, _ffi.HAM_DUBINS_REL, PAR, dxs, order=order, deep=True, external=lambda st: None))
    amax = [max(st.alpha_local[d] for st in steppers) for d in range(3)]
    for st in steppers:
        st.set_alpha_max(amax)

    def move_pads():
        torch.cuda.synchronize()
        for st in steppers:
            D, nl, sl = st.pad, st.n, st.slab
            if sl.hi is not None:
                nb = steppers[sl.hi]
                st.buf["cur"][D + nl:D + nl + D].copy_(nb.buf["cur"][nb.pad:nb.pad + D])
            if sl.lo is not None:
                nb = steppers[sl.lo]
                st.buf["cur"][0:D].copy_(nb.buf["cur"][nb.pad + nb.n - D:nb.pad + nb.n])
        torch.cuda.synchronize()

    full = torch.as_tensor(data, device="cuda")
    for st in steppers:
        st.set_state(full[st.slab.begin:st.slab.end])
    move_pads()
    t = 0.
    for _ in range(3):
        ts = [st.step(t) for st in steppers]
        move_pads()
        dt, t = ts[0][1], ts[0][0]
    t_ref, _, cur = _undivided(dg, sid, order, full, 3, dt)
    assert abs(t_ref - t) <= 1e-15
    names = set()
    for st in steppers:
        got = st.state()
        ref = cur[st.slab.begin:st.slab.end]
        assert torch.equal(got, ref), "rank %d differs by %g" % (st.slab.rank, float((got - ref).abs().max()))
        st.close()


@pytest.mark.parametrize("scheme,periodic0,world", [("WENO5_ASSHIPPED", False, 3), ("ENO3", True, 2), ("WENO5", False, 2), ("ENO2", True, 4)])
def test_per_substep_schedule_on_the_transposed_march_thread_ranks_bitwise(scheme, periodic0, world, monkeypatch):
    """SlabIntegrator + HipSlabBackend (edge planes, 3-plane exchange per substep, interior = hj_rk_substep over a plane range of the slab) with
    the interior ranges marched along axis 1 (HJ_XP=2), `world` in-process thread ranks: against the undivided grid on the axis-0 march."""
    import threading
    from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
    from test_gpu_round4 import ThreadRing
    n = (47, 26, 24)
    pd = (0, 2) if periodic0 else 2
    gmax0 = 2. * (1 - 2 / n[0]) if periodic0 else 2.
    g, og = mk([-2., -1.25, -np.pi], [gmax0, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
    full = torch.as_tensor(_field(g, og, n, 11), device="cuda")
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    sid = _ffi.SCHEME_IDS[scheme]
    dg = _ctx(g, monkeypatch, HJ_XP="0")
    monkeypatch.setenv("HJ_XP", "2")
    monkeypatch.setenv("HJ_PAIR", "2")
    tr = ThreadRing(world)
    out, errs, names = {}, [], set()

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                slab = SlabDecomposition(n[0], world, rank, periodic0, self_exchange=periodic0)
                be = HipSlabBackend(g, slab, sid, _ffi.HAM_DUBINS_REL, PAR)
                integ = SlabIntegrator(slab, be, dxs, 3, 0.8, needs_eps=(scheme == "WENO5"), exchanger=tr.exchanger(slab),
                                       allreduce_max=tr.allreduce_max(rank))
                integ.set_state(full[slab.begin:slab.end])
                t = 0.
                for _ in range(3):
                    t, dt = integ.step(t)
                be.sync()
                names.add(be.dg.lib.hj_last_kernel(be.dg.ctx))
                out[rank] = (slab.begin, slab.end, t, dt, integ.state().clone())
                be.sync()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            tr.bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(600)
    assert not errs, errs
    assert len(out) == world and XP_NAME in names, names
    t, dt = out[0][2], out[0][3]
    t_ref, _, ref = _undivided(dg, sid, 3, full, 3, dt)
    assert abs(t - t_ref) <= 1e-15
    for r in range(world):
        b, e, tr_, dtr, ys = out[r]
        if scheme == "WENO5":
            assert float((ys - ref[b:e]).abs().max()) <= 1e-12          # the all-reduced epsilon is reduced in another order
        else:
            assert torch.equal(ys, ref[b:e]), "rank %d differs by %g" % (r, float((ys - ref[b:e]).abs().max()))


@pytest.mark.parametrize("deep", [False, True])
@pytest.mark.parametrize("n,scheme", [((65, 120, 110), "WENO5_ASSHIPPED"), ((23, 40, 300), "ENO3")])
def test_native_rccl_self_ring_on_the_transposed_march_bitwise(n, scheme, deep, monkeypatch):
    """The native steppers (hj_slab_rk_step / _deep: RCCL send / recv inside the C library) as a one-rank periodic ring with the interior launches
    marched along axis 1, five RK3 steps against the in-kernel wrap on the axis-0 march: bitwise (what tools/thin_slab_ring.py times)."""
    import torch.distributed as dist
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    if deep and n[0] < 18:
        pytest.skip("too thin for the deep-halo stepper")
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29596")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        g, og = mk([-2., -1.25, -np.pi], [2. * (1 - 2 / n[0]), 1.25, np.pi * (1 - 2 / n[2])], n, (0, 2))
        full = torch.as_tensor(_field(g, og, n, 3), device="cuda")
        dxs = [float(v) for v in np.asarray(g.dx).ravel()]
        sid = _ffi.SCHEME_IDS[scheme]
        dg = _ctx(g, monkeypatch, HJ_XP="0")
        monkeypatch.setenv("HJ_XP", "2")
        monkeypatch.setenv("HJ_PAIR", "2")
        slab = SlabDecomposition(n[0], 1, 0, True, self_exchange=True)
        nat = NativeSlabStepper(g, slab, sid, _ffi.HAM_DUBINS_REL, PAR, dxs, deep=deep)
        nat.set_state(full)
        t = 0.
        for _ in range(5):
            t, dt = nat.step(t)
        got = nat.state().clone()
        torch.cuda.synchronize()
        nat.close()
        t_ref, _, ref = _undivided(dg, sid, 3, full, 5, dt)
        assert abs(t - t_ref) <= 1e-15
        assert torch.equal(got, ref), float((got - ref).abs().max())
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------------------ the DEFAULT small-grid path against the oracle
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((51, 51, 51), 2), ((24, 31, 40), (0, 2)), ((12, 50, 40), None)])
def test_small_grids_on_the_default_direct_kernel_vs_oracle(scheme, n, pd, monkeypatch):
    """tests/conftest.py sets HJ_DIRECT_BELOW=0 so that the suite's small grids exercise the TILED kernels; the PRODUCT sends 3-D grids below
    140 000 cells (C1: 51^3, every notebook of the reference) to direct_substep_kernel.  Here the switch is at its default: five odeCFL3 steps
    through the drop-in API against the oracle's full array -- ENO2 / ENO3 bit for bit, the WENO5 arithmetics 1e-11 -- and the kernel asserted."""
    from levelsetpy_amd.context import device_grid
    from test_gpu_parity import sdata, DERIV, close
    monkeypatch.delenv("HJ_DIRECT_BELOW", raising=False)
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    osys = O.DubinsRel(og, 1, 1)
    term = lambda tt, yy: O.term_lax_friedrichs(og, osys, scheme, tt, yy)  # noqa: E731
    y, t = d0.reshape(-1, 1), 0.
    yo, to = d0.reshape(-1, 1), 0.
    for _ in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        to, yo = O.ode_cfl_3(term, [to, 10.], yo, 0.8, single_step=True)
    dg = device_grid(g)
    assert dg.lib.hj_last_kernel(dg.ctx) == b"direct_substep_kernel", dg.lib.hj_last_kernel(dg.ctx)
    assert abs(t - to) <= 1e-14
    if scheme.startswith("ENO"):
        assert t == to
        assert np.array_equal(y, yo), "%s: %d cells differ, max %.3e" % (scheme, int((np.asarray(y) != yo).sum()), float(np.abs(np.asarray(y) - yo).max()))
    else:
        close(np.asarray(y), yo, 1e-11, what=scheme)
