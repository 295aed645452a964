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
    for k in ("HJ_XP", "HJ_XP_TRIALS", "HJ_PAIR", "HJ_FORCE_DIRECT", "HJ_MIN_CHUNK", "HJ_FULL_ROWS", "HJ_KEEP_BOUNDS"):
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


# ------------------------------------------------------------------------------ the 4-D full-row kernel (csrc/hj_flat4v.h)
def _p4_field(og, n, seed=5):
    rng = np.random.default_rng(seed)
    return O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2]) + 0.02 * rng.standard_normal(n)


@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "ENO2"])
@pytest.mark.parametrize("n,pd", [((8, 7, 9, 41), (0, 1, 2, 3)),        # all periodic, ODD last axis: single-cell slots at the row ends, wrap through the row's pad
                                  ((8, 7, 9, 40), (0, 1, 2, 3)),        # all periodic, even last axis
                                  ((7, 11, 6, 37), (0, 2)),             # axes 1 and 3 extrapolated: ghost rows, ghosts of an odd row's ends
                                  ((9, 5, 13, 34), None),               # nothing periodic, even rows; axis 2 exactly ... several tiles, axis 1 one
                                  ((6, 12, 8, 129), (1, 3)),            # C5's row length; axis 0 extrapolated (ghost planes staged), axis 2 extrapolated
                                  ((5, 3, 5, 131), (0, 1, 2, 3))])      # the longest row a box holds at its pitch, one tile per plane
def test_flat4_kernel_vs_fp64_oracle_pair4_and_direct(scheme, n, pd, monkeypatch):
    """termLaxFriedrichs on 4-D fp32 grids through fused_flat4_kernel (whole rows of the contiguous axis, 16-byte row loads) against the fp64
    oracle (1e-4 of max |ydot|; ENO2 masked as in round 5) and BITWISE against the compile-time-tile kernel, the generic pair kernel and the
    direct kernel -- the same per-cell arithmetic whatever the data path."""
    from test_gpu_configs import pendulum_grid
    from test_gpu_parity import sdata, DERIV
    g, og = pendulum_grid(n, pd)
    data = _p4_field(og, n)
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, 0., data.reshape(-1, 1))
    scale = float(np.max(np.abs(yo)))
    y32 = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=torch.float32)
    got = {}
    want = {"flat4": b"fused_flat4_kernel", "pair": b"fused_pair_kernel", "direct": b"direct_substep_kernel"}
    for name in ("flat4", "pair", "direct"):
        monkeypatch.setenv("HJ_FORCE_DIRECT", "1" if name == "direct" else "0")
        monkeypatch.setenv("HJ_PAIR", "2")
        monkeypatch.setenv("HJ_PAIR4", "0")
        monkeypatch.setenv("HJ_FLAT4", "2" if name == "flat4" else "0")        # (2: also on grids with extrapolated axes, which keep pair4 by default)
        g.__dict__.pop("_hj_device", None)
        yd, sb, _ = L.termLaxFriedrichs(0., y32, sdata(g, L.DoublePendulum4D(g, 1.0), DERIV[scheme]))
        dg = g.__dict__["_hj_device"]
        dg = dg[next(iter(dg))] if isinstance(dg, dict) else dg
        assert dg.lib.hj_last_kernel(dg.ctx) == want[name], (name, dg.lib.hj_last_kernel(dg.ctx))
        assert abs(sb - sbo) <= 1e-5 * sbo, (name, sb, sbo)
        got[name] = yd.cpu().numpy().astype(np.float64)
    g.__dict__.pop("_hj_device", None)
    rel = np.abs(got["flat4"] - yo) / scale
    if scheme.startswith("WENO"):
        assert rel.max() <= 1e-4, rel.max()
    else:
        assert np.mean(rel > 1e-4) <= 2e-3 and rel.max() <= 0.2, (float(np.mean(rel > 1e-4)), rel.max())
    for name in ("pair", "direct"):
        assert np.array_equal(got["flat4"], got[name]), (name, int((got["flat4"] != got[name]).sum()), float(np.max(np.abs(got["flat4"] - got[name]))))


@pytest.mark.parametrize("n,pd", [((37, 10, 12, 67), (0, 1, 2, 3)), ((23, 7, 9, 50), (0, 2))])
def test_flat4_rk3_steps_clamp_ranges_and_bound(n, pd, monkeypatch):
    """hj_rk_substep through the full-row kernel: the Euler / with-y0 / general (clamped) instantiations equal the direct kernel bit for bit, a
    stage computed as three plane ranges equals one launch, the in-kernel CFL maxima equal the definition (the dummy second cell of an odd
    row's last slot must not leak into them)."""
    from test_gpu_configs import pendulum_grid
    g, og = pendulum_grid(n, pd)
    d0 = torch.as_tensor(_p4_field(og, n, 7), device="cuda", dtype=torch.float32).contiguous()
    par = [1.0, 0., 0., 0.]
    dt = 2e-3
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]

    def sub(dg, stage, y, y0, out, p0=0, p1=None, slot=3, rs=0):
        _ffi.check(dg.lib.hj_rk_substep(dg.ctx, sid, _ffi.HAM_DOUBLE_PENDULUM, _ffi.darr(par), 0., stage, dt, rs, dg.ptr(y),
                                        dg.ptr(y0) if y0 is not None else None, dg.ptr(out), slot, p0, n[0] if p1 is None else p1))
    outs = {}
    for name, force in (("flat4", "0"), ("direct", "1")):
        monkeypatch.setenv("HJ_FORCE_DIRECT", force)
        monkeypatch.setenv("HJ_PAIR", "2")
        monkeypatch.setenv("HJ_FLAT4", "2")
        dg = DeviceGrid(g, "float32")
        dg.bind_stream()
        a, b, c, r = dg.empty(), dg.empty(), dg.empty(), dg.empty()
        sub(dg, _ffi.STAGE_EULER, d0, None, a)
        sub(dg, _ffi.STAGE_RK3_HALF, a, d0, b)
        sub(dg, _ffi.STAGE_RK3_FULL, b, d0, c)
        assert dg.lib.hj_last_kernel(dg.ctx) == (b"fused_flat4_kernel" if force == "0" else b"direct_substep_kernel")
        sub(dg, _ffi.STAGE_RK3_FULL, b, d0, r, rs=-1)
        if force == "0":
            c2 = torch.zeros_like(c)
            for k, (p0, p1) in enumerate([(0, 5), (5, 19), (19, n[0])]):
                sub(dg, _ffi.STAGE_RK3_FULL, b, d0, c2, p0, p1, slot=4 + k)
            dg.sync()
            assert torch.equal(c, c2), float((c - c2).abs().max())
            sb, am = C.c_double(), (C.c_double * 4)()
            _ffi.check(dg.lib.hj_read_step_bound(dg.ctx, 3, C.byref(sb), am))
            f = O.DoublePendulum4D(og, 1.0).drift()
            ref = [float(np.max(np.abs(f[0]))), float(np.max(np.abs(f[1]))) + 1.0, float(np.max(np.abs(f[2]))), float(np.max(np.abs(f[3]))) + 1.0]
            for d in range(4):
                assert abs(am[d] - ref[d]) <= 3e-6 * ref[d], (d, am[d], ref[d])
        dg.sync()
        outs[name] = (a, b, c, r)
    for k in range(4):
        assert torch.equal(outs["flat4"][k], outs["direct"][k]), (k, float((outs["flat4"][k] - outs["direct"][k]).abs().max()))
    assert float((outs["flat4"][2] - outs["flat4"][3]).abs().max()) > 0      # the clamp did something


def test_flat4_long_axis0_chunks_the_row_table_and_slab_ring(monkeypatch):
    """Several chunks per tile column (each with its own row table) on a long axis 0; and the deep-halo stepper on a ring of three virtual ranks
    (pad planes, plane windows beyond the slab) bitwise equal to the undivided grid -- all through the full-row kernel."""
    from test_gpu_configs import pendulum_grid
    from test_gpu_round4 import sphere4, undivided_steps, PAR_PENDULUM
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    monkeypatch.setenv("HJ_PAIR", "2")
    n = (61, 6, 7, 45)
    g, _ = pendulum_grid(n, low_mem=True)
    full = sphere4(g, noise=0.01, seed=9)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    world = 3
    steppers = []
    for r in range(world):
        slab = SlabDecomposition(n[0], world, r, True)
        steppers.append(NativeSlabStepper(g, slab, sid, _ffi.HAM_DOUBLE_PENDULUM, PAR_PENDULUM, dxs, "float32", order=3, deep=True, external=lambda st: None))
    amax = [max(st.alpha_local[d] for st in steppers) for d in range(4)]
    for st in steppers:
        st.set_alpha_max(amax)

    def move_pads():
        torch.cuda.synchronize()
        for st in steppers:
            D, nl, sl = st.pad, st.n, st.slab
            nb = steppers[sl.hi]
            st.buf["cur"][D + nl:D + nl + D].copy_(nb.buf["cur"][nb.pad:nb.pad + D])
            nb = steppers[sl.lo]
            st.buf["cur"][0:D].copy_(nb.buf["cur"][nb.pad + nb.n - D:nb.pad + nb.n])
        torch.cuda.synchronize()
    for st in steppers:
        st.set_state(full[st.slab.begin:st.slab.end])
    move_pads()
    t = 0.
    for _ in range(3):
        ts = [st.step(t) for st in steppers]
        move_pads()
        t, dt = ts[0]
    assert steppers[0].dg.lib.hj_last_kernel(steppers[0].dg.ctx) == b"fused_flat4_kernel"
    t_ref, ref = undivided_steps(g, full, "WENO5_ASSHIPPED", 3, 3, dt)
    assert abs(t_ref - t) <= 1e-15
    for st in steppers:
        got, want = st.state(), ref[st.slab.begin:st.slab.end]
        assert torch.equal(got, want), "rank %d differs by %g" % (st.slab.rank, float((got - want).abs().max()))
        st.close()


# ------------------------------------------------------------------------------ one cooperative launch per step on small grids (coop_rk_kernel)
@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "ENO2", "ENO3"])
@pytest.mark.parametrize("order", [2, 3])
@pytest.mark.parametrize("which,n,pd", [("dubins", (51, 51, 51), 2), ("dubins", (24, 31, 40), (0, 2)), ("dubins", (12, 50, 40), None), ("dint", (160, 300), None),
                                        ("dubins32", (33, 40, 21), 2)])
def test_one_cooperative_launch_per_step_equals_the_stage_launches_bitwise(scheme, order, which, n, pd, monkeypatch):
    """hj_rk_step of order 2 / 3 on a grid the direct kernel runs is ONE launch (coop_rk_kernel: the stages inside it, a grid barrier between
    them) with HJ_COOP=1 (default) and `order` launches of direct_substep_kernel with HJ_COOP=0: ten steps, the same bits, the same times;
    with a termRestrictUpdate clamp and a fused post-step minimum on the way."""
    monkeypatch.delenv("HJ_DIRECT_BELOW", raising=False)
    if which == "dint":
        monkeypatch.setenv("HJ_DIRECT_BELOW", "200000")        # (2-D grids have no small-grid switch by default)
        g, og = mk([-1, -1], [1, 1], n, pd)
        ham, par = _ffi.HAM_DOUBLE_INTEGRATOR, [1., 0., 0., 0.]
        data = O.shape_sphere(og, None, .3)
    else:
        g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
        ham, par = _ffi.HAM_DUBINS_REL, PAR
        data = O.shape_cylinder(og, 2, None, .5)
    dtype, tdt = ("float32", torch.float32) if which.endswith("32") else ("float64", torch.float64)
    y = torch.as_tensor(data + 0.02 * np.random.default_rng(3).standard_normal(og.shape), device="cuda").to(tdt)
    sid = _ffi.SCHEME_IDS[scheme]
    res = {}
    for coop in ("1", "0"):
        monkeypatch.setenv("HJ_COOP", coop)
        dg = DeviceGrid(g, dtype)
        dg.bind_stream()
        _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, 1))
        cur, nxt, w1 = y.clone(), torch.empty_like(y), torch.empty_like(y)
        tout, dtout = C.c_double(), C.c_double()
        t = 0.
        for k in range(10):
            # (the callers' own aliasing: for RK3 the first stage buffer doubles as the output)
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, sid, ham, _ffi.darr(par), t, 1e9, 0.8, 1e300, -1 if k % 2 else 0, dg.ptr(cur), dg.ptr(nxt),
                                         dg.ptr(nxt) if order == 3 else dg.ptr(w1), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur
            t = float(tout.value)
        dg.sync()
        assert dg.lib.hj_last_kernel(dg.ctx) == (b"coop_rk_kernel" if coop == "1" else b"direct_substep_kernel"), dg.lib.hj_last_kernel(dg.ctx)
        nl, fused = C.c_int(), C.c_int()
        _ffi.check(dg.lib.hj_rk_plan(dg.ctx, order, sid, ham, _ffi.darr(par), 0, C.byref(nl), C.byref(fused)))
        assert nl.value == (1 if coop == "1" else order)
        res[coop] = (t, cur.clone())
    assert res["1"][0] == res["0"][0]
    assert torch.equal(res["1"][1], res["0"][1]), float((res["1"][1] - res["0"][1]).abs().max())
    assert bool(torch.isfinite(res["1"][1]).all())


# ------------------------------------------------------------------------------ the run-time kernel table's key
def test_runtime_kernel_table_keeps_fast_eno_fp64_and_eno_fp32_apart():
    """Scheme ids run to HJ_ENO3_FAST = 5; the key of the run-time kernel table packed them with radix 4, so (fp64, ENO2_FAST = 4) collided with
    (fp32, ENO2 = 0) and a later fp32 launch reused the cached DOUBLE kernel with a float argument block.  One registered Hamiltonian, ENO2_FAST
    in fp64 first, then ENO2 in fp32, same grid and mode: each must equal the built-in system's result in its own precision."""
    from test_gpu_user_ham import DUBINS_REL_SRC, DUBINS_REL_COL
    n = (30, 28, 26)
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, 2)
    d0 = O.shape_cylinder(og, 2, None, .5) + 0.03 * np.random.default_rng(2).standard_normal(og.shape)
    reg = L.register_native_hamiltonian("dubins_rel_rt_keys", 3, DUBINS_REL_SRC, nparams=4, column_src=DUBINS_REL_COL, ncol=2)
    out = {}
    for dtype, sid in (("float64", 4), ("float32", _ffi.SCHEME_IDS["ENO2"]), ("float64", _ffi.SCHEME_IDS["ENO2"]), ("float32", 4)):
        for ham in ("user", "builtin"):
            dg = DeviceGrid(g, dtype)
            dg.bind_stream()
            y = torch.as_tensor(d0, device="cuda", dtype=getattr(torch, dtype))
            o = torch.empty_like(y)
            hid = reg.ham_id if ham == "user" else _ffi.HAM_DUBINS_REL
            _ffi.check(dg.lib.hj_rk_substep(dg.ctx, sid, hid, _ffi.darr(PAR), 0., _ffi.STAGE_EULER, 2e-3, 0, dg.ptr(y), None, dg.ptr(o), 3, 0, n[0]))
            dg.sync()
            assert bool(torch.isfinite(o).all()), (dtype, sid, ham)
            out[(dtype, sid, ham)] = o
    for dtype, sid in (("float64", 4), ("float32", _ffi.SCHEME_IDS["ENO2"]), ("float64", _ffi.SCHEME_IDS["ENO2"]), ("float32", 4)):
        a, b = out[(dtype, sid, "user")], out[(dtype, sid, "builtin")]
        tol = 1e-11 if dtype == "float64" else 2e-5
        d = (a - b).abs()
        assert float((d > tol).double().mean()) <= 2e-3 and float(d.max()) <= 1e-2, (dtype, sid, float(d.max()))


def test_thin_grids_take_the_transposed_march_by_default(monkeypatch):
    """The auto rule (hj_api.hip xp_wanted: thin 3-D ranges without neighbours at the pair kernel's sizes; hj_inst.hip launch_scheme: the plan model,
    then -- on a live context -- runs of both forms timed against each other): hj_rk_integrate ends up on the transposed launch BY ITSELF on a grid
    where it is ~20 % faster, gives the bits of the axis-0 march (HJ_XP=0) while it is still trying both, and HJ_XP_TRIALS=0 (the model alone) takes
    the transposed launch from the first call."""
    n = (20, 640, 520)
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, 2)
    x0 = torch.as_tensor(np.asarray(g.vs[0]).ravel(), device="cuda").reshape(-1, 1, 1)
    x1 = torch.as_tensor(np.asarray(g.vs[1]).ravel(), device="cuda").reshape(1, -1, 1)
    x2 = torch.as_tensor(np.asarray(g.vs[2]).ravel(), device="cuda").reshape(1, 1, -1)
    gen = torch.Generator(device="cuda").manual_seed(4)
    y = ((x0 * x0 + x1 * x1).sqrt() - 0.5 + 0.05 * torch.sin(3 * x2) + 0.01 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)).contiguous()
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    res = {}
    nsteps = 10         # (2 x 2 runs of 6 substeps are the trial; the 9th step runs the form that won)
    for tag, env in (("default", {}), ("model", dict(HJ_XP_TRIALS="0")), ("axis0", dict(HJ_XP="0"))):
        dg = _ctx(g, monkeypatch, **env)
        a, b, w = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
        tout, steps, where = C.c_double(), C.c_int64(), C.c_int()
        _ffi.check(dg.lib.hj_rk_integrate(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, _ffi.darr(PAR), 0., 1., 0.8, 1e300, 0, dg.ptr(y), dg.ptr(a), dg.ptr(b),
                                          dg.ptr(w), nsteps if tag != "model" else 1, -1., C.byref(tout), C.byref(steps), C.byref(where)))
        dg.sync()
        assert dg.lib.hj_last_kernel(dg.ctx) == (XP_NAME if tag != "axis0" else b"fused_pair_kernel"), (tag, dg.lib.hj_last_kernel(dg.ctx))
        if tag == "model":
            continue
        res[tag] = (tout.value, steps.value, (a if where.value == 1 else b).clone())
    assert res["default"][:2] == res["axis0"][:2] and res["default"][1] == nsteps
    assert torch.equal(res["default"][2], res["axis0"][2]), float((res["default"][2] - res["axis0"][2]).abs().max())


def test_bound_ring_reset_keeps_the_live_slots(monkeypatch):
    """The ring of CFL-bound keys is zeroed once per 2048 launches.  A step whose later-stage bounds are read together (hj_rk_step of a Hamiltonian
    whose alpha reads the costate range: two kept bounds per RK3 step) may straddle the reset: the entry of stage 2 used to be zeroed while stage 3
    was being enqueued ("bound slot holds no reduction").  Run across the reset at BOTH alignments; the run that straddles it must give the bits of
    the one that does not."""
    n = (26, 24, 22)
    g, og = mk([-1., -1., -1.], [1., 1., 1.], n, None)
    import levelsetpy_amd as L
    reg = L.register_native_hamiltonian("ring_reset_probe", 3, """
        H = par[0] * x[0] * p[1] + 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        alpha[0] = fmax(fabs(dmin[0]), fabs(dmax[0]));
        alpha[1] = fmax(fabs(dmin[1]), fabs(dmax[1])) + fabs(par[0] * x[0]);
        alpha[2] = fmax(fabs(dmin[2]), fabs(dmax[2]));
    """, nparams=1)
    xs = np.meshgrid(*[np.asarray(v).ravel() for v in g.vs], indexing="ij")
    y0 = torch.as_tensor(np.sqrt(sum(x * x for x in xs)) - 0.5, device="cuda").contiguous()
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    outs = []
    for shift in (0, 1):
        dg = _ctx(g, monkeypatch)
        cur, nxt, w0, w1 = y0.clone(), torch.empty_like(y0), torch.empty_like(y0), torch.empty_like(y0)
        for _ in range(shift):           # one more ring entry before the run: the other alignment
            _sub(dg, sid, _ffi.STAGE_EULER, 1e-3, cur, None, nxt, 0, n[0], slot=5)
        tout, dtout = C.c_double(), C.c_double()
        t = 0.
        for _ in range(1100):            # 2 entries per step: the reset falls inside this run
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, reg.ham_id, _ffi.darr([0.7]), t, 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1),
                                         C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur
            t = float(tout.value)
        sb, nb = (C.c_double * 4)(), C.c_int()
        _ffi.check(dg.lib.hj_rk_last_bounds(dg.ctx, sb, C.byref(nb)))
        assert nb.value == 3 and all(0 < sb[k] < 1 for k in range(3))
        outs.append((t, cur.clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("n", [(20, 640, 520), (64, 60, 66)])
def test_rk_steps_captured_into_a_graph(monkeypatch, n):
    """hj_rk_step enqueued while the stream is being captured (torch.cuda.graph -> hipGraph): the launch-form trial of thin grids (events, queries) and
    the tile-shape rotation stand still during a capture and the launches are plain kernel nodes; replays give the bits of the eager steps."""
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, 2)
    x0 = torch.as_tensor(np.asarray(g.vs[0]).ravel(), device="cuda").reshape(-1, 1, 1)
    x1 = torch.as_tensor(np.asarray(g.vs[1]).ravel(), device="cuda").reshape(1, -1, 1)
    y = ((x0 * x0 + x1 * x1).sqrt() - 0.5).expand(*n).contiguous()
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    dg = _ctx(g, monkeypatch)
    bufs = [y.clone(), torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)]
    tout, dtout = C.c_double(), C.c_double()

    def steps(k):
        cur, nxt = 0, 1
        for _ in range(k):
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, _ffi.darr(PAR), 0., 1e9, 0.8, 1e300, 0, dg.ptr(bufs[cur]), dg.ptr(bufs[nxt]),
                                         dg.ptr(bufs[2]), dg.ptr(bufs[3]), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur
        return cur
    last = steps(2)                       # eager: the static bound, the first trial runs
    dg.sync()
    ref2 = bufs[last].clone()
    bufs[0].copy_(y)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dg.bind_stream()
        with torch.cuda.graph(graph, stream=side):
            last_g = steps(2)
    dg.bind_stream()
    torch.cuda.synchronize()
    for _ in range(3):
        bufs[0].copy_(y)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(bufs[last_g], ref2)
    # and the context goes on eagerly afterwards
    bufs[0].copy_(y)
    last = steps(2)
    dg.sync()
    assert torch.equal(bufs[last], ref2)
