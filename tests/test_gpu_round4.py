"""GPU tests added in round 4 ("next" items 2 and 3):

  * FULL-ARRAY parity with the oracle at BASELINE sizes -- 51^3 (C1) five RK3 steps with every scheme, 201^3 (C2) one
    RK3 step through odeCFL3 (the pair kernel's instantiations), 4096^2 (C3) one ENO3 term evaluation; ENO2 / ENO3
    comparisons are bit for bit (the fused ENO substep is evaluated in the reference's operation order);
  * the stand-alone artificialDissipationGLF golden of the reference against the DEVICE function;
  * BASELINE C5 as decomposed: the double-pendulum 4-D grid, fp32, all axes periodic, split into a ring of axis-0 slabs
    -- deep-halo and per-substep schedules with 2 / 3 / 8 virtual ranks (one of them at the full 129^4 over 8), and the
    native RCCL stepper as a one-rank self ring -- all bitwise against the undivided grid.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd import _ffi  # noqa: E402
from levelsetpy_amd.context import DeviceGrid, device_grid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

from test_gpu_parity import mk, sdata, DERIV, SCHEMES, close, dubins  # noqa: E402
from test_gpu_configs import pendulum_grid  # noqa: E402

PAR_PENDULUM = [1.0, 0.0, 0.0, 0.0]


# ------------------------------------------------------------------------------ oracle parity at BASELINE sizes
@pytest.mark.parametrize("scheme", SCHEMES)
def test_c1_51_cubed_full_array_vs_oracle_five_rk3_steps(scheme):
    """BASELINE C1 (configs[0]): the whole 51^3 state after five odeCFL3 steps against the oracle, not five scalars.
    ENO2 / ENO3: bit for bit; the two WENO5 arithmetics: 1e-11 absolute."""
    g, og = dubins(51)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    assert np.array_equal(d0, O.shape_cylinder(og, 2, None, .5))
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    osys = O.DubinsRel(og, 1, 1)
    term = lambda tt, yy: O.term_lax_friedrichs(og, osys, scheme, tt, yy)  # noqa: E731
    y, t = d0.reshape(-1, 1), 0.
    yo, to = d0.reshape(-1, 1), 0.
    for _ in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        to, yo = O.ode_cfl_3(term, [to, 10.], yo, 0.8, single_step=True)
    assert abs(t - to) <= 1e-14
    if scheme.startswith("ENO"):
        assert t == to
        assert np.array_equal(y, yo), "%s: %d cells differ, max %.3e" % (scheme, int((y != yo).sum()), float(np.abs(y - yo).max()))
    else:
        close(y, yo, 1e-11, what=scheme)


def test_c2_201_cubed_one_rk3_step_vs_oracle_through_odecfl3():
    """BASELINE C2 at full size: one odeCFL3 step of the 201^3 Dubins problem through the drop-in API (the launches
    are the pair kernel's EULER / RK3_HALF / RK3_FULL instantiations, asserted) against the oracle's full array."""
    n = 201
    g, og = dubins(n)
    d0 = O.shape_cylinder(og, 2, None, .5)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y0 = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y0, op, sd)
    dg = device_grid(g)
    assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel"
    osys = O.DubinsRel(og, 1, 1)
    to, yo = O.ode_cfl_3(lambda tt, yy: O.term_lax_friedrichs(og, osys, "WENO5_ASSHIPPED", tt, yy), [0., 10.],
                         d0.reshape(-1, 1), 0.8, single_step=True)
    assert abs(float(t) - to) <= 1e-15
    close(y.cpu().numpy(), yo, 1e-11, what="201^3 RK3 step")
    assert torch.equal(y0, torch.as_tensor(d0.reshape(-1, 1), device="cuda"))      # the input is never written


def test_c3_4096_squared_eno3_term_vs_oracle_bitwise():
    """BASELINE C3 at full size: termLaxFriedrichs (ENO3 + GLF, double integrator) on the 4096^2 grid against the
    oracle -- array_equal, stepBound equal (the NumPy-ordered ENO arithmetic is claimed bit-faithful: here on
    16.8 M cells with the pair kernel), then one Euler substep of the state."""
    n = 4096
    g, og = mk([-1, -1], [1, 1], [n, n], None)
    d0 = O.shape_sphere(og, None, .25)
    y0 = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    sd = sdata(g, L.DoubleIntegrator(g, 1), L.upwindFirstENO3)
    yd, sb, _ = L.termLaxFriedrichs(0., y0, sd)
    dg = device_grid(g)
    assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel"
    yo, sbo = O.term_lax_friedrichs(og, O.DoubleIntegrator(og, 1), "ENO3", 0., d0.reshape(-1, 1))
    assert sb == sbo
    got = yd.cpu().numpy()
    assert np.array_equal(got, yo), "%d cells differ, max %.3e" % (int((got != yo).sum()), float(np.abs(got - yo).max()))
    # one odeCFL1 step (y + dt*ydot, ode_cfl_1.py / ode_cfl_3.py:151) of the whole state
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t1, y1, _ = L.odeCFL1(L.termLaxFriedrichs, [0., 10.], y0, op, sd)
    dt = 0.8 * sbo
    assert float(t1) == dt
    assert np.array_equal(y1.cpu().numpy(), d0.reshape(-1, 1) + dt * yo)


def test_glf_dissipation_reference_golden_on_device(golden):
    """The reference's stand-alone artificialDissipationGLF output (tests/golden/term.npz: glf_diss, glf_sb for the
    reference's own derivL / derivR arrays) against the DEVICE function fed device tensors (round 3 only checked the
    oracle against it), and against the NumPy-in / NumPy-out form of the same call."""
    G = golden("term.npz")
    n = tuple(int(v) for v in np.asarray(G["dub_N"]).ravel())
    g, og = mk(np.asarray(G["dub_min"]).ravel(), np.asarray(G["dub_max"]).ravel(), n, 2)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstENO3)
    dl = [torch.as_tensor(G["glf_dL%d" % i], device="cuda") for i in range(3)]
    dr = [torch.as_tensor(G["glf_dR%d" % i], device="cuda") for i in range(3)]
    data = torch.as_tensor(G["dub_data"], device="cuda")
    diss, sb = L.artificialDissipationGLF(0., data, dl, dr, sd)
    assert type(diss).__module__.startswith("torch") and diss.is_cuda
    close(diss.cpu().numpy().reshape(G["glf_diss"].shape), G["glf_diss"], 1e-12, what="glf diss (device)")
    assert abs(sb - float(G["glf_sb"])) <= 1e-14 * sb
    diss_np, sb_np = L.artificialDissipationGLF(0., G["dub_data"], [G["glf_dL%d" % i] for i in range(3)],
                                                [G["glf_dR%d" % i] for i in range(3)], sd)
    assert isinstance(diss_np, np.ndarray)
    close(diss_np.reshape(G["glf_diss"].shape), G["glf_diss"], 1e-12, what="glf diss (NumPy)")
    assert abs(sb_np - float(G["glf_sb"])) <= 1e-14 * sb


# ------------------------------------------------------------------------------ C5 as decomposed: 4-D fp32 periodic ring
def sphere4(g, noise=0.0, seed=0):
    """4-D sphere r = .5 (+ optional smooth and random perturbation) in fp32 on the device."""
    xs = [torch.as_tensor(np.asarray(v).ravel(), device="cuda", dtype=torch.float64) for v in g.vs]
    r2 = (xs[0] ** 2).reshape(-1, 1, 1, 1) + (xs[1] ** 2).reshape(1, -1, 1, 1) + (xs[2] ** 2).reshape(1, 1, -1, 1) \
        + (xs[3] ** 2).reshape(1, 1, 1, -1)
    d = r2.sqrt() - 0.5
    if noise:
        gen = torch.Generator(device="cuda").manual_seed(seed)
        d = d + 0.1 * torch.sin(xs[0]).reshape(-1, 1, 1, 1) * torch.cos(xs[2]).reshape(1, 1, -1, 1) \
            + noise * torch.randn(d.shape, generator=gen, device="cuda", dtype=torch.float64)
    return d.to(torch.float32).contiguous()


def undivided_steps(g, full, scheme, order, steps, dt):
    dg = DeviceGrid(g, "float32")
    dg.bind_stream()
    sid = _ffi.SCHEME_IDS[scheme]
    cur, nxt, w0, w1 = full.clone(), torch.empty_like(full), torch.empty_like(full), torch.empty_like(full)
    tout, dtout = C.c_double(), C.c_double()
    t = 0.
    for _ in range(steps):
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, sid, _ffi.HAM_DOUBLE_PENDULUM, _ffi.darr(PAR_PENDULUM), t, 1e9, 0.8, dt, 0,
                                     dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
        cur, nxt = nxt, cur
        t = float(tout.value)
        assert dtout.value == dt
    torch.cuda.synchronize()
    return t, cur


@pytest.mark.parametrize("scheme,world,n,order", [("WENO5_ASSHIPPED", 2, (40, 9, 10, 11), 3), ("WENO5_ASSHIPPED", 3, (57, 8, 9, 12), 3),
                                                  ("ENO3", 2, (38, 7, 12, 9), 2), ("WENO5_ASSHIPPED", 8, (150, 6, 7, 8), 3)])
def test_c5_ring_deep_halo_virtual_ranks_bitwise(scheme, world, n, order):
    """hj_slab_rk_step_deep on the 4-D all-periodic pendulum grid in fp32: `world` virtual ranks in one process, the ring
    closed rank world-1 <-> 0 (two ranks: both neighbours are the same rank), pad planes (3*order deep, with their own
    sin/cos tables of the axis-0 node: hj_ctx_set_axis0_pad) moved by the test.  Bitwise equal to the undivided grid."""
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    g, _ = pendulum_grid(n, low_mem=True)
    full = sphere4(g, noise=0.01, seed=world)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    sid = _ffi.SCHEME_IDS[scheme]
    steppers = []
    for r in range(world):
        slab = SlabDecomposition(n[0], world, r, True)
        assert slab.lo == (r - 1) % world and slab.hi == (r + 1) % world
        steppers.append(NativeSlabStepper(g, slab, sid, _ffi.HAM_DOUBLE_PENDULUM, PAR_PENDULUM, dxs, "float32", order=order,
                                          deep=True, external=lambda st: None))
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
        assert all(a == ts[0] for a in ts)
        t, dt = ts[0]
    t_ref, ref = undivided_steps(g, full, scheme, order, 3, dt)
    assert abs(t_ref - t) <= 1e-15
    for st in steppers:
        got, want = st.state(), ref[st.slab.begin:st.slab.end]
        assert torch.equal(got, want), "rank %d differs by %g" % (st.slab.rank, float((got - want).abs().max()))
        st.close()


class ThreadRing(object):
    """`world` in-process ranks (threads, one stream each) for SlabIntegrator + HipSlabBackend: 3-plane halo exchange per
    substep and the all-reduce through shared tensors and a barrier (the transport of test_gpu_parity._LocalTransport,
    for any world size and the closed ring)."""

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.box = {}

    def exchanger(self, slab):
        tr = self

        class Ex(object):
            def start(self, buf):
                n = slab.n_local
                torch.cuda.current_stream().synchronize()
                tr.box[(slab.rank, "low")] = buf[3:6].clone()
                tr.box[(slab.rank, "high")] = buf[n:n + 3].clone()
                torch.cuda.current_stream().synchronize()
                tr.bar.wait()
                if slab.hi is not None:
                    buf[n + 3:n + 6].copy_(tr.box[(slab.hi, "low")])
                if slab.lo is not None:
                    buf[0:3].copy_(tr.box[(slab.lo, "high")])
                torch.cuda.current_stream().synchronize()
                tr.bar.wait()
                return []

            @staticmethod
            def finish(reqs):
                pass

            def exchange(self, buf):
                self.start(buf)
        return Ex()

    def allreduce_max(self, rank):
        tr = self

        def f(t):
            torch.cuda.current_stream().synchronize()
            tr.box[(rank, "ar")] = t.clone()
            torch.cuda.current_stream().synchronize()
            tr.bar.wait()
            m = tr.box[(0, "ar")]
            for r in range(1, tr.world):
                m = torch.maximum(m, tr.box[(r, "ar")])
            torch.cuda.current_stream().synchronize()
            tr.bar.wait()
            t.copy_(m)
        return f


def run_ring_per_substep(g, n, world, scheme, full, steps):
    from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    tr = ThreadRing(world)
    out, errs = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                slab = SlabDecomposition(n[0], world, rank, True, self_exchange=True)
                be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS[scheme], _ffi.HAM_DOUBLE_PENDULUM, PAR_PENDULUM, "float32")
                integ = SlabIntegrator(slab, be, dxs, 3, 0.8, exchanger=tr.exchanger(slab), allreduce_max=tr.allreduce_max(rank))
                integ.set_state(full[slab.begin:slab.end])
                t = 0.
                for _ in range(steps):
                    t, dt = integ.step(t)
                be.sync()
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
    assert len(out) == world
    return out


@pytest.mark.parametrize("scheme,world,n", [("WENO5_ASSHIPPED", 2, (14, 9, 10, 11)), ("WENO5_ASSHIPPED", 3, (19, 8, 9, 12)),
                                            ("ENO3", 2, (12, 7, 12, 9)), ("WENO5_ASSHIPPED", 8, (43, 6, 7, 40)),
                                            ("WENO5_ASSHIPPED", 1, (13, 6, 7, 9))])
def test_c5_ring_per_substep_virtual_ranks_bitwise(scheme, world, n):
    """SlabIntegrator + HipSlabBackend (edge planes, 3-plane exchange per substep, interior) on the 4-D all-periodic
    pendulum grid in fp32, ring closed across the last and the first rank (world 1: rank 0 <-> rank 0 through the
    transport); slabs as thin as 5 planes (43 = 8*5 + 3).  Bitwise equal to the undivided grid."""
    g, _ = pendulum_grid(n, low_mem=True)
    full = sphere4(g, noise=0.01, seed=7)
    out = run_ring_per_substep(g, n, world, scheme, full, 2)
    t, dt = out[0][2], out[0][3]
    t_ref, ref = undivided_steps(g, full, scheme, 3, 2, dt)
    for r in range(world):
        b, e, tr_, dtr, ys = out[r]
        assert tr_ == t and dtr == dt and abs(t - t_ref) <= 1e-15
        assert torch.equal(ys, ref[b:e]), "rank %d differs by %g" % (r, float((ys - ref[b:e]).abs().max()))


def test_c5_129_to_the_4_over_eight_virtual_ranks_bitwise():
    """BASELINE C5 as BASELINE.json decomposes it: 129^4 fp32, all axes periodic, 8 ranks -> slabs of 17 / 16 planes
    (thinner than the 18 the deep-halo stepper needs: the per-substep schedule is the one that runs), ring closed 7 <-> 0.
    One RK3 step, bitwise against the undivided 129^4 grid; the pair kernel runs on the slabs."""
    from levelsetpy_amd.dist import SlabDecomposition
    n, world = (129, 129, 129, 129), 8
    counts = SlabDecomposition(129, world, 0, True).counts
    assert counts == [17] + [16] * 7 and min(counts) < 2 * 3 * 3
    g, _ = pendulum_grid(n, low_mem=True)
    full = sphere4(g, noise=0.0)
    out = run_ring_per_substep(g, n, world, "WENO5_ASSHIPPED", full, 1)
    t, dt = out[0][2], out[0][3]
    t_ref, ref = undivided_steps(g, full, "WENO5_ASSHIPPED", 3, 1, dt)
    assert abs(t - t_ref) <= 1e-15
    for r in range(world):
        b, e, _t, _dt, ys = out[r]
        assert torch.equal(ys, ref[b:e]), "rank %d differs by %g" % (r, float((ys - ref[b:e]).abs().max()))


def test_c5_native_rccl_stepper_self_ring_4d_fp32():
    """hj_slab_rk_step (ncclSend/ncclRecv inside the C library) on the 4-D fp32 pendulum grid as a one-rank ring: axis 0
    closed through a self send/recv instead of the in-kernel wrap, every per-substep schedule + the deep-halo one."""
    import torch.distributed as dist
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29593")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        n = (24, 7, 9, 10)
        g, _ = pendulum_grid(n, low_mem=True)
        full = sphere4(g, noise=0.01, seed=11)
        dxs = [float(v) for v in np.asarray(g.dx).ravel()]
        ref = None
        for deep, sched in ((True, None), (False, "overlap"), (False, "serial"), (False, "gated")):
            if sched is None:
                os.environ.pop("HJ_SLAB_SCHEDULE", None)
            else:
                os.environ["HJ_SLAB_SCHEDULE"] = sched
            try:
                slab = SlabDecomposition(n[0], 1, 0, True, self_exchange=True)
                nat = NativeSlabStepper(g, slab, _ffi.WENO5_ASSHIPPED, _ffi.HAM_DOUBLE_PENDULUM, PAR_PENDULUM, dxs, "float32", deep=deep)
            finally:
                os.environ.pop("HJ_SLAB_SCHEDULE", None)
            assert nat.nranks == 1
            nat.set_state(full)
            t = 0.
            for _ in range(3):
                t, dt = nat.step(t)
            got = nat.state().clone()
            torch.cuda.synchronize()
            nat.close()
            if ref is None:
                t_ref, ref = undivided_steps(g, full, "WENO5_ASSHIPPED", 3, 3, dt)
            assert abs(t - t_ref) <= 1e-15
            assert torch.equal(got, ref), "deep=%s %s differs by %g" % (deep, sched, float((got - ref).abs().max()))
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------------------ NumPy callers stay in HBM (lazy.HostView)
def test_numpy_loop_stays_on_the_device_and_equals_the_tensor_loop():
    """The reference's driver loop (NumPy in, NumPy out, every result fed back: hji_solver.py:542, Notes/rcbrt.ipynb cell 4)
    through odeCFL3 + termRestrictUpdate: results are HostViews consumed on the device by the next call; np.asarray() of
    them equals the tensor-in loop bit for bit, inputs are never mutated, writes through a view detach it."""
    from levelsetpy_amd.lazy import HostView
    g, og = dubins([33, 31, 29])
    d0 = O.shape_cylinder(og, 2, None, .5) + 0.02 * np.random.default_rng(1).standard_normal(og.shape)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5)
    sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sd, positive=0))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y_np = d0.flatten()
    keep = y_np.copy()
    y_t = torch.as_tensor(y_np, device="cuda")
    t1 = t2 = 0.
    seen = []
    for k in range(4):
        t1, y_np, _ = L.odeCFL3(L.termRestrictUpdate, [t1, 10.], y_np, op, sdr)
        t2, y_t, _ = L.odeCFL3(L.termRestrictUpdate, [t2, 10.], y_t, op, sdr)
        assert isinstance(y_np, HostView) and y_np.device_tensor() is not None and y_np.shape == (og.shape[0] * og.shape[1] * og.shape[2],)
        seen.append(y_np)
        if k == 1:
            looked = np.asarray(y_np)           # looking does not break the chain
            assert looked.shape == y_np.shape and not looked.flags.writeable
    assert t1 == t2 and np.array_equal(keep, d0.flatten())
    assert seen[0]._h is None and seen[2]._h is None and seen[3]._h is None      # nobody looked: nothing crossed PCIe
    assert np.array_equal(np.asarray(y_np), y_t.cpu().numpy())
    assert np.array_equal(np.asarray(seen[1]), looked)                  # earlier results stay valid (nothing is recycled)
    vr = y_np.reshape(og.shape)
    assert isinstance(vr, HostView) and vr.shape == og.shape and float(vr[3, 4, 5]) == float(y_t.reshape(og.shape)[3, 4, 5])
    # termLaxFriedrichs alone: (N,1) column in, HostView column out, arithmetic gives ndarrays
    yd, sb, _ = L.termLaxFriedrichs(0., d0.reshape(-1, 1), sd)
    ydt, sbt, _ = L.termLaxFriedrichs(0., torch.as_tensor(d0.reshape(-1, 1), device="cuda"), sd)
    assert isinstance(yd, HostView) and yd.shape == (d0.size, 1) and sb == sbt
    e = d0.reshape(-1, 1) + 0.5 * sb * yd
    assert isinstance(e, np.ndarray) and np.array_equal(e, d0.reshape(-1, 1) + 0.5 * sb * ydt.cpu().numpy())
    # a write through the view detaches it; the next call uploads the modified values
    y_mod = seen[-1]
    y_mod[0] = 123.0
    assert y_mod.device_tensor() is None
    t3, y3, _ = L.odeCFL3(L.termRestrictUpdate, [0., 10.], y_mod, op, sdr)
    ref_in = y_t.clone()
    ref_in[0] = 123.0
    t4, y4, _ = L.odeCFL3(L.termRestrictUpdate, [0., 10.], ref_in, op, sdr)
    assert np.array_equal(np.asarray(y3), y4.cpu().numpy())


def test_numpy_callers_get_plain_ndarrays_with_lazy_off():
    from levelsetpy_amd import lazy
    g, og = dubins([17, 15, 13])
    d0 = O.shape_cylinder(og, 2, None, .5)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstENO3)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t1, y1, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], d0.reshape(-1, 1), op, sd)
    lazy.set_lazy(False)
    try:
        t2, y2, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], d0.reshape(-1, 1), op, sd)
    finally:
        lazy.set_lazy(True)
    assert isinstance(y2, np.ndarray) and not isinstance(y1, np.ndarray) and np.array_equal(y1, y2) and t1 == t2


def test_term_convection_tensor_velocity_forms():
    """termConvection with torch velocity components that are not full arrays: a 0-dim tensor is a scalar
    speed, a broadcast-shaped tensor (N0,1,1) is expanded like its NumPy twin."""
    g, og = dubins((15, 14, 12))
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * np.random.default_rng(3).standard_normal(g.shape)
    prof = np.sin(2 * np.asarray(og.vs[0]).ravel()).reshape(-1, 1, 1) + 0.3
    vel_np = [0.7, prof, -0.4 * np.ones(g.shape)]
    vel_t = [torch.tensor(0.7, dtype=torch.float64, device="cuda"), torch.as_tensor(prof, device="cuda"),
             torch.as_tensor(vel_np[2], device="cuda")]
    y = torch.as_tensor(data.reshape(-1, 1), device="cuda")
    ref, sb0, _ = L.termConvection(0., y, L.Bundle(dict(grid=g, velocity=vel_np, derivFunc=L.upwindFirstENO3)))
    got, sb1, _ = L.termConvection(0., y, L.Bundle(dict(grid=g, velocity=vel_t, derivFunc=L.upwindFirstENO3)))
    assert sb0 == sb1 and torch.equal(ref, got)
    yo, sbo = O.term_convection(og, [0.7, np.broadcast_to(prof, g.shape), vel_np[2]], "ENO3", 0., data.reshape(-1, 1))
    close(got.cpu().numpy(), yo)
    assert abs(sb1 - sbo) <= 1e-14 * sbo


@pytest.mark.parametrize("n,scheme", [((65, 120, 110), "WENO5_ASSHIPPED"), ((23, 40, 300), "ENO3"), ((9, 64, 64), "WENO5")])
def test_gated_slab_schedule_self_ring_bitwise(n, scheme):
    """HJ_SLAB_SCHEDULE=gated (round 4): edge chunks and interior in ONE launch, the RCCL exchange gated on the counter the
    edge workgroups publish (hipStreamWaitValue64).  One-rank periodic ring through a real self send/recv: several tiles
    per plane, several interior chunks, slabs down to 9 planes; five RK3 steps bitwise equal to the in-kernel wrap."""
    import torch.distributed as dist
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29594")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        g, og = mk([-2., -1.25, -np.pi], [2. * (1 - 2 / n[0]), 1.25, np.pi * (1 - 2 / n[2])], n, (0, 2))
        x0 = torch.as_tensor(np.asarray(g.vs[0]).ravel(), device="cuda").reshape(-1, 1, 1)
        x1 = torch.as_tensor(np.asarray(g.vs[1]).ravel(), device="cuda").reshape(1, -1, 1)
        x2 = torch.as_tensor(np.asarray(g.vs[2]).ravel(), device="cuda").reshape(1, 1, -1)
        gen = torch.Generator(device="cuda").manual_seed(9)
        full = ((x0 * x0 + x1 * x1).sqrt() - 0.5 + 0.1 * torch.sin(3 * x0) * torch.cos(2 * x2)
                + 0.01 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)).contiguous()
        dxs = [float(v) for v in np.asarray(g.dx).ravel()]
        sid = _ffi.SCHEME_IDS[scheme]
        os.environ["HJ_SLAB_SCHEDULE"] = "gated"
        try:
            slab = SlabDecomposition(n[0], 1, 0, True, self_exchange=True)
            nat = NativeSlabStepper(g, slab, sid, _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.], dxs, deep=False)
        finally:
            os.environ.pop("HJ_SLAB_SCHEDULE", None)
        nat.set_state(full)
        t = 0.
        for _ in range(5):
            t, dt = nat.step(t)
        got = nat.state().clone()
        torch.cuda.synchronize()
        nat.close()
        dg = DeviceGrid(g)
        dg.bind_stream()
        cur, nxt, w0, w1 = full.clone(), torch.empty_like(full), torch.empty_like(full), torch.empty_like(full)
        tout, dtout = C.c_double(), C.c_double()
        tr = 0.
        for _ in range(5):
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, _ffi.darr([1., 1., 1., 2.]), tr, 1e9, 0.8, dt, 0,
                                         dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur
            tr = float(tout.value)
        torch.cuda.synchronize()
        assert abs(t - tr) <= 1e-15
        if scheme == "WENO5":
            assert float((got - cur).abs().max()) <= 1e-12        # the all-reduced epsilon is reduced in another order
        else:
            assert torch.equal(got, cur), float((got - cur).abs().max())
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((37, 21, 19), 2), ((64, 40, 130), 2), ((45, 300), None)])
def test_paired_chunk_directions_bitwise(scheme, n, pd, monkeypatch):
    """HJ_PAIR_DIRS=1: the chunks of a tile column march pairwise in opposite directions (the axis-0 queue then holds the
    planes in march order and the stencil is handed the reversed view).  A cell's value must not depend on the direction:
    three RK3 steps bitwise equal to the all-upward march and to the direct kernel, every scheme, odd chunk counts, 2-D.
    (The default build compiles the down-marching support out -- csrc/hj_fused.h, HJ_MAYDOWN: measured slower than doing
    without -- and ignores the knob; the test then compares three all-upward runs.  It bites in -DHJ_MAYDOWN=1 builds.)"""
    nd = len(n)
    if nd == 3:
        g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
        ham, par = _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.]
        data = O.shape_cylinder(og, 2, None, .5)
    else:
        g, og = mk([-1, -1], [1, 1], n, pd)
        ham, par = _ffi.HAM_DOUBLE_INTEGRATOR, [1., 0., 0., 0.]
        data = O.shape_sphere(og, None, .3)
    y0 = torch.as_tensor(data + 0.02 * np.random.default_rng(17).standard_normal(og.shape), device="cuda")
    sid = _ffi.SCHEME_IDS[scheme]
    outs = {}
    for tag, env in (("up", {"HJ_PAIR": "2", "HJ_PAIR_DIRS": "0", "HJ_MIN_CHUNK": "4"}), ("paired", {"HJ_PAIR": "2", "HJ_PAIR_DIRS": "1", "HJ_MIN_CHUNK": "4"}),
                     ("direct", {"HJ_FORCE_DIRECT": "1"})):
        for k in ("HJ_PAIR", "HJ_PAIR_DIRS", "HJ_FORCE_DIRECT", "HJ_MIN_CHUNK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        cur, nxt, w0, w1 = y0.clone(), torch.empty_like(y0), torch.empty_like(y0), torch.empty_like(y0)
        tout, dtout = C.c_double(), C.c_double()
        t = 0.
        for _ in range(3):
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, ham, _ffi.darr(par), t, 1e9, 0.8, 1e300, 0, dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0),
                                         dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt = nxt, cur
            t = float(tout.value)
        dg.sync()
        if tag != "direct":
            assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel"
            ext = (C.c_int * 4)()
            _ffi.check(dg.lib.hj_last_tile(dg.ctx, ext))
            assert ext[0] < n[0]                       # several chunks per tile column: there ARE pairs
        outs[tag] = cur.clone()
    if scheme == "WENO5":
        # (the intended WENO5's epsilon comes from a reduction whose order does not depend on the march either)
        pass
    assert torch.equal(outs["up"], outs["paired"]), float((outs["up"] - outs["paired"]).abs().max())
    assert torch.equal(outs["paired"], outs["direct"]), float((outs["paired"] - outs["direct"]).abs().max())


def test_grid_of_a_billion_cells_tiled_equals_direct():
    """1025^3 fp64 (1.08e9 cells, 8.6 GB per array -- a grid for the 288 GB of this card): one odeCFL3 step through the tiled
    kernels against the direct kernel, bitwise (tools/big_grid_check.py in its own process: ~55 GB peak)."""
    import subprocess
    import sys
    free, _total = torch.cuda.mem_get_info()
    if free < 80e9:
        pytest.skip("needs 80 GB of free device memory")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "big_grid_check.py"), "1025"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    assert "bitwise: True" in p.stdout


@pytest.mark.parametrize("scheme", SCHEMES)
def test_sixty_rk3_steps_at_41_cubed_vs_oracle(scheme):
    """A long horizon: 60 odeCFL3 steps of the 41^3 Dubins problem (until the front has crossed a quarter of the grid) against the
    oracle.  ENO2 / ENO3 stay the reference's bit for bit over all 60 steps (every stencil choice equal); the WENO5 arithmetics
    within 1e-10 (rounding differences of the contracted forms accumulate linearly)."""
    g, og = dubins(41)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    osys = O.DubinsRel(og, 1, 1)
    term = lambda tt, yy: O.term_lax_friedrichs(og, osys, scheme, tt, yy)  # noqa: E731
    y, t = torch.as_tensor(d0.reshape(-1, 1), device="cuda"), 0.
    yo, to = d0.reshape(-1, 1), 0.
    for _ in range(60):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        to, yo = O.ode_cfl_3(term, [to, 10.], yo, 0.8, single_step=True)
    got = y.cpu().numpy()
    assert float(np.abs(yo - d0.reshape(-1, 1)).max()) > 0.1          # the state did move
    if scheme.startswith("ENO"):
        assert t == to
        assert np.array_equal(got, yo), "%s: %d cells differ, max %.3e" % (scheme, int((got != yo).sum()), float(np.abs(got - yo).max()))
    else:
        assert abs(t - to) <= 1e-12
        close(got, yo, 1e-10, what=scheme)


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3"])
def test_eighty_rk2_steps_double_integrator_restricted_vs_oracle_bitwise(scheme):
    """The 2-D stencil path (C3's kernels at 97 x 143) over a long horizon through termRestrictUpdate and odeCFL2, as the air3D-style
    drivers call it ((N,) vectors): 80 steps bit for bit the oracle's (= the reference's operation order)."""
    n = (97, 143)
    g, og = mk([-1., -1.5], [1., 1.5], n, None)
    d0 = L.shapeSphere(g, np.zeros((2, 1)), .45)
    sd = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sdata(g, L.DoubleIntegrator(g, 1.25), DERIV[scheme]), positive=0))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.7, singleStep='on')))
    inner = lambda tt, yy: O.term_lax_friedrichs(og, O.DoubleIntegrator(og, 1.25), scheme, tt, yy)  # noqa: E731
    term = O.term_restrict_update(inner, positive=False)
    y, t = torch.as_tensor(d0.reshape(-1), device="cuda"), 0.
    yo, to = d0.reshape(-1), 0.
    for _ in range(80):
        t, y, _ = L.odeCFL2(L.termRestrictUpdate, [t, 10.], y, op, sd)
        to, yo = O.ode_cfl_2(term, [to, 10.], yo, 0.7, single_step=True)
    got = y.cpu().numpy()
    assert float(np.abs(yo - d0.reshape(-1)).max()) > 0.05
    assert t == to
    assert np.array_equal(got, yo.reshape(got.shape)), "%d cells differ, max %.3e" % (int((got != yo.reshape(got.shape)).sum()), float(np.abs(got - yo.reshape(got.shape)).max()))


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((37, 21, 19), 2), ((20, 70, 45), (0, 2)), ((45, 300), None), ((64, 33), 1), ((9, 8, 7), None)])
def test_terms_through_the_tiled_kernel_equal_the_direct_kernel_bitwise(scheme, n, pd, monkeypatch):
    """Round 4: on large fp64 grids termNormal / termReinit / termConvection run through the tiled substep kernel (TermOp,
    csrc/hj_termop.h: LDS-staged stencils, register queue along axis 0) instead of the one-thread-per-cell term_kernel.  Both
    call the same cell function: ydot and the step bound must be equal to the last bit -- every scheme, both sub-cell orders,
    scalar and array coefficients, periodic and extrapolated axes, extents below the tile, chunks of a few planes."""
    nd = len(n)
    lo, hi = [-1.0] * nd, [1.0] * nd
    rng = np.random.default_rng(23)
    res = {}
    for tag, env in (("tiled", "0"), ("direct", "-1")):
        monkeypatch.setenv("HJ_TERM_TILED_FROM", env)
        monkeypatch.setenv("HJ_MIN_CHUNK", "4")
        g, og = mk(lo, hi, n, pd)
        if tag == "tiled":
            phi = O.shape_sphere(og, None, .45) * (1.0 + 0.4 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[1])) + 0.02 * rng.standard_normal(n)
            speed = 0.5 + 0.3 * np.cos(og.xs[0]) * np.ones(n)
            vels = [0.7 * np.ones(n) * np.sin(2 * og.xs[nd - 1]), -0.4 + 0.5 * np.sin(3 * og.xs[1]) * np.ones(n), -0.2][:nd]
        y = torch.as_tensor(phi.reshape(-1, 1), device="cuda")
        tt = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
        cases = [("normal array", L.termNormal, dict(speed=tt(speed))), ("normal scalar", L.termNormal, dict(speed=-1.25)),
                 ("reinit 0", L.termReinit, dict(initial=tt(phi), subcell_fix_order=0)), ("reinit 1", L.termReinit, dict(initial=tt(phi), subcell_fix_order=1)),
                 ("convection arrays", L.termConvection, dict(velocity=[tt(v) if isinstance(v, np.ndarray) else v for v in vels])),
                 ("convection scalars", L.termConvection, dict(velocity=[0.3, -0.6, 0.2][:nd]))]
        out = {}
        for name, fn, extra in cases:
            a, sb, _ = fn(0., y, L.Bundle(dict(grid=g, derivFunc=DERIV[scheme], **extra)))
            dg = device_grid(g, "float64")
            out[name] = (a.clone(), sb, dg.lib.hj_last_kernel(dg.ctx))
        res[tag] = out
    for name in res["tiled"]:
        a, sba, ka = res["tiled"][name]
        b, sbb, kb = res["direct"][name]
        assert ka == b"fused_substep_kernel" and kb == b"term_kernel", (ka, kb)
        assert bool(torch.isfinite(a).all())
        assert torch.equal(a, b), "%s: %d cells differ, max %.3e" % (name, int((a != b).sum()), float((a - b).abs().max()))
        assert sba == sbb, (name, sba, sbb)


def test_bench_line_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the keys the driver reads: metric / value / unit /
    n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config{workload} + roofline
    {bound, achieved, peak, unit, frac, traffic} + cpu_baseline{value, unit, cores, kind, sample} + parity (ok).  (The extra
    workloads and the rocprofv3 child passes are switched off here: they are exercised by every default bench run.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--repeats", "5",
                        "--no-also", "--no-live-traffic"], capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"].startswith("grid-cell RK-substep updates/sec, Dubins-3D HJI 201^3 fp64") and d["unit"] == "cell-substeps/s"
    assert (d["n_gpus"], d["steps"], d["warmup"], d["higher_is_better"], d["scaling"], d["dtype"], d["data"]) == (1, 5, 2, True, "weak", "f64", "synthetic")
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 5e10 and abs(d["value"] - 8120601 * 3 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "infinity-cache/fabric") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.2 < r["frac"] < 1.0 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "cell-substeps/s" and c["value"] > 1e5 and c["sample"]
    assert d["parity"]["ok"] is True and d["parity"]["max_abs_diff"] <= d["parity"]["tol"] == 1e-11


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd,dtype", [((23, 18, 31), 2, "float64"), ((40, 77), None, "float64"), ((9, 8, 10, 11), (0, 3), "float32"),
                                        ((33, 20, 17), (0, 2), "float32")])
def test_all_dimension_derivatives_in_one_launch_equal_the_per_dimension_kernel(scheme, n, pd, dtype):
    """hj_lf_split_begin (round 4: ONE launch for every dimension, the stencils of a cell gathered once) against hj_upwind
    (one launch per dimension): derivL / derivR bitwise equal, and the min / max the launch reduces equal to the arrays' own."""
    from levelsetpy_amd.spatial import upwind_all_dims, cached_minmax
    nd = len(n)
    g, og = mk([-1.0] * nd, [1.0] * nd, n, pd)
    rng = np.random.default_rng(31)
    data = (O.shape_sphere(og, None, .4) + 0.05 * rng.standard_normal(n)).astype(dtype)
    y = torch.as_tensor(data, device="cuda")
    both = upwind_all_dims(DERIV[scheme], g, y)
    assert both is not None
    dL, dR = both
    for d in range(nd):
        l1, r1 = DERIV[scheme](g, y, d)
        assert torch.equal(dL[d], l1) and torch.equal(dR[d], r1), "dim %d: %g / %g" % (d, float((dL[d] - l1).abs().max()), float((dR[d] - r1).abs().max()))
        mm = cached_minmax(dL[d], dR[d])
        assert mm is not None
        assert mm[0] == min(float(dL[d].min()), float(dR[d].min())) and mm[1] == max(float(dL[d].max()), float(dR[d].max()))
