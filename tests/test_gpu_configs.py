"""GPU tests at the BASELINE.json workloads the round-1 suite did not run:

  C4  Dubins relative 3-D, 513^3 fp64 -- single domain AND slab-decomposed over 8 (virtual) ranks
  C5  double pendulum 4-D, 129^4 fp32, all four axes periodic

Both are far beyond what the NumPy oracle finishes in a test, so they are checked through
size-independent properties (tiled kernel = independent direct kernel, plane-range split bitwise,
closed-form CFL bound, decomposed = undivided bitwise), and the fp32 4-D instantiations are compared
with the fp64 oracle on small odd shapes (1e-4 relative; the reference has no fp32 path).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd import _ffi  # noqa: E402
from levelsetpy_amd.context import DeviceGrid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

from test_gpu_parity import mk, sdata, DERIV, SCHEMES, _substep, dubins  # noqa: E402

PAR_DUBINS = [1., 1., 1., 2.]


def dubins_lowmem(n):
    gmin = np.array([[-.75, -1.25, -np.pi]]).T
    gmax = np.array([[3.25, 1.25, np.pi * (1 - 2 / n)]]).T
    return L.createGrid(gmin, gmax, n * np.ones((3, 1), dtype=np.int64), 2, low_mem=True)


def cylinder_on_device(g, noise=0.0, seed=0):
    """shapeCylinder(g, 2, 0, .5) built on the GPU from grid.vs (cylinder.py:55-59): sqrt(x0^2 + x1^2) - r,
    constant along axis 2; optional smooth + random perturbation so that no two stencil values tie."""
    x0 = torch.as_tensor(np.asarray(g.vs[0]).ravel(), device="cuda").reshape(-1, 1, 1)
    x1 = torch.as_tensor(np.asarray(g.vs[1]).ravel(), device="cuda").reshape(1, -1, 1)
    x2 = torch.as_tensor(np.asarray(g.vs[2]).ravel(), device="cuda").reshape(1, 1, -1)
    d = (x0 * x0 + x1 * x1).sqrt() - 0.5 + 0 * x2
    if noise:
        gen = torch.Generator(device="cuda").manual_seed(seed)
        d = d + 0.1 * torch.sin(3 * x0) * torch.cos(2 * x2) + noise * torch.randn(d.shape, generator=gen, device="cuda",
                                                                                  dtype=torch.float64)
    return d.contiguous()


# ------------------------------------------------------------------------------ C4: 513^3
@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED"])
def test_c4_513_cubed_single_domain_properties(scheme, monkeypatch):
    """One RK3 step of the 513^3 Dubins problem: tiled = direct kernel to rounding, the last substep
    computed as three plane ranges equals one launch bitwise, stepBound equals its closed form."""
    n = 513
    g = dubins_lowmem(n)
    d0 = cylinder_on_device(g)
    outs = {}
    for force in ("0", "1"):
        monkeypatch.setenv("HJ_FORCE_DIRECT", force)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        a, b, c = dg.empty(), dg.empty(), dg.empty()
        _substep(dg, scheme, _ffi.HAM_DUBINS_REL, PAR_DUBINS, _ffi.STAGE_EULER, 8e-4, d0, None, a)
        _substep(dg, scheme, _ffi.HAM_DUBINS_REL, PAR_DUBINS, _ffi.STAGE_RK3_HALF, 8e-4, a, d0, b)
        _substep(dg, scheme, _ffi.HAM_DUBINS_REL, PAR_DUBINS, _ffi.STAGE_RK3_FULL, 8e-4, b, d0, c)
        if force == "0":
            c2 = torch.zeros_like(c)
            for k, (p0, p1) in enumerate([(0, 65), (65, 449), (449, n)]):     # a rank-0 slab, the middle, a last slab
                _substep(dg, scheme, _ffi.HAM_DUBINS_REL, PAR_DUBINS, _ffi.STAGE_RK3_FULL, 8e-4, b, d0, c2, p0, p1, slot=4 + k)
            dg.sync()
            assert torch.equal(c, c2), float((c - c2).abs().max())
            sb, am = C.c_double(), (C.c_double * 4)()
            _ffi.check(dg.lib.hj_read_step_bound(dg.ctx, 3, C.byref(sb), am))
            x0, x1, x2 = (np.asarray(v).ravel() for v in g.vs)
            a0 = np.max(np.abs(1 - np.cos(x2))) + np.max(np.abs(x1))
            a1 = np.max(np.abs(np.sin(x2))) + np.max(np.abs(x0))
            dx = np.asarray(g.dx).ravel()
            assert abs(sb.value - 1 / (a0 / dx[0] + a1 / dx[1] + 2 / dx[2])) <= 1e-13 * sb.value
        dg.sync()
        outs[force] = c
        del a, b
    # round 3: the direct kernel uses the tiled kernels' per-cell functions AND stage expressions: bit for bit
    assert torch.equal(outs["0"], outs["1"]), float((outs["0"] - outs["1"]).abs().max())
    assert bool(torch.isfinite(outs["0"]).all())
    # mirror symmetry (x2, x3) -> (-x2, -x3) of the Dubins problem survives the step (see the 201^3 test)
    u = outs["0"]
    k = torch.arange(n, device="cuda")
    mirror = u.flip(1)[:, :, (n - k) % n]
    assert float((u - mirror).abs().max()) <= 1e-10


@pytest.mark.parametrize("scheme,order", [("WENO5_ASSHIPPED", 3), ("ENO3", 2)])
def test_c4_513_cubed_eight_virtual_ranks_deep_halo_bitwise(scheme, order):
    """BASELINE C4 as decomposed: 513 = 8*64 + 1 planes over 8 ranks (65, 64, ..., 64), deep-halo stepper
    (one exchange of 3*order planes per step) with the pad planes moved by the test
    (hj_comm_init_external).  Three steps must equal the undivided 513^3 grid BITWISE."""
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    n, world = 513, 8
    g = dubins_lowmem(n)
    full = cylinder_on_device(g, noise=0.01, seed=3)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    sid = _ffi.SCHEME_IDS[scheme]
    steppers = []
    for r in range(world):
        slab = SlabDecomposition(n, world, r, False)
        steppers.append(NativeSlabStepper(g, slab, sid, _ffi.HAM_DUBINS_REL, PAR_DUBINS, dxs, order=order, deep=True,
                                          external=lambda st: None))
    assert [st.n for st in steppers] == [65] + [64] * 7
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

    for st in steppers:
        st.set_state(full[st.slab.begin:st.slab.end])
    move_pads()
    dg = DeviceGrid(g)
    dg.bind_stream()
    cur, nxt, w0, w1 = full, torch.empty_like(full), torch.empty_like(full), torch.empty_like(full)
    tout, dtout = C.c_double(), C.c_double()
    t = t_ref = 0.
    for _ in range(3):
        ts = [st.step(t) for st in steppers]
        move_pads()
        assert all(a[0] == ts[0][0] for a in ts)
        t, dt = ts[0]
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, sid, _ffi.HAM_DUBINS_REL, _ffi.darr(PAR_DUBINS), t_ref, 1e9, 0.8, dt, 0,
                                     dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
        cur, nxt = nxt, cur
        t_ref = float(tout.value)
        assert dtout.value == dt and abs(t_ref - t) <= 1e-15
    torch.cuda.synchronize()
    for st in steppers:
        got, ref = st.state(), cur[st.slab.begin:st.slab.end]
        assert torch.equal(got, ref), "rank %d differs by %g" % (st.slab.rank, float((got - ref).abs().max()))
        st.close()


# ------------------------------------------------------------------------------ C5: 129^4 fp32, all periodic
def pendulum_grid(n, pd=(0, 1, 2, 3), low_mem=False):
    n = [int(v) for v in (n if np.ndim(n) else [n] * 4)]
    gmin = [-np.pi, -8., -np.pi, -8.]
    gmax = [np.pi * (1 - 2 / n[0]), 8 * (1 - 2 / n[1]), np.pi * (1 - 2 / n[2]), 8 * (1 - 2 / n[3])]
    if low_mem:
        g = L.createGrid(np.array(gmin).reshape(-1, 1), np.array(gmax).reshape(-1, 1),
                         np.array(n, dtype=np.int64).reshape(-1, 1), list(pd) if pd else None, low_mem=True)
        return g, None
    return mk(gmin, gmax, n, pd)


def _pendulum_alpha_max(g, u):
    """max_x alpha_d of DoublePendulum4D from its definition (dynamics.py:140-159), in fp64 on the GPU with
    torch (test-side arithmetic only), one th1 node at a time so that 129^4 needs no 4-D temporaries."""
    th1, w1, th2, w2 = (torch.as_tensor(np.asarray(v).ravel(), device="cuda", dtype=torch.float64) for v in g.vs)
    W1, W2 = w1.reshape(-1, 1, 1), w2.reshape(1, 1, -1)
    s2, c2 = torch.sin(th2).reshape(1, -1, 1), torch.cos(th2).reshape(1, -1, 1)
    m1 = m3 = 0.0
    for a in th1.tolist():
        s1, c1 = float(np.sin(a)), float(np.cos(a))
        sd, cd = s2 * c1 - c2 * s1, c2 * c1 + s2 * s1
        den1 = 2.0 - cd * cd
        f1 = (W1 * W1 * sd * cd + 9.8 * s2 * cd + W2 * W2 * sd - 2 * 9.8 * s1) / den1
        f3 = (-W2 * W2 * sd * cd + 2 * 9.8 * s1 * cd - 2 * W1 * W1 * sd - 2 * 9.8 * s2) / den1
        m1, m3 = max(m1, float(f1.abs().max())), max(m3, float(f3.abs().max()))
    return [float(w1.abs().max()), m1 + u, float(w2.abs().max()), m3 + u]


def test_c5_129_to_the_4_fp32_all_periodic_properties(monkeypatch):
    """BASELINE C5 at full size and precision (129^4 = 277 M cells, fp32, every axis periodic): the kernel the
    library picks (round 5: the compile-time-tile pair kernel, hj_fused4v.h), the one-cell-per-lane
    kernel (1024,1,3,2,2) and the independent direct kernel on an RK3 step -- bit for bit the same state; the last
    substep split into plane ranges bitwise, the CFL bound against the definition."""
    n = 129
    g, _ = pendulum_grid(n, low_mem=True)
    xs = [torch.as_tensor(np.asarray(v).ravel(), device="cuda", dtype=torch.float64) for v in g.vs]
    r2 = (xs[0] ** 2).reshape(-1, 1, 1, 1) + (xs[1] ** 2).reshape(1, -1, 1, 1) + (xs[2] ** 2).reshape(1, 1, -1, 1) \
        + (xs[3] ** 2).reshape(1, 1, 1, -1)
    d0 = (r2.sqrt() - 0.5).to(torch.float32).contiguous()       # 4-D sphere r = .5
    del r2
    par = [1.0, 0., 0., 0.]
    dt = 2e-4
    outs = {}
    for name, force, pair, kern in (("default", "0", None, b"fused_flat4_kernel"), ("single", "0", "0", b"fused_substep_kernel"),
                                    ("direct", "1", None, b"direct_substep_kernel")):
        monkeypatch.setenv("HJ_FORCE_DIRECT", force)
        if pair is None:
            monkeypatch.delenv("HJ_PAIR", raising=False)
        else:
            monkeypatch.setenv("HJ_PAIR", pair)
        dg = DeviceGrid(g, "float32")
        dg.bind_stream()
        a, b, c = dg.empty(), dg.empty(), dg.empty()
        _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_EULER, dt, d0, None, a)
        _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_RK3_HALF, dt, a, d0, b)
        _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_RK3_FULL, dt, b, d0, c)
        assert dg.lib.hj_last_kernel(dg.ctx) == kern, (name, dg.lib.hj_last_kernel(dg.ctx))
        if force == "0":
            c2 = torch.zeros_like(c)
            for k, (p0, p1) in enumerate([(0, 17), (17, 100), (100, n)]):
                _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_RK3_FULL, dt, b, d0, c2, p0, p1, slot=4 + k)
            dg.sync()
            assert torch.equal(c, c2), (name, float((c - c2).abs().max()))
            del c2
            sb, am = C.c_double(), (C.c_double * 4)()
            _ffi.check(dg.lib.hj_read_step_bound(dg.ctx, 3, C.byref(sb), am))
            ref = _pendulum_alpha_max(g, 1.0)
            for d in range(4):
                assert abs(am[d] - ref[d]) <= 2e-6 * ref[d], (name, d, am[d], ref[d])     # fp32 tables and arithmetic
        dg.sync()
        outs[name] = c
        del a, b, dg
    assert bool(torch.isfinite(outs["default"]).all())
    # every cell is a pure function of its inputs through the same per-cell functions and stage expressions
    assert torch.equal(outs["default"], outs["single"]), float((outs["default"] - outs["single"]).abs().max())
    assert torch.equal(outs["default"], outs["direct"]), float((outs["default"] - outs["direct"]).abs().max())
    # the update moved the state (a kernel that copies its input would pass everything above)
    assert float((outs["default"] - d0).abs().max()) > 1e-4


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((9, 11, 13, 7), (0, 1, 2, 3)), ((14, 13, 17, 15), (0, 2)), ((21, 7, 8, 33), None)])
def test_fp32_4d_tiled_and_direct_vs_fp64_oracle(scheme, n, pd, monkeypatch):
    """The fp32 4-D instantiations C5 runs (tiled (1024,1,3,2,2) and direct), every scheme, small odd shapes,
    against the fp64 oracle at 1e-4 relative.  ENO2/ENO3 choose stencils by comparing
    |D2|, |D3|: in fp32 a comparison whose margin is below fp32 rounding may go the other way, so for
    those two the bound holds on all but a small fraction of cells (masked comparison, fraction < 2e-3
    asserted) and every cell stays within the candidates' spread."""
    g, og = pendulum_grid(n, pd)
    rng = np.random.default_rng(21)
    data = O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2]) + 0.02 * rng.standard_normal(n)
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, 0., data.reshape(-1, 1))
    scale = float(np.max(np.abs(yo)))
    y32 = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=torch.float32)
    got = {}
    # "0": one cell per lane (1024,1,3,2,2); "1": direct; "pair": two cells per lane (256,2,6,2: 5 pair + 1 single halo slots per thread), built for the light stencils
    # "flat" (round 6): the full-row kernel (hj_flat4v.h) where the last axis has at least 8 cells, else the pair kernel again
    variants = ("0", "1", "pair", "flat") if scheme in ("WENO5_ASSHIPPED", "ENO2") else ("0", "1")
    for force in variants:
        monkeypatch.setenv("HJ_FORCE_DIRECT", "1" if force == "1" else "0")
        monkeypatch.setenv("HJ_PAIR", "2" if force in ("pair", "flat") else "0")
        monkeypatch.setenv("HJ_FLAT4", "2" if force == "flat" else "0")
        g.__dict__.pop("_hj_device", None)
        yd, sb, _ = L.termLaxFriedrichs(0., y32, sdata(g, L.DoublePendulum4D(g, 1.0), DERIV[scheme]))
        dg = g.__dict__["_hj_device"]
        dg = dg[next(iter(dg))] if isinstance(dg, dict) else dg
        assert dg.lib.hj_last_kernel(dg.ctx) == {"0": b"fused_substep_kernel", "1": b"direct_substep_kernel", "pair": b"fused_pair_kernel",
                                                 "flat": b"fused_flat4_kernel" if n[3] >= 8 else b"fused_pair_kernel"}[force]
        assert yd.dtype == torch.float32
        assert abs(sb - sbo) <= 1e-5 * sbo
        got[force] = yd.cpu().numpy().astype(np.float64)
    g.__dict__.pop("_hj_device", None)
    for force, yd in got.items():
        rel = np.abs(yd - yo) / scale
        if scheme.startswith("WENO"):
            assert rel.max() <= 1e-4, (force, rel.max())
        else:
            assert np.mean(rel > 1e-4) <= 2e-3, (force, float(np.mean(rel > 1e-4)))
            assert rel.max() <= 0.2, (force, rel.max())
    # the fp32 kernels share the per-cell arithmetic and (round 3) the stage expressions: bit for bit
    for force in variants[1:]:
        assert np.array_equal(got["0"], got[force]), (force, np.max(np.abs(got["0"] - got[force])))


@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "ENO3"])
def test_fp32_4d_rk3_steps_vs_fp64_oracle(scheme):
    """Five odeCFL3 steps of the all-periodic 4-D pendulum problem in fp32 (device tensors in, tensors out)
    against the fp64 oracle: 1e-4 relative on the state, t to fp32 rounding of stepBound."""
    n = (11, 9, 12, 10)
    g, og = pendulum_grid(n)
    data = O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2])
    sys_ = L.DoublePendulum4D(g, 1.0)
    sd = sdata(g, sys_, DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=torch.float32)
    term = lambda tt, yy: O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, tt, yy)  # noqa: E731
    yo, t, to = data.reshape(-1, 1), 0., 0.
    for _ in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        to, yo = O.ode_cfl_3(term, [to, 10.], yo, 0.8, single_step=True)
    assert y.dtype == torch.float32
    assert abs(t - to) <= 1e-5 * to
    rel = np.abs(y.cpu().numpy().astype(np.float64) - yo) / np.max(np.abs(yo))
    if scheme.startswith("WENO"):
        assert rel.max() <= 1e-4, rel.max()
    else:
        assert np.mean(rel > 1e-4) <= 5e-3 and rel.max() <= 0.05, (float(np.mean(rel > 1e-4)), rel.max())


# ------------------------------------------------------------------------------ kernel variants
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((33, 21, 19), 2), ((20, 45), None), ((9, 8, 10, 7), (0, 1, 2, 3))])
def test_plain_stage_kernels_bitwise_equal_generic(scheme, n, pd, monkeypatch):
    """The flag-free instantiations (MODE 1/2: plain Euler / convex-combination stages) against the
    runtime-flag kernel (HJ_NO_PLAIN=1) on one RK3 and one RK2 step: bitwise."""
    nd = len(n)
    if nd == 3:
        g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
        ham, par = _ffi.HAM_DUBINS_REL, PAR_DUBINS
        data = O.shape_cylinder(og, 2, None, .5)
    elif nd == 2:
        g, og = mk([-1, -1], [1, 1], n, pd)
        ham, par = _ffi.HAM_DOUBLE_INTEGRATOR, [1.5, 0, 0, 0]
        data = O.shape_sphere(og, None, .25)
    else:
        g, og = pendulum_grid(n, pd)
        ham, par = _ffi.HAM_DOUBLE_PENDULUM, [1.0, 0, 0, 0]
        data = O.shape_sphere(og, None, 1.5)
    data = data + 0.02 * np.random.default_rng(8).standard_normal(n)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("HJ_NO_PLAIN", flag)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        y = dg.to_device(data)
        outs = []
        for order in (3, 2):
            nxt, w0, w1 = dg.empty(), dg.empty(), dg.empty()
            tout, dtout = C.c_double(), C.c_double()
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, _ffi.SCHEME_IDS[scheme], ham, _ffi.darr(par), 0., 1e9, 0.8, 1e300, 0,
                                         dg.ptr(y), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            dg.sync()
            outs.append(nxt)
        res[flag] = outs
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert float((res["0"][0] - torch.as_tensor(data, device="cuda")).abs().max()) > 0


# ------------------------------------------------------------------------------ stage-fused kernel (RK stages 1+2)
def _stage12_case(n, pd, tz=None):
    nd = len(n)
    if nd == 3:
        g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
        ham, par = _ffi.HAM_DUBINS_REL, PAR_DUBINS
        data = O.shape_cylinder(og, 2, None, .5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[1])
    else:
        g, og = mk([-1, -1], [1, 1], n, pd)
        ham, par = _ffi.HAM_DOUBLE_INTEGRATOR, [1.5, 0, 0, 0]
        data = O.shape_sphere(og, None, .25) + 0.05 * np.sin(5 * og.xs[0] + 3 * og.xs[1])
    if tz:
        g.bdryData = [L.Bundle(dict(towardZero=True)) if d in tz else None for d in range(nd)]
    data = data + 0.02 * np.random.default_rng(4).standard_normal(n)
    return g, ham, par, data


@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "ENO3", "ENO2"])
@pytest.mark.parametrize("n,pd,tz", [
    ((40, 37, 29), 2, None),            # Dubins: axis 2 periodic, axes 0/1 extrapolated; several tiles per axis
    ((23, 90, 70), (0, 2), None),       # periodic march axis
    ((31, 45, 33), None, (1,)),         # all extrapolated, towardZero ghost data on axis 1
    ((9, 8, 200), (1,), None),          # thin grid, long rows, periodic axis 1 narrower than a ring wrap
    ((64, 700), None, None),            # 2-D: one tile per row
    ((30, 5000), (1,), None),           # 2-D: several tiles, periodic along the row
    ((50, 41), (0,), None),             # 2-D: periodic march axis
])
@pytest.mark.parametrize("pair", ["0", "2"])
def test_stage_fused_kernel_bitwise_equals_two_substeps(scheme, n, pd, tz, pair, monkeypatch):
    """hj_rk_stage12 (RK stages 1+2 in one launch, y1 kept on chip) against hj_rk_substep(EULER) followed by
    hj_rk_substep(RK3_HALF / RK2_FULL): BITWISE, for both coefficient pairs; tilings forced small so that
    interior tiles, edge tiles, shifted last tiles and several chunks all occur.  pair = "2": the two-cells-per-lane
    kernel of round 3 (hj_fused12v.h; HJ_F12_PAIR=2 fails instead of falling back), "0": the one-cell-per-lane one."""
    g, ham, par, data = _stage12_case(n, pd, tz)
    monkeypatch.setenv("HJ_F12_PAIR", pair)
    small = {"HJ_F12_R": "2", "HJ_TARGET_BLOCKS": "40"}
    if pair == "2":        # several tiles on both plane axes, shifted last tiles, an odd tile origin
        small = {"HJ_F12_E2": "12", "HJ_F12_E1": "10", "HJ_TARGET_BLOCKS": "40"}
    for knobs in ({}, small):
        for k in ("HJ_F12_R", "HJ_TARGET_BLOCKS", "HJ_F12_E2", "HJ_F12_E1"):
            monkeypatch.delenv(k, raising=False)
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        y = dg.to_device(data)
        for (ca, cb, st2) in ((0.75, 0.25, _ffi.STAGE_RK3_HALF), (0.5, 0.5, _ffi.STAGE_RK2_FULL)):
            y1, ref, got = dg.empty(), dg.empty(), torch.full(dg.shape, float("nan"), dtype=torch.float64, device="cuda")
            dt = 1.5e-3
            _substep(dg, scheme, ham, par, _ffi.STAGE_EULER, dt, y, None, y1)
            _substep(dg, scheme, ham, par, st2, dt, y1, y, ref, slot=4)
            _ffi.check(dg.lib.hj_rk_stage12(dg.ctx, _ffi.SCHEME_IDS[scheme], ham, _ffi.darr(par), dt, ca, cb,
                                            dg.ptr(y), dg.ptr(got), 5))
            dg.sync()
            assert dg.lib.hj_last_kernel(dg.ctx) == (b"fused12_pair_kernel" if pair == "2" else b"fused12_kernel")
            assert bool(torch.isfinite(got).all()), "cells left unwritten: %d" % int((~torch.isfinite(got)).sum())
            assert torch.equal(got, ref), "%s max diff %g at %s" % (knobs, float((got - ref).abs().max()),
                                                                  np.unravel_index(int((got - ref).abs().argmax()), n))
            sb1, sb2 = C.c_double(), C.c_double()
            _ffi.check(dg.lib.hj_read_step_bound(dg.ctx, 4, C.byref(sb1), None))
            _ffi.check(dg.lib.hj_read_step_bound(dg.ctx, 5, C.byref(sb2), None))
            assert sb1.value == sb2.value


@pytest.mark.parametrize("order", [2, 3])
def test_rk_step_with_stage_fusion_equals_unfused(order, monkeypatch):
    """hj_rk_step with HJ_FUSE12=1 against HJ_FUSE12=0 (three / two launches): bitwise, same t and dt; and
    hj_rk_plan reports the launch counts."""
    g, ham, par, data = _stage12_case((36, 50, 44), 2)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("HJ_FUSE12", flag)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        cur, nxt, w0, w1 = dg.to_device(data), dg.empty(), dg.empty(), dg.empty()
        tout, dtout = C.c_double(), C.c_double()
        nl, fused = C.c_int(), C.c_int()
        _ffi.check(dg.lib.hj_rk_plan(dg.ctx, order, _ffi.WENO5_ASSHIPPED, ham, _ffi.darr(par), 0, C.byref(nl), C.byref(fused)))
        assert (nl.value, fused.value) == ((order - 1, 1) if flag == "1" else (order, 0))
        t = 0.
        for _ in range(3):
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, _ffi.WENO5_ASSHIPPED, ham, _ffi.darr(par), t, 1e9, 0.8, 1e300, 0,
                                         dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            cur, nxt, t = nxt, cur, float(tout.value)
        dg.sync()
        res[flag] = (cur.clone(), t, dtout.value)
    assert res["0"][1:] == res["1"][1:]
    assert torch.equal(res["0"][0], res["1"][0]), float((res["0"][0] - res["1"][0]).abs().max())


# ------------------------------------------------------------------------------ split path on device kernels
class _ForeignDubins(object):
    """A user-side system the package knows nothing about (plain callables, torch arithmetic on whatever
    arrays it is handed): what every non-native system of the reference looks like to termLaxFriedrichs
    (DynamicalSystems/bird.py:277,346, flock.py:190,237)."""

    def __init__(self, g, v=1.0, w=1.0):
        self.g, self.v, self.w = g, v, w
        self.calls = 0
        # trigonometry once, in NumPy, so that the NumPy and the tensor runs see the same bits
        self.tab = {"x1": np.asarray(g.xs[0]), "x2": np.asarray(g.xs[1]),
                    "c3": np.cos(np.asarray(g.xs[2])), "s3": np.sin(np.asarray(g.xs[2]))}
        self.tab = {k: np.ascontiguousarray(np.broadcast_to(a, g.shape)) for k, a in self.tab.items()}
        self.ttab = {}

    def _t(self, name, like):
        if not torch.is_tensor(like):
            return self.tab[name]
        key = (name, like.device)
        if key not in self.ttab:
            self.ttab[key] = torch.as_tensor(self.tab[name], device=like.device)
        return self.ttab[key]

    def ham(self, t, data, p, sd):
        self.calls += 1
        x1, x2, c3, s3 = (self._t(k, p[0]) for k in ("x1", "x2", "c3", "s3"))
        return (p[0] * (self.v - self.v * c3) - p[1] * (self.v * s3)
                - self.w * abs(p[0] * x2 - p[1] * x1 - p[2]) + self.w * abs(p[2]))

    def part(self, t, data, dmin, dmax, sd, dim):
        x1, x2, c3, s3 = (self._t(k, data) for k in ("x1", "x2", "c3", "s3"))
        if dim == 0:
            return abs(self.v - self.v * c3) + abs(self.w * x2)
        if dim == 1:
            return abs(self.v * s3) + abs(self.w * x1)
        return 2 * self.w                      # a scalar alpha: used as is (artificial_diss_glf.py:101-104)


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED", "WENO5"])
def test_split_path_device_kernels_vs_reference_arithmetic(scheme):
    """Foreign hamFunc / partialFunc on device tensors: hj_lf_split_begin (all dims, one sync) -> callbacks ->
    hj_lf_split_end (diss, -(ham - diss), max alpha in one kernel).  Against (1) the same pipeline on NumPy
    arrays (per-dimension hj_upwind + the reference's array expressions): bit-identical ydot and stepBound;
    (2) the oracle; (3) the fused native path."""
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / 17)], (19, 18, 17), 2)
    data = O.shape_cylinder(og, 2, None, .5) + 0.02 * np.random.default_rng(3).standard_normal(og.shape)
    fs = _ForeignDubins(g)
    sd = L.Bundle(dict(grid=g, hamFunc=fs.ham, partialFunc=fs.part, dissFunc=L.artificialDissipationGLF,
                       CoStateCalc=DERIV[scheme]))
    y = data.reshape(-1, 1)
    yd_np, sb_np, _ = L.termLaxFriedrichs(0., y, sd)
    yt = torch.as_tensor(y, device="cuda")
    yd_t, sb_t, _ = L.termLaxFriedrichs(0., yt, sd)
    assert torch.is_tensor(yd_t) and yd_t.shape == yt.shape
    assert sb_t == sb_np
    assert np.array_equal(yd_t.cpu().numpy(), yd_np)
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), scheme, 0., y)
    assert abs(sb_t - sbo) <= 1e-13 * sbo
    assert np.max(np.abs(yd_np - yo)) <= 1e-11 * max(1.0, np.max(np.abs(yo)))
    fused, sbf, _ = L.termLaxFriedrichs(0., yt, sdata(g, L.DubinsVehicleRel(g, 1, 1), DERIV[scheme]))
    assert abs(sbf - sb_t) <= 1e-13 * sbf
    assert float((fused - yd_t).abs().max()) <= 1e-11 * max(1.0, float(fused.abs().max()))
    # the dissipation function on its own (ham = None): diss and stepBound, tensors in -> tensor out
    dL, dR = zip(*[DERIV[scheme](g, torch.as_tensor(data, device="cuda"), i) for i in range(3)])
    diss_t, sb2 = L.artificialDissipationGLF(0., torch.as_tensor(data, device="cuda"), list(dL), list(dR), sd)
    dLn, dRn = [a.cpu().numpy() for a in dL], [a.cpu().numpy() for a in dR]
    diss_n, sb3 = L.artificialDissipationGLF(0., data, dLn, dRn, sd)
    assert sb2 == sb3 == sb_t
    assert np.array_equal(diss_t.cpu().numpy(), np.asarray(diss_n))


@pytest.mark.parametrize("order", [1, 2, 3])
def test_generic_integrator_stage_kernels_bitwise(order):
    """odeCFLn with a foreign schemeFunc (the generic loop): on device tensors every stage expression is one
    hj_rk_combine launch; the result must equal the NumPy-array run of the same loop bit for bit, and the
    native device integrator to rounding."""
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / 15)], (17, 16, 15), 2)
    data = O.shape_cylinder(og, 2, None, .5) + 0.02 * np.random.default_rng(5).standard_normal(og.shape)
    fs = _ForeignDubins(g)
    sd = L.Bundle(dict(grid=g, hamFunc=fs.ham, partialFunc=fs.part, dissFunc=L.artificialDissipationGLF,
                       CoStateCalc=L.upwindFirstWENO5))
    ode = {1: L.odeCFL1, 2: L.odeCFL2, 3: L.odeCFL3}[order]
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='off')))
    y0 = data.reshape(-1, 1)
    t_np, y_np, _ = ode(L.termLaxFriedrichs, [0., 0.02], y0, op, sd)
    t_t, y_t, _ = ode(L.termLaxFriedrichs, [0., 0.02], torch.as_tensor(y0, device="cuda"), op, sd)
    assert t_np == t_t and torch.is_tensor(y_t)
    assert np.array_equal(y_t.cpu().numpy(), y_np)
    sdn = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5)
    t_n, y_n, _ = ode(L.termLaxFriedrichs, [0., 0.02], y0, op, sdn)
    assert abs(t_n - t_np) <= 1e-14
    assert np.max(np.abs(y_n - y_np)) <= 1e-11
    # the kernel against the reference's expressions, every mode, odd length
    dg = DeviceGrid(g, "float64")
    rng = np.random.default_rng(9)
    x0, yy, zz = (rng.standard_normal(4097) for _ in range(3))
    dt = 0.0123
    want = {1: yy + dt * zz, 2: 0.25 * (3 * x0 + (yy + dt * zz)), 3: (1 / 3) * (x0 + 2 * (yy + dt * zz)),
            4: 0.5 * (x0 + (yy + dt * zz))}
    tx, ty, tz = (torch.as_tensor(a, device="cuda") for a in (x0, yy, zz))
    for mode, ref in want.items():
        out = torch.empty_like(tx)
        _ffi.check(dg.lib.hj_rk_combine(dg.ctx, mode, dt, dg.ptr(tx), dg.ptr(ty), dg.ptr(tz), dg.ptr(out), out.numel()))
        assert np.array_equal(out.cpu().numpy(), ref), mode


def test_eno3a_helper_returns_reference_dd_bundle(golden):
    """upwindFirstENO3aHelper: candidates AND the divided-difference Bundle (stripped tables D1 N+1, D2 N+2,
    D3 N+3 entries along dim: ENO3aHelper.py:99-112,190) against the reference's goldens; approx4's fourth
    element equals the second up to rounding (ENO3aHelper.py:28-32)."""
    G = golden("deriv.npz")
    for tag, nd in (("g2", 2), ("g3", 3)):
        pd = 1 if tag == "g2" else 2
        n = G[tag + "_data"].shape
        g, og = mk(G[tag + "_min"], G[tag + "_max"], n, pd)
        data = G[tag + "_data"]
        for dim in range(nd):
            dL, dR, DD = L.upwindFirstENO3aHelper(g, data, dim, True, False)
            assert len(dL) == 4 and len(dR) == 4
            for k in range(3):
                assert np.max(np.abs(dL[k] - G["%s_helper_dL%d_d%d" % (tag, k, dim)])) <= 1e-11
                assert np.max(np.abs(dR[k] - G["%s_helper_dR%d_d%d" % (tag, k, dim)])) <= 1e-11
            assert np.max(np.abs(dL[3] - dL[1])) <= 1e-10 and np.max(np.abs(dR[3] - dR[1])) <= 1e-10
            for name in ("D1", "D2", "D3"):
                ref = G["%s_helper_%s_d%d" % (tag, name, dim)]
                got = getattr(DD, name)
                assert got.shape == ref.shape
                assert np.max(np.abs(got - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))


# ------------------------------------------------------------------------------ more terms
def _noisy_circle(n=(41, 37)):
    g, og = mk([-1, -1.1], [1, 1.1], n, None)
    rng = np.random.default_rng(11)
    phi = O.shape_sphere(og, None, .5) * (1.0 + 0.4 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[1])) + 0.01 * rng.standard_normal(n)
    return g, og, phi


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_term_normal_vs_oracle_and_unit_speed_growth(scheme):
    """termNormal on NumPy arrays and on device tensors against oracle.term_normal (speed as scalar, array and
    callable), and its behaviour: a circle moving along its normal at unit speed stays a signed distance, phi(t) =
    phi(0) - t, away from the centre kink and the domain edge."""
    g, og, phi = _noisy_circle()
    speed = 0.5 + 0.3 * np.cos(og.xs[0])
    for sp in (1.5, speed, lambda t, data, sd: speed):
        sd = L.Bundle(dict(grid=g, derivFunc=DERIV[scheme], speed=sp))
        yo, sbo = O.term_normal(og, sp if not callable(sp) else speed, scheme, 0., phi.reshape(-1, 1))
        for y in (phi.reshape(-1, 1), torch.as_tensor(phi.reshape(-1, 1), device="cuda")):
            yd, sb, _ = L.termNormal(0., y, sd)
            ydn = yd.cpu().numpy() if torch.is_tensor(yd) else yd
            assert ydn.shape == yo.shape
            assert np.max(np.abs(ydn - yo)) <= 1e-11 * max(1.0, np.max(np.abs(yo)))
            assert abs(sb - sbo) <= 1e-12 * sbo
    g2, og2 = mk([-1, -1], [1, 1], (81, 81), None)
    sdf = O.shape_sphere(og2, None, .3)
    sd = L.Bundle(dict(grid=g2, derivFunc=DERIV[scheme], speed=1.0))
    t, y, _ = L.odeCFL3(L.termNormal, [0., 0.2], torch.as_tensor(sdf.reshape(-1, 1), device="cuda"),
                        L.odeCFLset(factorCFL=.5), sd)
    got = y.cpu().numpy().reshape(81, 81)
    r = np.sqrt(og2.xs[0] ** 2 + og2.xs[1] ** 2)
    band = (r > 0.35) & (r < 0.85)
    assert abs(t - 0.2) <= 1e-12
    assert np.max(np.abs(got[band] - (sdf[band] - 0.2))) <= 2e-3


@pytest.mark.parametrize("scheme", ["ENO2", "WENO5_ASSHIPPED"])
@pytest.mark.parametrize("order", [0, 1])
def test_term_reinit_vs_oracle_and_signed_distance(scheme, order):
    """termReinit (with and without the subcell fix) on NumPy arrays and device tensors against
    oracle.term_reinit, and its purpose: a distorted implicit function relaxes to a signed distance function
    (|grad phi| -> 1 near the interface) while the zero level set stays where it was."""
    g, og, phi = _noisy_circle()
    sd = L.Bundle(dict(grid=g, derivFunc=DERIV[scheme], initial=phi, subcell_fix_order=order))
    yo, sbo = O.term_reinit(og, phi, scheme, 0., phi.reshape(-1, 1), order)
    for y in (phi.reshape(-1, 1), torch.as_tensor(phi.reshape(-1, 1), device="cuda")):
        yd, sb, _ = L.termReinit(0., y, sd)
        ydn = yd.cpu().numpy() if torch.is_tensor(yd) else yd
        assert np.max(np.abs(ydn - yo)) <= 1e-10 * max(1.0, np.max(np.abs(yo)))
        assert abs(sb - sbo) <= 1e-12 * sbo
    g2, og2 = mk([-1, -1], [1, 1], (101, 101), None)
    true = O.shape_sphere(og2, None, .5)
    bad = true * (1.5 + np.sin(4 * og2.xs[0]) * np.cos(3 * og2.xs[1]))        # same zero level set, wrong slopes
    sd = L.Bundle(dict(grid=g2, derivFunc=DERIV[scheme], initial=torch.as_tensor(bad, device="cuda"), subcell_fix_order=order))
    t, y, _ = L.odeCFL3(L.termReinit, [0., 0.6], torch.as_tensor(bad.reshape(-1, 1), device="cuda"),
                        L.odeCFLset(factorCFL=.5), sd)
    got = y.cpu().numpy().reshape(101, 101)
    band = np.abs(true) < 0.2
    assert np.max(np.abs(got[band] - true[band])) <= (0.03 if order == 0 else 0.015), np.max(np.abs(got[band] - true[band]))


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5", "WENO5_ASSHIPPED"])
def test_fused_term_kernels_equal_the_array_path(scheme):
    """Round 3: termNormal / termReinit / termConvection with one of this package's derivative functions are ONE
    kernel launch each (hj_term_*, csrc/hj_terms.h).  The same terms with a FOREIGN derivFunc (a wrapper the package
    cannot recognise) take the derivatives from hj_upwind and run the array expressions of normal_reinit.py /
    convection.py.  Both evaluate the same expressions in the same order (contraction off in the kernel): equal to
    the last bit (termReinit: to 1 ulp) on a 3-D grid with periodic and extrapolated axes; step bounds equal to rounding."""
    g, og = dubins((19, 16, 14))
    rng = np.random.default_rng(5)
    phi = O.shape_cylinder(og, 2, None, .5) * (1.0 + 0.3 * np.sin(2 * og.xs[0])) + 0.02 * rng.standard_normal(g.shape)
    native = DERIV[scheme]
    foreign = lambda grid, data, dim: native(grid, data, dim)     # noqa: E731  (no _hj_scheme tag: array path)
    y = torch.as_tensor(phi.reshape(-1, 1), device="cuda")
    speed = torch.as_tensor(0.5 + 0.3 * np.cos(og.xs[0]) * np.ones(g.shape), device="cuda")
    vel = [0.7, torch.as_tensor(-0.4 + 0.5 * np.sin(3 * og.xs[1]) * np.ones(g.shape), device="cuda"), -0.2]
    cases = [
        (L.termNormal, dict(speed=speed)),
        (L.termNormal, dict(speed=-1.25)),
        (L.termReinit, dict(initial=torch.as_tensor(phi, device="cuda"), subcell_fix_order=0)),
        (L.termReinit, dict(initial=torch.as_tensor(phi, device="cuda"), subcell_fix_order=1)),
        (L.termConvection, dict(velocity=vel)),
    ]
    for fn, extra in cases:
        a, sba, _ = fn(0., y, L.Bundle(dict(grid=g, derivFunc=native, **extra)))
        b, sbb, _ = fn(0., y, L.Bundle(dict(grid=g, derivFunc=foreign, **extra)))
        assert torch.is_tensor(a) and a.is_cuda and a.shape == b.shape == y.shape
        # termNormal and termConvection: to the last bit; termReinit's quotient chain S*p/|p| is rounded differently by
        # torch's elementwise kernels in places (measured: 1 ulp, 2.8e-17 absolute)
        if fn is L.termReinit:
            assert float((a - b).abs().max()) <= 4e-16 * max(1.0, float(b.abs().max())), float((a - b).abs().max())
        else:
            assert torch.equal(a, b), "%s %s: max diff %g" % (fn.__name__, sorted(extra), float((a - b).abs().max()))
        assert abs(sba - sbb) <= 4e-16 * abs(sbb), (fn.__name__, sba, sbb)
        # NumPy in -> NumPy out through the same kernel
        c, sbc, _ = fn(0., phi.reshape(-1, 1), L.Bundle(dict(grid=g, derivFunc=native, **extra)))
        assert isinstance(c, np.ndarray) and np.array_equal(c, a.cpu().numpy()) and sbc == sba


class _DoubleIntegratorPlant(object):
    """dynSys protocol of computeOptTraj (compute_opt_traj.py:124-131) for xddot = u, |u| <= 1."""

    def __init__(self, x):
        self.x = np.asarray(x, dtype=np.float64)

    def get_opt_u(self, t, deriv, uMode, x):
        s = np.sign(deriv[1]) if deriv[1] != 0 else 1.0
        return -s if uMode == 'min' else s

    def update_state(self, u, dt, x, d=None):
        k = lambda z: np.array([z[1], u])                                   # noqa: E731  RK4 of (x1' = x2, x2' = u)
        k1 = k(x); k2 = k(x + .5 * dt * k1); k3 = k(x + .5 * dt * k2); k4 = k(x + dt * k3)
        self.x = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        return self.x


def test_compute_opt_traj_reaches_the_target_in_minimum_time():
    """HJIPDE_solve (native double integrator, minVOverTime, all times stored, flipped) -> computeOptTraj: the
    bang-bang trajectory from (0.5, 0) enters the target disc, and it does so no later than the closed-form
    minimum time to the origin, 2*sqrt(0.5) (double_integrator.py:91-119), plus one sampling interval."""
    n = 101
    g, og = mk([-1, -1], [1, 1], (n, n), None)
    data0 = L.shapeSphere(g, np.zeros((2, 1)), .1)
    sys_ = L.DoubleIntegrator(g, 1)
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=L.artificialDissipationGLF,
                       derivFunc=L.upwindFirstENO3))
    tau = np.linspace(0, 1.6, 33)
    data, tau2, _ = L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, flipOutput=True)))
    assert data.shape == (33, n, n)
    plant = _DoubleIntegratorPlant([0.5, 0.0])
    traj, ttau = L.computeOptTraj(g, data, tau, plant, L.Bundle(dict(uMode='min', subSamples=8)))
    assert traj.shape[0] == 2 and traj.shape[1] == ttau.shape[0] and traj.shape[1] >= 10
    end = traj[:, -1]
    assert np.hypot(end[0], end[1]) <= 0.1 + 2 * float(np.asarray(g.dx).max()), end
    assert ttau[-1] <= 2 * np.sqrt(0.5) + 0.05 + 1e-9, ttau[-1]
    assert np.all(np.isfinite(traj))
    # the switching structure: decelerate first (x2 goes negative), then brake
    assert traj[1].min() < -0.4


class _DubinsPursuerPlant(object):
    """A 3-D plant on the Dubins-relative grid (periodic heading): x' = (-v + v cos x3 + u x2, v sin x3 - u x1, -u),
    |u| <= 1, the control chosen from the costate as the reference's systems do (dubins_relative.py:83-88)."""

    def __init__(self, x):
        self.x = np.asarray(x, dtype=np.float64)

    def get_opt_u(self, t, deriv, uMode, x):
        det = deriv[0] * x[1] - deriv[1] * x[0] - deriv[2]
        s = 1.0 if det >= 0 else -1.0
        return -s if uMode == 'min' else s

    def get_opt_v(self, t, deriv, dMode, x):
        return 0.0

    def update_state(self, u, dt, x, d=None):
        f = lambda z: np.array([-1.0 + np.cos(z[2]) + u * z[1], np.sin(z[2]) - u * z[0], -u])    # noqa: E731
        k1 = f(x); k2 = f(x + .5 * dt * k1); k3 = f(x + .5 * dt * k2); k4 = f(x + dt * k3)
        self.x = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        return self.x


@pytest.mark.parametrize("case", ["double_integrator", "dubins"])
def test_compute_opt_traj_equals_oracle_point_by_point(case):
    """computeOptTraj (costates from the HIP upwind kernels, 2^dim corner values read per evaluation) against
    oracle.compute_opt_traj (NumPy restatement: pinned derivatives, eval_u as the reference's RegularGridInterpolator
    on the periodically augmented table) on the SAME stored value function: every trajectory point within 1e-9, the
    same time stamps.  The shipped reference raises: parity unpinned, this pins the product to
    the oracle."""
    if case == "double_integrator":
        n = 61
        g, og = mk([-1, -1], [1, 1], (n, n), None)
        data0 = L.shapeSphere(g, np.zeros((2, 1)), .1)
        sys_ = L.DoubleIntegrator(g, 1)
        tau = np.linspace(0, 1.2, 25)
        mk_plant = lambda: _DoubleIntegratorPlant([0.45, 0.05])        # noqa: E731
        deriv = L.upwindFirstENO3
    else:
        g, og = dubins((31, 31, 24))
        data0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
        sys_ = L.DubinsVehicleRel(g, 1, 1)
        tau = np.linspace(0, 0.6, 13)
        mk_plant = lambda: _DubinsPursuerPlant([1.1, 0.3, 2.9])        # noqa: E731   (heading near the periodic seam)
        deriv = L.upwindFirstWENO5
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=L.artificialDissipationGLF,
                       derivFunc=deriv))
    data, _, _ = L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, flipOutput=True)))
    data = np.asarray(data)
    pa, pb = mk_plant(), mk_plant()
    traj, ttau = L.computeOptTraj(g, data, tau, pa, L.Bundle(dict(uMode='min', subSamples=4)))
    otraj, otau = O.compute_opt_traj(og, data, tau, pb, 'min', None, 4)
    assert traj.shape == otraj.shape and traj.shape[1] >= 5, (traj.shape, otraj.shape)
    assert np.array_equal(ttau, otau)
    assert np.max(np.abs(traj - otraj)) <= 1e-9, np.max(np.abs(traj - otraj))
    # and on a device-resident table
    pc = mk_plant()
    traj2, _ = L.computeOptTraj(g, torch.as_tensor(data, device="cuda"), tau, pc, L.Bundle(dict(uMode='min', subSamples=4)))
    assert np.max(np.abs(traj2 - otraj)) <= 1e-9


def test_llf_scalar_alpha_and_fourth_candidate_vs_reference_golden(golden):
    """The product's artificialDissipationLLF against the reference's own output for the case the shipped
    function runs (a 0-d alpha for every dimension: tests/golden/make_golden.py:gen_extra), on NumPy arrays and
    on device tensors; and upwindFirstENO3aHelper's approx4 candidate against the reference."""
    G = golden("extra.npz")
    n = G["g3_data"].shape
    g, og = mk(G["g3_min"], G["g3_max"], n, 2)
    seen = []

    def partial(t, data, derivMin, derivMax, schemeData, dim):
        j = (dim + 1) % 3
        glob = max(abs(float(derivMin[j])), abs(float(derivMax[j])))
        loc = max(abs(float(derivMin[dim].min())), abs(float(derivMax[dim].max())))
        seen.append([int(np.ndim(derivMin[k]) if not torch.is_tensor(derivMin[k]) else derivMin[k].dim()) for k in range(3)])
        return 0.25 * (dim + 1) + 0.5 * glob + 0.125 * loc

    sd = L.Bundle(dict(grid=g, partialFunc=partial))
    for conv in (np.asarray, lambda a: torch.as_tensor(np.asarray(a), device="cuda")):
        dL = [conv(G["llf_dL%d" % i]) for i in range(3)]
        dR = [conv(G["llf_dR%d" % i]) for i in range(3)]
        del seen[:]
        diss, sb = L.artificialDissipationLLF(0., conv(G["g3_data"]), dL, dR, sd)
        diss = diss.cpu().numpy() if torch.is_tensor(diss) else np.asarray(diss)
        assert np.max(np.abs(diss - G["llf_diss"])) <= 1e-12 * max(1.0, np.max(np.abs(G["llf_diss"])))
        assert abs(sb - float(G["llf_sb"])) <= 1e-14 * sb
        assert seen == G["llf_range_ndims"].tolist()
    for dim in range(3):
        dL, dR, DD = L.upwindFirstENO3aHelper(g, G["g3_data"], dim, True)
        assert np.max(np.abs(dL[3] - G["g3_helper4_dL3_d%d" % dim])) <= 1e-11
        assert np.max(np.abs(dR[3] - G["g3_helper4_dR3_d%d" % dim])) <= 1e-11


def test_eno3_non_finite_selection_semantics():
    """Documented divergence: the reference forms all three ENO3 candidates and multiplies
    them by boolean masks (upwind_first_eno3a.py:133-141), so an Inf anywhere in a cell's 7-point footprint turns
    the cell into NaN even if the candidate holding it is not selected; the kernels select first and only then
    form the chosen candidate.  Away from the non-finite value both agree; inside its footprint the oracle
    (reference semantics) is non-finite everywhere, the kernel only where the selected stencil touches it."""
    g, og = mk([-1, -1], [1, 1], (24, 40), None)
    rng = np.random.default_rng(2)
    data = O.shape_sphere(og, None, .4) + 0.05 * rng.standard_normal((24, 40))
    data[12, 20] = np.inf
    for dim in (0, 1):
        Lo, Ro = O.upwind_first_eno3(og, data, dim)
        Lk, Rk = L.upwindFirstENO3(g, data, dim)
        idx = np.arange(data.shape[dim])
        far = np.abs(idx - (12 if dim == 0 else 20)) > 3
        sl = [slice(None)] * 2
        sl[dim] = far
        for a, b in ((Lk, Lo), (Rk, Ro)):
            assert np.all(np.isfinite(b[tuple(sl)])) and np.max(np.abs(a[tuple(sl)] - b[tuple(sl)])) <= 1e-11
        line_o = (Lo[:, 20], Ro[:, 20]) if dim == 0 else (Lo[12, :], Ro[12, :])
        line_k = (Lk[:, 20], Rk[:, 20]) if dim == 0 else (Lk[12, :], Rk[12, :])
        c = 12 if dim == 0 else 20
        near = slice(c - 2, c + 3)
        assert not np.any(np.isfinite(line_o[0][near])) and not np.any(np.isfinite(line_o[1][near]))
        bad_k = (~np.isfinite(line_k[0])).sum() + (~np.isfinite(line_k[1])).sum()
        bad_o = (~np.isfinite(line_o[0])).sum() + (~np.isfinite(line_o[1])).sum()
        assert 0 < bad_k <= bad_o
        assert not np.isfinite(line_k[0][c]) or not np.isfinite(line_k[1][c])


# ------------------------------------------------------------------------------ two cells per lane (hj_fusedv.h)
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd,tz", [
    ((40, 37, 29), 2, None),            # odd row length: the last tile of the row is shifted back onto an odd cell
    ((23, 30, 70), (0, 2), None),       # periodic march axis
    ((31, 45, 34), None, (1, 2)),       # all extrapolated, towardZero on two axes
    ((9, 8, 200), (1,), None),          # long rows, several tiles per row
    ((64, 701), None, None),            # 2-D, odd row
    ((30, 5000), (1,), None),           # 2-D, several tiles, periodic along the row
])
def test_pair_kernel_bitwise_equals_scalar_kernel(scheme, n, pd, tz, monkeypatch):
    """fused_pair_kernel (two adjacent cells per lane: 16-byte HBM / LDS accesses), with the double-buffered and with
    the 5-plane-ring LDS schedule, against fused_substep_kernel on an RK3 step, a clamped ydot-only term evaluation and
    a post-step-fused RK2 step: BITWISE."""
    g, ham, par, data = _stage12_case(n, pd, tz)
    res = {}
    flags = ("0", "2", "2r1", "2r2", "2r3")        # 2: the pair kernel whatever the grid size; rK: LDS halo ring K planes ahead
    for flag in flags:
        monkeypatch.setenv("HJ_PAIR", flag[0])
        monkeypatch.setenv("HJ_PAIR_RING", "1" if "r" in flag else "0")
        monkeypatch.setenv("HJ_PAIR_AH", flag[-1] if "r" in flag else "3")
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        y = dg.to_device(data)
        outs = []
        for order, post in ((3, 0), (2, 1)):
            _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, post))
            nxt, w0, w1 = dg.empty(), dg.empty(), dg.empty()
            tout, dtout = C.c_double(), C.c_double()
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, _ffi.SCHEME_IDS[scheme], ham, _ffi.darr(par), 0., 1e9, 0.8, 1e300, 0,
                                         dg.ptr(y), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
            outs.append(nxt)
        _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, 0))
        yd, sb = dg.empty(), C.c_double()
        _ffi.check(dg.lib.hj_lf_term(dg.ctx, _ffi.SCHEME_IDS[scheme], ham, _ffi.darr(par), 0., -1, dg.ptr(y), dg.ptr(yd), C.byref(sb)))
        dg.sync()
        # the comparison must not pass vacuously: the kernel and the LDS schedule that actually ran
        nbuf, ahead = C.c_int(), C.c_int()
        _ffi.check(dg.lib.hj_last_launch(dg.ctx, C.byref(nbuf), C.byref(ahead)))
        assert dg.lib.hj_last_kernel(dg.ctx) == (b"fused_substep_kernel" if flag == "0" else b"fused_pair_kernel"), flag
        if flag != "0":
            assert (nbuf.value, ahead.value) == ((2 + int(flag[-1]), int(flag[-1])) if "r" in flag else (2, 0)), (flag, nbuf.value, ahead.value)
        outs.append(yd)
        res[flag] = (outs, sb.value)
    for other in flags[1:]:
        assert res["0"][1] == res[other][1]
        for a, b in zip(res["0"][0], res[other][0]):
            assert torch.equal(a, b), "%s: max diff %g at %s" % (other, float((a - b).abs().max()),
                                                                 np.unravel_index(int((a - b).abs().argmax()), n))
    assert float((res["0"][0][0] - torch.as_tensor(data, device="cuda")).abs().max()) > 0


@pytest.mark.parametrize("pair", ["0", "2"])
@pytest.mark.parametrize("case", ["dubins_odd", "dubins_big", "integrator_2d"])
def test_weno5_epsilon_reduced_in_the_producing_launch_is_bitwise_the_pre_pass(case, pair, monkeypatch):
    """Intended WENO5: epsilon_d = 1e-6 max(D1_d^2) of a stage's input (upwind_first_weno5a.py:153-156).  Inside hj_rk_step /
    hj_rk_integrate the tiled kernels reduce max(D1^2) of their OUTPUT themselves (pairs inside a tile and chunk;
    eps_seam_kernel adds tile / chunk seams and periodic wrap pairs and folds the rows), so that only the first stage of a
    step runs the two-launch pre-pass.  A max over the same set of values: states and times must be bit for bit those of
    HJ_EPS_FUSE=0, through both tiled kernels, step-wise and inside hj_rk_integrate, RK1 / RK2 / RK3."""
    if case == "integrator_2d":
        g, _ = mk([-1.3, -1.1], [1.2, 1.4], [157, 203], None)
        ham, par = _ffi.HAM_DOUBLE_INTEGRATOR, [1.0, 0., 0., 0.]
        rng = np.random.default_rng(5)
        x0, x1 = np.meshgrid(np.linspace(-1.3, 1.2, 157), np.linspace(-1.1, 1.4, 203), indexing="ij")
        data = np.sqrt(x0 ** 2 + x1 ** 2) - 0.6 + 0.01 * rng.standard_normal(x0.shape)
    else:
        n = (37, 29, 42) if case == "dubins_odd" else (150, 140, 130)       # 2.7 M cells: the pair kernel by default
        g, _ = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], list(n), 2)
        ham, par = _ffi.HAM_DUBINS_REL, [1.0, 1.0, 1.0, 2.0]
        rng = np.random.default_rng(6)
        xs = np.meshgrid(*[np.linspace(-1, 1, k) for k in n], indexing="ij")
        data = np.sqrt(xs[0] ** 2 + xs[1] ** 2) - 0.5 + 0.1 * np.sin(3 * xs[2]) + 0.01 * rng.standard_normal(n)
    res = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("HJ_EPS_FUSE", fuse)
        monkeypatch.setenv("HJ_EPS_FUSE_MIN_CELLS", "0")      # default: only grids of 2 M cells and more (launch floors below)
        monkeypatch.setenv("HJ_PAIR", pair)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        y = torch.as_tensor(data, device="cuda", dtype=torch.float64).contiguous()
        outs = []
        for order in (3, 2, 1):
            # two single steps through hj_rk_step ...
            cur, t = y, 0.0
            for _ in range(2):
                nxt, w0, w1 = dg.empty(), dg.empty(), dg.empty()
                tn, dt = C.c_double(), C.c_double()
                _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, _ffi.SCHEME_IDS["WENO5"], ham, _ffi.darr(par), t, 10.0, 0.8, 1e30, 0,
                                             dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tn), C.byref(dt)))
                cur, t = nxt, tn.value
            dg.sync()
            outs.append((cur.clone(), t))
            # ... and three steps inside one hj_rk_integrate call (the epsilon chain crosses the steps)
            a, b, w = dg.empty(), dg.empty(), dg.empty()
            tt, ns, which = C.c_double(), C.c_int64(), C.c_int()
            _ffi.check(dg.lib.hj_rk_integrate(dg.ctx, order, _ffi.SCHEME_IDS["WENO5"], ham, _ffi.darr(par), 0.0, 10.0, 0.8, 1e30, 0,
                                              dg.ptr(y), dg.ptr(a), dg.ptr(b), dg.ptr(w), 3, -1.0, C.byref(tt), C.byref(ns), C.byref(which)))
            dg.sync()
            assert ns.value == 3
            outs.append(((y, a, b)[which.value].clone(), tt.value))
        assert dg.lib.hj_last_kernel(dg.ctx) == (b"fused_pair_kernel" if pair == "2" else b"fused_substep_kernel")
        nl, fz = C.c_int(), C.c_int()
        _ffi.check(dg.lib.hj_rk_plan(dg.ctx, 3, _ffi.SCHEME_IDS["WENO5"], ham, _ffi.darr(par), 0, C.byref(nl), C.byref(fz)))
        assert nl.value == (6 if fuse == "1" else 9)      # pre-pass (one launch) + 3 substep launches + 2 seam launches | 3 x 3
        # the runtime-flag instantiation (MODE 0): termRestrictUpdate clamp + fused post-step minimum, four steps in one call --
        # the epsilon of a step's first stage then comes from the CLAMPED, MINIMISED output of the previous step's last launch
        _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, 1))
        a, b, w = dg.empty(), dg.empty(), dg.empty()
        tt, ns, which = C.c_double(), C.c_int64(), C.c_int()
        _ffi.check(dg.lib.hj_rk_integrate(dg.ctx, 3, _ffi.SCHEME_IDS["WENO5"], ham, _ffi.darr(par), 0.0, 10.0, 0.8, 1e30, 1,
                                          dg.ptr(y), dg.ptr(a), dg.ptr(b), dg.ptr(w), 4, -1.0, C.byref(tt), C.byref(ns), C.byref(which)))
        dg.sync()
        outs.append(((y, a, b)[which.value].clone(), tt.value))
        _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, 0))
        res[fuse] = outs
        del dg
    for (ya, ta), (yb, tb) in zip(res["0"], res["1"]):
        assert ta == tb
        assert torch.equal(ya, yb), float((ya - yb).abs().max())
    assert float((res["1"][0][0] - torch.as_tensor(data, device="cuda")).abs().max()) > 1e-6


@pytest.mark.parametrize("scheme,pair", [("WENO5_ASSHIPPED", "2"), ("WENO5_ASSHIPPED", "0"), ("ENO3", "2"), ("WENO5", "2")])
def test_tile_shape_rotation_of_the_launch_time_tuner_does_not_change_a_bit(scheme, pair, monkeypatch):
    """On grids of >= 40 M cells the first launches of a launch shape take turns through candidate tile shapes, each timed
    with a pair of events, before the fastest is kept (TuneState, hj_inst.hip).  The launches of the rotation are ordinary
    launches: with the threshold lowered to 0 on a 96 x 90 x 100 grid, 14 RK3 steps (84 launches: through the rotation and past
    its end) must give the state and time of HJ_AUTOTUNE=0 bit for bit, and the rotation must really have happened."""
    n = (96, 90, 100)
    g, _ = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], list(n), 2)
    rng = np.random.default_rng(8)
    xs = np.meshgrid(*[np.linspace(-1, 1, k) for k in n], indexing="ij")
    data = np.sqrt(xs[0] ** 2 + xs[1] ** 2) - 0.5 + 0.1 * np.sin(3 * xs[2]) + 0.01 * rng.standard_normal(n)
    par = [1.0, 1.0, 1.0, 2.0]
    res = {}
    for tune in ("0", "1"):
        monkeypatch.setenv("HJ_AUTOTUNE", tune)
        monkeypatch.setenv("HJ_AUTOTUNE_MIN_MCELLS", "0")
        monkeypatch.setenv("HJ_AUTOTUNE_PASSES", "3")
        monkeypatch.setenv("HJ_PAIR", pair)
        monkeypatch.setenv("HJ_EPS_FUSE_MIN_CELLS", "0")
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        cur, t = torch.as_tensor(data, device="cuda", dtype=torch.float64).contiguous(), 0.0
        shapes = set()
        for _ in range(14):
            nxt, w0, w1 = dg.empty(), dg.empty(), dg.empty()
            tn, dt = C.c_double(), C.c_double()
            _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, _ffi.SCHEME_IDS[scheme], _ffi.HAM_DUBINS_REL, _ffi.darr(par), t, 10.0, 0.8, 1e30, 0,
                                         dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tn), C.byref(dt)))
            e = (C.c_int * 4)()
            _ffi.check(dg.lib.hj_last_tile(dg.ctx, e))
            shapes.add(tuple(e)[1:])
            cur, t = nxt, tn.value
        dg.sync()
        assert dg.lib.hj_last_kernel(dg.ctx) == (b"fused_pair_kernel" if pair == "2" else b"fused_substep_kernel")
        res[tune] = (cur.clone(), t, shapes)
        del dg
    assert res["0"][1] == res["1"][1]
    assert torch.equal(res["0"][0], res["1"][0]), float((res["0"][0] - res["1"][0]).abs().max())
    assert len(res["0"][2]) == 1, res["0"][2]
    assert len(res["1"][2]) >= 2, "the tuner did not rotate through tile shapes: %r" % (res["1"][2],)
