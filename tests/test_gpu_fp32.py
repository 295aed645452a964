"""Single precision on 2-D / 3-D grids with the built-in systems (round 5, late).  A kernel-trace of the whole GPU suite showed that it launched
none of the float instantiations of the Dubins / double-integrator kernels (tools/experiments/r05_run52.sh: fp32 was exercised on 4-D grids and with
run-time Hamiltonians only), although float32 data on such grids select them.  Here: every scheme, the three size classes the launcher distinguishes
(direct kernel, one-cell-per-lane tiled kernel, pair kernel from 6.5 M cells), term and integrator steps against the fp64 oracle (small sizes) or the
fp64 product path (large sizes: itself checked against the oracle elsewhere), and the fp32 instantiations of the helper kernels (post-step min / max,
NaN guard, restricted update, split path, termNormal / termReinit / termConvection, computeGradients)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd.context import device_grid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

from test_gpu_parity import mk, sdata, DERIV, SCHEMES, dubins  # noqa: E402


def _fp32_close(got, ref, scheme, what=""):
    """fp32 result against an fp64 reference: the WENO stencils are smooth in the data (1e-4 of the largest value); an ENO stencil choice may flip
    where two divided differences agree to fp32 rounding -- isolated cells, bounded in number and size (the rule of the 4-D fp32 tests)."""
    got, ref = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    rel = np.abs(got - ref) / max(float(np.abs(ref).max()), 1e-30)
    if scheme.startswith("WENO"):
        assert rel.max() <= 2e-4, (what, rel.max())
    else:
        assert np.mean(rel > 2e-4) <= 3e-3 and rel.max() <= 0.2, (what, float(np.mean(rel > 2e-4)), rel.max())


def _kernel(g):
    dg = device_grid(g, "float32")
    return dg.lib.hj_last_kernel(dg.ctx)


CASES = {
    # (system, grid shape): small -> the direct kernel with the product's default threshold; mid -> one cell per lane; big -> the pair kernel
    "dubins": {"small": (31, 29, 27), "mid": (66, 60, 58), "big": (190, 186, 188)},
    "integrator": {"small": (60, 50), "mid": (400, 380), "big": (2600, 2600)},
}


def _setup(system, shape, rounded=True):
    if system == "dubins":
        g, og = dubins(list(shape))
        sys_, osys = L.DubinsVehicleRel(g, 1, 1), O.DubinsRel(og, 1, 1)
        d0 = O.shape_cylinder(og, 2, None, .5) + 0.05 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    else:
        g, og = mk([-1., -1.5], [1., 1.5], shape, None)
        sys_, osys = L.DoubleIntegrator(g, 1.25), O.DoubleIntegrator(og, 1.25)
        d0 = O.shape_sphere(og, None, .45) + 0.05 * np.sin(4 * og.xs[0]) * np.cos(3 * og.xs[1])
    return g, og, sys_, osys, (d0.astype(np.float32) if rounded else d0)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("size", ["small", "mid", "big"])
@pytest.mark.parametrize("system", ["dubins", "integrator"])
def test_fp32_builtin_systems_every_scheme_and_size_class(system, size, scheme, monkeypatch):
    if size == "small":
        monkeypatch.setenv("HJ_DIRECT_BELOW", "140000")          # the product default (this suite runs with 0: tests/conftest.py)
    g, og, sys_, osys, d0 = _setup(system, CASES[system][size])
    y32 = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    y64 = y32.double()
    assert y32.dtype == torch.float32
    sd = sdata(g, sys_, DERIV[scheme])
    f32, sb32, _ = L.termLaxFriedrichs(0., y32, sd)
    kern = _kernel(g)
    assert f32.dtype == torch.float32
    if system == "dubins":
        assert kern == {"small": b"direct_substep_kernel", "mid": b"fused_substep_kernel", "big": b"fused_pair_kernel"}[size], kern
    else:       # (an explicit HJ_DIRECT_BELOW applies to 2-D grids too; the default only to 3-D ones)
        assert kern == {"small": b"direct_substep_kernel", "mid": b"fused_substep_kernel", "big": b"fused_pair_kernel"}[size], kern
    if size == "big":            # the fp64 product path on the same (fp32-representable) data: checked against the oracle at this size elsewhere
        ref, sbr, _ = L.termLaxFriedrichs(0., y64, sd)
        ref = ref.cpu().numpy()
    else:
        ref, sbr = O.term_lax_friedrichs(og, osys, scheme, 0., d0.astype(np.float64).reshape(-1, 1))
    _fp32_close(f32.cpu().numpy(), ref, scheme, "term")
    assert abs(sb32 - sbr) <= 1e-5 * sbr, (sb32, sbr)
    # one step of every order
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    for order, ode, oode in ((1, L.odeCFL1, O.ode_cfl_1), (2, L.odeCFL2, O.ode_cfl_2), (3, L.odeCFL3, O.ode_cfl_3)):
        t, y1, _ = ode(L.termLaxFriedrichs, [0., 10.], y32, op, sd)
        assert y1.dtype == torch.float32
        if size == "big":
            tr, yr, _ = ode(L.termLaxFriedrichs, [0., 10.], y64, op, sd)
            yr = yr.cpu().numpy()
        else:
            tr, yr = oode(lambda tt, v: O.term_lax_friedrichs(og, osys, scheme, tt, v), [0., 10.], d0.astype(np.float64).reshape(-1, 1), 0.8, single_step=True)
        assert abs(t - tr) <= 1e-5 * tr, (order, t, tr)
        # y1 = y0 + O(dt) * ydot: the state is compared at fp32 resolution of the state itself
        diff = np.abs(y1.cpu().numpy().astype(np.float64).reshape(-1) - np.asarray(yr).reshape(-1))
        scale = max(1.0, float(np.abs(yr).max()))
        if scheme.startswith("WENO"):
            assert diff.max() <= 2e-6 * scale, (order, diff.max())
        else:
            assert np.mean(diff > 2e-6 * scale) <= 3e-3 and diff.max() <= 1e-3 * scale, (order, float(np.mean(diff > 2e-6 * scale)), diff.max())


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3"])
@pytest.mark.parametrize("system,size", [("dubins", "mid"), ("integrator", "big"), ("dubins", "small")])
def test_fp32_fast_eno_mode(system, size, scheme, monkeypatch):
    """set_eno_mode('fast') in single precision (scheme ids 4 / 5: their float instantiations) against the exact mode's fp32 result."""
    if size == "small":
        monkeypatch.setenv("HJ_DIRECT_BELOW", "140000")
    g, og, sys_, osys, d0 = _setup(system, CASES[system][size])
    y32 = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    sd = sdata(g, sys_, DERIV[scheme])
    exact, sbe, _ = L.termLaxFriedrichs(0., y32, sd)
    L.set_eno_mode('fast')
    try:
        fast, sbf, _ = L.termLaxFriedrichs(0., y32, sd)
    finally:
        L.set_eno_mode('exact')
    assert fast.dtype == torch.float32 and abs(sbe - sbf) <= 1e-6 * sbe
    _fp32_close(fast.cpu().numpy(), exact.cpu().numpy().astype(np.float64), scheme, "fast vs exact")


def test_fp32_helper_kernels_through_the_c_abi():
    """The float instantiations of the helper kernels on a float32 context (HJIPDE_solve itself computes in fp64, as the reference does): the post-step
    min folded into hj_rk_step, hj_minmax_with, hj_any_nan, hj_rk_combine, and the restricted update through odeCFL2 -- against NumPy on the same values."""
    import ctypes as C
    from levelsetpy_amd import _ffi
    from levelsetpy_amd.context import DeviceGrid
    g, og, sys_, osys, d0 = _setup("dubins", (40, 38, 36))
    dg = DeviceGrid(g, "float32")
    dg.bind_stream()
    y = torch.as_tensor(d0, device="cuda")
    par, sid = _ffi.darr([1., 1., 1., 2.]), _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    outs = {}
    for post in (0, 1):                                      # 1: out = min(out, state at the start of the step) (minVOverTime)
        _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, post))
        nxt, w0, w1 = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
        tout, dtout = C.c_double(), C.c_double()
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DUBINS_REL, par, 0., 1e9, 0.8, 1e300, 0, dg.ptr(y), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1),
                                     C.byref(tout), C.byref(dtout)))
        outs[post] = nxt
    _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, 0))
    torch.cuda.synchronize()
    assert outs[0].dtype == torch.float32 and torch.equal(outs[1], torch.minimum(outs[0], y))
    assert bool((outs[0] > y).any()) and bool((outs[0] < y).any())               # the min really selects on both sides
    a, b = outs[0].clone(), y.clone()
    _ffi.check(dg.lib.hj_minmax_with(dg.ctx, _ffi.OP_MAX, dg.ptr(a), dg.ptr(b), a.numel()))
    assert torch.equal(a, torch.maximum(outs[0], y))
    has = C.c_int(-1)
    _ffi.check(dg.lib.hj_any_nan(dg.ctx, dg.ptr(a), a.numel(), C.byref(has)))
    assert has.value == 0
    a.view(-1)[12345] = float("nan")
    _ffi.check(dg.lib.hj_any_nan(dg.ctx, dg.ptr(a), a.numel(), C.byref(has)))
    assert has.value == 1
    rng = np.random.default_rng(9)
    x0, yy, zz = (rng.standard_normal(4097).astype(np.float32) for _ in range(3))
    tx, ty, tz = (torch.as_tensor(v, device="cuda") for v in (x0, yy, zz))
    dt = np.float32(0.0123)
    for mode, ref in {1: yy + dt * zz, 4: np.float32(0.5) * (x0 + (yy + dt * zz))}.items():
        out = torch.empty_like(tx)
        _ffi.check(dg.lib.hj_rk_combine(dg.ctx, mode, float(dt), dg.ptr(tx), dg.ptr(ty), dg.ptr(tz), dg.ptr(out), out.numel()))
        assert float(np.abs(out.cpu().numpy() - ref).max()) <= 1e-6
    # termRestrictUpdate on float32 tensors: ydot clamped inside the kernel (MODE 0 float instantiation)
    sd = sdata(g, sys_, L.upwindFirstWENO5)
    sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sd, positive=0))
    y32 = y.reshape(-1)
    f, sb, _ = L.termRestrictUpdate(0., y32, sdr)
    fu, sbu, _ = L.termLaxFriedrichs(0., y32, sd)
    assert f.dtype == torch.float32 and torch.equal(f.reshape(-1), torch.clamp(fu.reshape(-1), max=0.)) and sb == sbu
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t, y1, _ = L.odeCFL2(L.termRestrictUpdate, [0., 10.], y32, op, sdr)
    t64, y164, _ = L.odeCFL2(L.termRestrictUpdate, [0., 10.], y32.double(), op, sdr)
    assert y1.dtype == torch.float32 and abs(t - t64) <= 1e-5 * t64 and float((y1.double() - y164).abs().max()) <= 5e-6
    assert bool((y1 <= y32 + 1e-7).all())


def test_fp32_split_path_terms_and_gradients():
    """The float instantiations off the fused path: foreign Python callbacks (derivative kernels + lf_split_end), computeGradients (upwind_all) and
    termNormal / termReinit / termConvection (term_kernel) on float32 data against their fp64 runs."""
    g, og, sys_, osys, d0 = _setup("dubins", (36, 34, 30))
    y32 = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    y64 = y32.double()

    class Foreign(object):                      # not recognised as a built-in system: the split path
        def hamiltonian(self, t, data, p, sd=None):
            return sys_.hamiltonian(t, data, p, sd)

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            return sys_.dissipation(t, data, dmin, dmax, sd, dim)
    fo = Foreign()
    for scheme in ("WENO5_ASSHIPPED", "ENO2"):
        sd = L.Bundle(dict(grid=g, hamFunc=fo.hamiltonian, partialFunc=fo.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=DERIV[scheme]))
        a, sba, _ = L.termLaxFriedrichs(0., y32, sd)
        b, sbb, _ = L.termLaxFriedrichs(0., y64, sd)
        assert a.dtype == torch.float32 and abs(sba - sbb) <= 1e-5 * sbb
        _fp32_close(a.cpu().numpy(), b.cpu().numpy(), scheme, "split path")
    grads32 = L.computeGradients(g, y32.reshape(g.shape))
    grads64 = L.computeGradients(g, y64.reshape(g.shape))
    for a, b in zip(grads32[0], grads64[0]):
        assert a.dtype == torch.float32
        _fp32_close(a.cpu().numpy(), b.cpu().numpy(), "WENO5_ASSHIPPED", "computeGradients")
    speed = 0.5 + 0.2 * np.cos(og.xs[0])
    sp32 = torch.as_tensor(speed.astype(np.float32), device="cuda")
    for name in ("normal", "reinit", "convection"):
        fn = {"normal": L.termNormal, "reinit": L.termReinit, "convection": L.termConvection}[name]
        extra32 = {"normal": dict(speed=sp32), "reinit": dict(initial=y32.reshape(g.shape)), "convection": dict(velocity=[0.3, -0.2, 0.5])}[name]
        extra64 = {"normal": dict(speed=sp32.double()), "reinit": dict(initial=y64.reshape(g.shape)), "convection": dict(velocity=[0.3, -0.2, 0.5])}[name]
        a, sba, _ = fn(0., y32, L.Bundle(dict(grid=g, derivFunc=L.upwindFirstWENO5, **extra32)))
        b, sbb, _ = fn(0., y64, L.Bundle(dict(grid=g, derivFunc=L.upwindFirstWENO5, **extra64)))
        assert a.dtype == torch.float32 and abs(sba - sbb) <= 1e-4 * sbb, (name, sba, sbb)
        ra = np.abs(a.cpu().numpy().astype(np.float64) - b.cpu().numpy()).reshape(-1)
        scale = float(b.abs().max())
        # (termReinit's sign function and Godunov switches are discontinuous in the data: isolated cells may take the other branch in fp32)
        assert np.mean(ra > 5e-4 * scale) <= (5e-3 if name == "reinit" else 1e-4) and (name == "reinit" or ra.max() <= 5e-4 * scale), (name, ra.max(), scale)


# ------------------------------------------------------------------------------ the 4-D fp32 kernel with compile-time tiles: every tile, the other light stencils
@pytest.mark.parametrize("scheme", ["ENO2", "WENO5_ASSHIPPED", "ENO2_FAST"])
@pytest.mark.parametrize("n,pd,tile", [((30, 24, 26, 140), (0, 1, 2, 3), (5, 6, 66)), ((200, 4, 40, 80), (0, 1, 2, 3), (3, 5, 66)),
                                       ((40, 40, 40, 40), (0, 1, 2, 3), (5, 6, 34)), ((30, 24, 26, 140), (0, 2), (5, 6, 66))])
def test_pair4_kernel_every_tile_and_light_stencil_equals_the_generic_pair_kernel(n, pd, tile, scheme, monkeypatch):
    """fused_pair4_kernel (hj_fused4v.h) is instantiated per tile (5x6x66, 3x5x66, 5x6x34), light stencil (ENO2, as-shipped WENO5, fast ENO2), boundary
    class (all-periodic or not) and stage class; the C5 tests run ONE of them.  Here each tile on a grid that selects it (2.5 M cells and more), each
    stencil, periodic and mixed boundaries, the term (flag-carrying instantiation) and an RK3 step (the plain-stage instantiations) against the generic
    pair kernel (HJ_PAIR4=0) on the same data: the same per-cell arithmetic, equal at fp32 rounding."""
    import ctypes as C
    from levelsetpy_amd import _ffi
    from levelsetpy_amd.context import DeviceGrid
    from test_gpu_configs import pendulum_grid
    from test_gpu_round4 import sphere4, PAR_PENDULUM
    fast = scheme.endswith("_FAST")
    sid = {"ENO2": _ffi.SCHEME_IDS["ENO2"], "WENO5_ASSHIPPED": _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], "ENO2_FAST": 4}[scheme]
    res = {}
    monkeypatch.setenv("HJ_FLAT4", "0")        # (the full-row kernel of round 6 would take these grids first: tests/test_gpu_round6.py)
    for pair4 in ("1", "0"):
        monkeypatch.setenv("HJ_PAIR4", pair4)
        g, _ = pendulum_grid(n, pd=pd, low_mem=True)
        dg = DeviceGrid(g, "float32")
        dg.bind_stream()
        y = sphere4(g, noise=0.01, seed=3)
        par = _ffi.darr(PAR_PENDULUM)
        yd, sb = torch.empty_like(y), C.c_double()
        _ffi.check(dg.lib.hj_lf_term(dg.ctx, sid, _ffi.HAM_DOUBLE_PENDULUM, par, 0., 0, dg.ptr(y), dg.ptr(yd), C.byref(sb)))
        kern = dg.lib.hj_last_kernel(dg.ctx)
        e = (C.c_int * 4)()
        _ffi.check(dg.lib.hj_last_tile(dg.ctx, e))
        if pair4 == "1":
            assert kern == b"fused_pair4_kernel" and (e[1], e[2], e[3]) == tile, (kern, list(e))
        else:
            assert kern == b"fused_pair_kernel", kern
        nxt, w0, w1 = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
        tout, dtout = C.c_double(), C.c_double()
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, _ffi.HAM_DOUBLE_PENDULUM, par, 0., 1e9, 0.8, 1e300, 0, dg.ptr(y), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1),
                                     C.byref(tout), C.byref(dtout)))
        torch.cuda.synchronize()
        res[pair4] = (yd, sb.value, nxt, tout.value, y)
    a, b = res["1"], res["0"]
    assert abs(a[1] - b[1]) <= 1e-6 * b[1] and abs(a[3] - b[3]) <= 1e-6 * b[3]
    scale = float(b[0].abs().max())
    d = (a[0] - b[0]).abs()
    if fast or scheme == "ENO2":
        assert float((d > 1e-4 * scale).float().mean()) <= 1e-3 and float(d.max()) <= 0.2 * scale, (float(d.max()), scale)
    else:
        assert float(d.max()) <= 1e-4 * scale, (float(d.max()), scale)
    d2 = (a[2] - b[2]).abs()
    assert float((d2 > 2e-6).float().mean()) <= 1e-3 and float(d2.max()) <= 1e-3, float(d2.max())
    assert bool(torch.isfinite(a[2]).all()) and float((a[2] - a[4]).abs().max()) > 1e-5          # the step moved the data


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3"])
@pytest.mark.parametrize("system,size", [("dubins", "mid"), ("integrator", "big"), ("dubins", "small"), ("dubins", "big")])
def test_fp64_fast_eno_mode_across_size_classes(system, size, scheme, monkeypatch):
    """set_eno_mode('fast') in double precision on every size class (direct, one-cell-per-lane, pair kernel): the lean arithmetic against the
    reference-order one -- the same stencil choice except where two divided differences tie to rounding, so equal to ~1e-12 nearly everywhere."""
    if size == "small":
        monkeypatch.setenv("HJ_DIRECT_BELOW", "140000")
    # (data NOT rounded to fp32: on the fine grids the third differences of fp32-rounded data are rounding noise, and every stencil choice a tie)
    g, og, sys_, osys, d0 = _setup(system, CASES[system][size], rounded=False)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    sd = sdata(g, sys_, DERIV[scheme])
    exact, sbe, _ = L.termLaxFriedrichs(0., y, sd)
    L.set_eno_mode('fast')
    try:
        fast, sbf, _ = L.termLaxFriedrichs(0., y, sd)
        op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
        tf, yf, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y, op, sd)
    finally:
        L.set_eno_mode('exact')
    te, ye, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y, op, sd)
    assert abs(sbe - sbf) <= 1e-13 * sbe and abs(te - tf) <= 1e-13 * te
    scale = float(exact.abs().max())
    d = (fast - exact).abs()
    assert float((d > 1e-10 * scale).double().mean()) <= 2e-3 and float(d.max()) <= 0.2 * scale, (float((d > 1e-10 * scale).double().mean()), float(d.max()))
    d2 = (yf - ye).abs()
    assert float((d2 > 1e-11).double().mean()) <= 2e-3 and float(d2.max()) <= 1e-3
