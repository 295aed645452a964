"""GPU: Python hamFunc / partialFunc callbacks nobody wrote a kernel for run FUSED (levelsetpy_amd/trace_ham.py, term.native_plan).

The callbacks are ordinary array code (NumPy, torch or both).  The library traces them once, compiles the recorded expression
with hipRTC and checks the kernel against the callbacks on the first data it meets.  Compared here with (1) the split path the
same schemeData took before (HJ_TRACE=0: derivative kernels -> the callbacks on device arrays -> the dissipation kernel), (2) the
CPU oracle driven by the same callbacks on NumPy arrays, for the term and for the integrators; the costate-range protocol of
artificial_diss_glf.py:80-99; the refusal and the failed-check paths."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd import trace_ham as TH  # noqa: E402
from levelsetpy_amd.context import device_grid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

from test_gpu_parity import mk, sdata, DERIV, SCHEMES, close  # noqa: E402


def _is_t(a):
    return type(a).__module__.startswith("torch")


def _x(grid, d, like):
    x = grid.xs[d]
    return torch.as_tensor(np.asarray(x), device=like.device) if _is_t(like) else x


class DubinsAbs(object):
    """Dubins car in absolute coordinates, worst-case turn rate (reference DynamicalSystems/dubins_absolute.py:150-170), written once for
    NumPy and torch arrays alike."""

    def __init__(self, grid, v, w):
        self.grid, self.v, self.w = grid, v, w

    def hamiltonian(self, t, data, p, sd=None):
        xp = torch if _is_t(p[0]) else np
        x3 = _x(self.grid, 2, p[0])
        return self.v * (p[0] * xp.cos(x3) + p[1] * xp.sin(x3)) + self.w * abs(p[2])

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        xp = torch if _is_t(data) else np
        x3 = _x(self.grid, 2, data)
        if dim == 0:
            return abs(self.v * xp.cos(x3)) + 0 * data
        if dim == 1:
            return abs(self.v * xp.sin(x3)) + 0 * data
        return self.w


class RangeReader(object):
    """H = |p|^2 / 2 + c x0 p1 with where / maximum nodes; alpha from the costate RANGE (artificial_diss_glf.py:80-99)."""

    def __init__(self, grid, c):
        self.grid, self.c = grid, c

    def hamiltonian(self, t, data, p, sd=None):
        xp = torch if _is_t(p[0]) else np
        x0 = _x(self.grid, 0, p[0])
        kin = 0.5 * (p[0] ** 2 + p[1] ** 2 + p[2] ** 2)
        return kin + self.c * x0 * p[1] + 0.1 * xp.where(p[2] > 0, p[2], -0.5 * p[2])

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        lo, hi = abs(dmin[dim]), abs(dmax[dim])         # numbers (GLF: the range over the grid) or arrays (the local variants)
        a = torch.maximum(lo, hi) if _is_t(lo) else np.maximum(lo, hi)
        if dim == 1:
            return a + abs(self.c * _x(self.grid, 0, data))
        if dim == 2:
            return a + 0.15
        return a


def _kernel(g):
    dg = device_grid(g)
    return dg.lib.hj_last_kernel(dg.ctx).decode()


@pytest.mark.parametrize("scheme", SCHEMES)
def test_traced_callbacks_run_fused_and_match_split_path_and_oracle(scheme, monkeypatch):
    n = (22, 20, 24)
    g, og = mk([-2., -2., -np.pi], [2., 2., np.pi * (1 - 2 / n[2])], n, 2)
    d0 = O.shape_sphere(og, None, 1.0) + 0.03 * np.random.default_rng(5).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    veh = DubinsAbs(g, 1.3, 0.7)
    monkeypatch.setenv("HJ_TRACE", "0")
    split, sb_s, _ = L.termLaxFriedrichs(0., y, sdata(g, veh, DERIV[scheme]))
    assert "hipRTC" not in _kernel(g)
    monkeypatch.delenv("HJ_TRACE")
    sd = sdata(g, veh, DERIV[scheme])
    with warnings.catch_warnings():
        warnings.simplefilter("error")                 # the check against the callbacks passes silently
        fused, sb_f, _ = L.termLaxFriedrichs(0., y, sd)
    assert "hipRTC" in _kernel(g), _kernel(g)
    assert abs(sb_f - sb_s) <= 1e-13 * sb_s
    close(fused.cpu().numpy(), split.cpu().numpy(), 1e-11, what="traced vs split")
    yo, sbo = O.term_lax_friedrichs(og, DubinsAbs(og, 1.3, 0.7), scheme, 0., d0.reshape(-1, 1))
    close(fused.cpu().numpy(), yo, 1e-11, what="traced vs oracle")
    assert abs(sb_f - sbo) <= 1e-13 * sbo
    # a speed changed in place: new par[] values, the same kernel (nothing is compiled again)
    before = L.kernel_cache_stats()
    veh.v = 2.0
    f2, sb2, _ = L.termLaxFriedrichs(0., y, sd)
    assert L.kernel_cache_stats() == before and "hipRTC" in _kernel(g)
    yo2, sbo2 = O.term_lax_friedrichs(og, DubinsAbs(og, 2.0, 0.7), scheme, 0., d0.reshape(-1, 1))
    close(f2.cpu().numpy(), yo2, 1e-11)
    assert abs(sb2 - sbo2) <= 1e-13 * sbo2
    # the integrators take the traced plan like any native one: a time span (hj_rk_integrate) against the oracle
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.7)))
    tf = 3.2 * 0.7 * sbo2
    t, yn, _ = L.odeCFL3(L.termLaxFriedrichs, [0., tf], y, op, sd)
    to, yoo = O.ode_cfl_3(lambda tt, yy: O.term_lax_friedrichs(og, DubinsAbs(og, 2.0, 0.7), scheme, tt, yy), [0., tf], d0.reshape(-1, 1), 0.7)
    assert abs(float(t) - to) <= 1e-13
    if scheme.startswith("WENO"):
        close(yn.cpu().numpy(), yoo, 1e-11, what="span")
    else:
        diff = np.abs(yn.cpu().numpy() - yoo)
        assert float(np.mean(diff > 1e-11)) <= 2e-3 and diff.max() <= 1e-3


@pytest.mark.parametrize("diss", ["glf", "llf", "lllf"])
def test_traced_partialfunc_reading_the_costate_range(diss, monkeypatch):
    n = (26, 24, 22)
    g, og = mk([-1., -1., -1.], [1., 1., 1.], n, None)
    d0 = O.shape_sphere(og, None, 0.5) + 0.02 * np.random.default_rng(9).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    obj = RangeReader(g, 0.6)
    dfn = {"glf": L.artificialDissipationGLF, "llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[diss]

    def bundle():
        sd = sdata(g, obj, L.upwindFirstWENO5)
        sd.dissFunc = dfn
        return sd
    monkeypatch.setenv("HJ_TRACE", "0")
    split, sb_s, _ = L.termLaxFriedrichs(0., y, bundle())
    monkeypatch.delenv("HJ_TRACE")
    sd = bundle()
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sd)
    assert "hipRTC" in _kernel(g), _kernel(g)
    assert abs(sb_f - sb_s) <= 1e-12 * sb_s
    close(fused.cpu().numpy(), split.cpu().numpy(), 1e-11, what="traced vs split (%s)" % diss)
    yo, sbo = O.term_lax_friedrichs(og, RangeReader(og, 0.6), "WENO5_ASSHIPPED", 0., d0.reshape(-1, 1), diss=diss)
    close(fused.cpu().numpy(), yo, 1e-11, what="traced vs oracle (%s)" % diss)
    assert abs(sb_f - sbo) <= 1e-12 * sbo
    # three single steps: deltaT comes from the data-dependent bound of the first stage (ode_cfl_3.py:142)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    monkeypatch.setenv("HJ_TRACE", "0")
    sds = bundle()
    ts, ys = 0., y
    for _ in range(3):
        ts, ys, _ = L.odeCFL3(L.termLaxFriedrichs, [ts, 10.], ys, op, sds)
    monkeypatch.delenv("HJ_TRACE")
    tf_, yf = 0., y
    for _ in range(3):
        tf_, yf, _ = L.odeCFL3(L.termLaxFriedrichs, [tf_, 10.], yf, op, sd)
    assert abs(float(tf_) - float(ts)) <= 1e-12
    close(yf.cpu().numpy(), ys.cpu().numpy(), 1e-10, what="three RK3 steps (%s)" % diss)


def test_a_trace_that_disagrees_with_the_callbacks_is_dropped(monkeypatch):
    """The generated kernel is checked against the callbacks on the first data: a tracer defect (here: cos written as sin) costs a warning and
    the split path, never a wrong result."""
    n = (20, 18, 22)
    g, og = mk([-2., -2., -np.pi], [2., 2., np.pi * (1 - 2 / n[2])], n, 2)
    d0 = O.shape_sphere(og, None, 1.0)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    veh = DubinsAbs(g, 0.9, 1.1)
    monkeypatch.setenv("HJ_TRACE", "0")
    split, sb_s, _ = L.termLaxFriedrichs(0., y, sdata(g, veh, L.upwindFirstENO3))
    monkeypatch.delenv("HJ_TRACE")
    monkeypatch.setitem(TH._NUM_UNARY, "cos", "sin({0})")
    sd = sdata(g, veh, L.upwindFirstENO3)
    with pytest.warns(UserWarning, match="disagrees with the callbacks"):
        out, sb, _ = L.termLaxFriedrichs(0., y, sd)
    assert torch.equal(out, split) and sb == sb_s            # the callbacks' own result, bit for bit
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out2, _, _ = L.termLaxFriedrichs(0., y, sd)            # remembered: no second attempt, no second warning
    assert torch.equal(out2, split)


def test_untraceable_callbacks_keep_the_split_path(monkeypatch):
    n = (20, 18, 22)
    g, og = mk([-2., -2., -np.pi], [2., 2., np.pi * (1 - 2 / n[2])], n, 2)
    y = torch.as_tensor(O.shape_sphere(og, None, 1.0).reshape(-1, 1), device="cuda")

    class Branchy(DubinsAbs):
        def hamiltonian(self, t, data, p, sd=None):
            if float(p[0].max()) > 0:                    # Python control flow on array values
                return DubinsAbs.hamiltonian(self, t, data, p, sd)
            return 0 * p[0]
    veh = Branchy(g, 1.0, 1.0)
    a, sa, _ = L.termLaxFriedrichs(0., y, sdata(g, veh, L.upwindFirstWENO5))
    assert "hipRTC" not in _kernel(g)
    monkeypatch.setenv("HJ_TRACE", "0")
    b, sb, _ = L.termLaxFriedrichs(0., y, sdata(g, veh, L.upwindFirstWENO5))
    assert torch.equal(a, b) and sa == sb


def test_traced_2d_restrict_update_and_hjipde_solve():
    """2-D: a double integrator written as Python callbacks against the built-in system -- termRestrictUpdate steps and HJIPDE_solve."""
    g, og = mk([-1, -1], [1, 1], [70, 64], None)
    d0 = O.shape_sphere(og, None, .35)

    class DInt(object):
        def __init__(self, grid, u):
            self.grid, self.u = grid, u

        def hamiltonian(self, t, data, p, sd=None):
            return -(p[0] * _x(self.grid, 1, p[0]) - abs(p[1]) * self.u)

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            return abs(_x(self.grid, 1, data)) + 0 * data if dim == 0 else abs(self.u)
    user, builtin = DInt(g, 1.0), L.DoubleIntegrator(g, 1)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    res = []
    for sys_ in (user, builtin):
        sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sdata(g, sys_, L.upwindFirstWENO5), positive=0))
        y, t = torch.as_tensor(d0.reshape(-1), device="cuda"), 0.
        for _ in range(4):
            t, y, _ = L.odeCFL3(L.termRestrictUpdate, [t, 10.], y, op, sdr)
        if sys_ is user:
            assert "hipRTC" in _kernel(g), _kernel(g)
        res.append((t, y.cpu().numpy()))
    assert abs(res[0][0] - res[1][0]) <= 1e-15
    close(res[0][1], res[1][1], 1e-12, what="restrict update")
    outs = []
    for sys_ in (user, builtin):
        sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation))
        data, tau, _ = L.HJIPDE_solve(d0, [0., 0.02, 0.05], sd, 'minVOverTime', L.Bundle(dict(quiet=True, keepLast=True)))
        outs.append(np.asarray(data))
    close(outs[0], outs[1], 1e-12, what="HJIPDE_solve")


def test_traced_4d_fp32():
    """4-D, single precision: the traced kernel is instantiated for float as well; against the split path at fp32 rounding."""
    n = (14, 12, 13, 20)
    g, og = mk([-1., -1., -1., -1.], [1., 1., 1., 1.], n, [0, 1, 2, 3])

    class Quad(object):
        def __init__(self, grid):
            self.grid = grid

        def hamiltonian(self, t, data, p, sd=None):
            xp = torch if _is_t(p[0]) else np
            return p[0] * _x(self.grid, 1, p[0]) - 0.5 * abs(p[1]) + xp.cos(_x(self.grid, 3, p[0])) * p[2] + 0.25 * p[3] ** 2

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            return [abs(_x(self.grid, 1, data)) + 0 * data, 0.5, 1.0, 0.6][dim]
    rng = np.random.default_rng(11)
    y = torch.as_tensor((O.shape_sphere(og, None, 0.5) + 0.01 * rng.standard_normal(og.shape)).reshape(-1, 1), device="cuda", dtype=torch.float32)
    obj = Quad(g)
    import os
    os.environ["HJ_TRACE"] = "0"
    try:
        split, sb_s, _ = L.termLaxFriedrichs(0., y, sdata(g, obj, L.upwindFirstENO2))
    finally:
        del os.environ["HJ_TRACE"]
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sdata(g, obj, L.upwindFirstENO2))
    dg = device_grid(g, "float32")
    assert b"hipRTC" in dg.lib.hj_last_kernel(dg.ctx)
    assert fused.dtype == torch.float32 and abs(sb_f - sb_s) <= 1e-5 * sb_s
    close(fused.cpu().numpy(), split.cpu().numpy(), 2e-4, what="4-D fp32 traced vs split")


def test_traced_first_use_with_numpy_arrays():
    """A NumPy caller: the check against the callbacks runs on host arrays, the result comes back as an ndarray(-like) and equals the split path's."""
    n = (24, 22, 20)
    g, og = mk([-2., -2., -np.pi], [2., 2., np.pi * (1 - 2 / n[2])], n, 2)

    class Car(DubinsAbs):              # a class of its own: an expression no other test has registered and verified
        def hamiltonian(self, t, data, p, sd=None):
            return DubinsAbs.hamiltonian(self, t, data, p, sd) + 0.125 * p[0] * p[1]

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            a = DubinsAbs.dissipation(self, t, data, dmin, dmax, sd, dim)
            return a + 0.125 * np.maximum(abs(dmin[1 - dim]), abs(dmax[1 - dim])) if dim < 2 else a
    y = (O.shape_sphere(og, None, 1.0) + 0.02 * np.random.default_rng(4).standard_normal(og.shape)).reshape(-1, 1)
    veh = Car(g, 1.1, 0.8)
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sdata(g, veh, L.upwindFirstENO3))
    assert "hipRTC" in _kernel(g), _kernel(g)
    yo, sbo = O.term_lax_friedrichs(og, Car(og, 1.1, 0.8), "ENO3", 0., y)
    close(np.asarray(fused), yo, 1e-11, what="NumPy in: traced vs oracle")
    assert abs(sb_f - sbo) <= 1e-12 * sbo


@pytest.mark.parametrize("where", ["numpy", "cuda"])
def test_traced_stored_per_axis_tables(where, monkeypatch):
    """Per-axis arrays computed before the call (NumPy arrays, or device tensors the split path multiplies with directly) are tables in the traced
    kernel: fused vs the split path vs the oracle."""
    n = (23, 21, 26)
    g, og = mk([-2., -2., -np.pi], [2., 2., np.pi * (1 - 2 / n[2])], n, 2)

    class Stored(object):
        def __init__(self, grid, dev):
            self.grid, self.v = grid, 1.2
            conv = (lambda a: torch.as_tensor(a, device="cuda")) if dev else (lambda a: a)
            self.c3 = conv(np.cos(np.asarray(grid.xs[2])))
            self.s3 = conv(np.sin(np.asarray(grid.xs[2])))
            self.gain = conv(np.linspace(1., 1.5, int(grid.shape[0])).reshape(-1, 1, 1))

        def _a(self, a, like):
            return torch.as_tensor(a, device=like.device) if (_is_t(like) and not _is_t(a)) else a

        def hamiltonian(self, t, data, p, sd=None):
            c3, s3, gain = (self._a(a, p[0]) for a in (self.c3, self.s3, self.gain))
            return self.v * (p[0] * c3 + p[1] * s3) * gain + 0.8 * abs(p[2])

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            c3, s3, gain = (self._a(a, data) for a in (self.c3, self.s3, self.gain))
            # (a product of two stored arrays of DIFFERENT axes would be a two-axis array -- refused; each meets the data on its own here)
            return [(abs(self.v * c3) + 0 * data) * gain, (abs(self.v * s3) + 0 * data) * gain, 0.8][dim]
    d0 = O.shape_sphere(og, None, 1.0) + 0.03 * np.random.default_rng(6).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    obj = Stored(g, where == "cuda")
    monkeypatch.setenv("HJ_TRACE", "0")
    split, sb_s, _ = L.termLaxFriedrichs(0., y, sdata(g, obj, L.upwindFirstWENO5))
    monkeypatch.delenv("HJ_TRACE")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fused, sb_f, _ = L.termLaxFriedrichs(0., y, sdata(g, obj, L.upwindFirstWENO5))
    assert "hipRTC" in _kernel(g), _kernel(g)
    assert abs(sb_f - sb_s) <= 1e-13 * sb_s
    close(fused.cpu().numpy(), split.cpu().numpy(), 1e-11, what="tables: traced vs split")
    yo, sbo = O.term_lax_friedrichs(og, Stored(og, False), "WENO5_ASSHIPPED", 0., d0.reshape(-1, 1))
    close(fused.cpu().numpy(), yo, 1e-11, what="tables: traced vs oracle")
    assert abs(sb_f - sbo) <= 1e-13 * sbo


def test_traced_more_than_four_parameters_in_place(monkeypatch):
    """Eight parameter slots (HamTables::par): seven floats of a system, one of them changed in place -- same kernel, new value."""
    n = (21, 20, 22)
    g, og = mk([-1., -1., -1.], [1., 1., 1.], n, None)

    class Seven(object):
        def __init__(self, grid):
            self.grid = grid
            self.a, self.b, self.c, self.d, self.e, self.f, self.g = 0.31, 0.47, 0.59, 0.73, 0.83, 0.97, 1.13

        def hamiltonian(self, t, data, p, sd=None):
            x = [_x(self.grid, k, p[0]) for k in range(3)]
            return self.a * p[0] * x[1] + self.b * p[1] * x[2] + self.c * p[2] * x[0] + self.d * abs(p[0]) + self.e * abs(p[1]) + self.f * p[0] * p[1] \
                + self.g * p[2] ** 2

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            x = [_x(self.grid, k, data) for k in range(3)]
            lo, hi = abs(dmin[dim]), abs(dmax[dim])
            m = torch.maximum(lo, hi) if _is_t(lo) else np.maximum(lo, hi)
            return [abs(self.a * x[1]) + self.d + self.f * m, abs(self.b * x[2]) + self.e + self.f * m, abs(self.c * x[0]) + 2 * self.g * m][dim] + 0 * data
    d0 = O.shape_sphere(og, None, 0.5) + 0.02 * np.random.default_rng(12).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    obj = Seven(g)
    assert len(TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None).params) == 8          # a .. g and the product 2 g: all eight slots
    sd = sdata(g, obj, L.upwindFirstENO3)
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sd)
    assert "hipRTC" in _kernel(g), _kernel(g)
    ov = Seven(og)
    yo, sbo = O.term_lax_friedrichs(og, ov, "ENO3", 0., d0.reshape(-1, 1))
    close(fused.cpu().numpy(), yo, 1e-11, what="seven parameters")
    assert abs(sb_f - sbo) <= 1e-12 * sbo
    before = L.kernel_cache_stats()
    obj.g = ov.g = 0.4                   # the LAST one: slot 6
    f2, sb2, _ = L.termLaxFriedrichs(0., y, sd)
    assert L.kernel_cache_stats() == before
    yo2, sbo2 = O.term_lax_friedrichs(og, ov, "ENO3", 0., d0.reshape(-1, 1))
    close(f2.cpu().numpy(), yo2, 1e-11, what="seven parameters, one changed")
    assert abs(sb2 - sbo2) <= 1e-12 * sbo2 and abs(sb2 - sb_f) > 1e-6


def test_traced_scalar_idioms_of_the_glf_protocol(monkeypatch):
    """partialFunc written for the numbers GLF hands it -- max(abs(float(derivMin[d])), abs(float(derivMax[d]))) -- is traced after the source rewrite
    and runs the range pass + fused substep; against the split path (the callbacks as written) and the oracle."""
    from test_trace_ham import ScalarIdioms
    n = (24, 22, 20)
    g, og = mk([-1., -1., -1.], [1., 1., 1.], n, None)
    d0 = O.shape_sphere(og, None, 0.5) + 0.02 * np.random.default_rng(3).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")

    class Sys(ScalarIdioms):
        def hamiltonian(self, t, data, p, sd=None):
            x0 = _x(self.grid, 0, p[0])
            return 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + self.c * x0 * p[1] + 0.955 * p[2]

        def dissipation(self, t, data, derivMin, derivMax, sd, dim):
            a = max(abs(float(derivMin[dim])), abs(float(derivMax[dim])))
            a = a + self.gain if dim == 2 else a
            if dim != 1:
                return a
            return a + abs(self.c * _x(self.grid, 0, data))
    obj = Sys(g, 0.7)
    monkeypatch.setenv("HJ_TRACE", "0")
    split, sb_s, _ = L.termLaxFriedrichs(0., y, sdata(g, obj, L.upwindFirstWENO5))
    monkeypatch.delenv("HJ_TRACE")
    sd = sdata(g, obj, L.upwindFirstWENO5)
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sd)
    assert "hipRTC" in _kernel(g), _kernel(g)
    assert abs(sb_f - sb_s) <= 1e-12 * sb_s
    close(fused.cpu().numpy(), split.cpu().numpy(), 1e-11, what="scalar idioms: traced vs split")
    yo, sbo = O.term_lax_friedrichs(og, Sys(og, 0.7), "WENO5_ASSHIPPED", 0., d0.reshape(-1, 1))
    close(fused.cpu().numpy(), yo, 1e-11, what="scalar idioms: traced vs oracle")
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t1, y1 = 0., y
    for _ in range(3):
        t1, y1, _ = L.odeCFL3(L.termLaxFriedrichs, [t1, 10.], y1, op, sd)
    to, yoo = 0., d0.reshape(-1, 1)
    for _ in range(3):
        to, yoo = O.ode_cfl_3(lambda tt, yy: O.term_lax_friedrichs(og, Sys(og, 0.7), "WENO5_ASSHIPPED", tt, yy), [to, 10.], yoo, 0.8, single_step=True)
    assert abs(float(t1) - to) <= 1e-12
    close(y1.cpu().numpy(), yoo, 1e-10, what="scalar idioms: three RK3 steps vs oracle")
