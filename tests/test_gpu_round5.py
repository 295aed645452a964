"""Round-5 GPU tests.

* the 4-D compile-time-tile kernel (csrc/hj_fused4v.h: `fused_pair4_kernel`, the kernel BASELINE C5 runs from round 5):
  against the fp64 oracle at the tolerance SURVEY 8(c) states for fp32 (1e-4 relative), bit for bit against the independent
  direct kernel and the one-cell-per-lane kernel, on grids just above its 5 x 6 x 34 tile (shifted last tiles, all-periodic
  and mixed boundary conditions: the PG instantiation), through plane ranges, the termRestrictUpdate clamp and the CFL bound.
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

from test_gpu_parity import sdata, DERIV, _substep  # noqa: E402
from test_gpu_configs import pendulum_grid  # noqa: E402


def _substep_rs(dg, scheme, ham, par, stage, dt, y, y0, out, restrict_sign):
    _ffi.check(dg.lib.hj_rk_substep(dg.ctx, _ffi.SCHEME_IDS[scheme], ham, _ffi.darr(par), 0., stage, dt, restrict_sign,
                                    dg.ptr(y), dg.ptr(y0) if y0 is not None else None, dg.ptr(out), 3, 0, dg.shape[0]))


@pytest.fixture(autouse=True)
def _compile_time_tile_kernel(monkeypatch):
    """This file's 4-D tests are about fused_pair4_kernel (hj_fused4v.h).  Since round 6 grids whose last axis fits a row of the LDS box take
    fused_flat4_kernel (hj_flat4v.h) first -- tests/test_gpu_round6.py covers that one; here it is switched off."""
    monkeypatch.setenv("HJ_FLAT4", "0")


def _last_kernel(g):
    dg = g.__dict__["_hj_device"]
    dg = dg[next(iter(dg))] if isinstance(dg, dict) else dg
    return dg.lib.hj_last_kernel(dg.ctx)


@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "ENO2"])
@pytest.mark.parametrize("n,pd", [((8, 7, 9, 40), (0, 1, 2, 3)),        # all periodic: the lean instantiation; every axis has a shifted last tile
                                  ((7, 11, 6, 36), (0, 2)),             # axes 1 and 3 extrapolated: ghosts of the plane axes (PG)
                                  ((9, 5, 13, 34), None),               # nothing periodic; axis 1 and axis 3 exactly one tile
                                  ((6, 12, 8, 70), (1, 3))])            # axis 0 extrapolated (ghost planes), axis 2 extrapolated
def test_pair4_kernel_vs_fp64_oracle_and_direct(scheme, n, pd, monkeypatch):
    g, og = pendulum_grid(n, pd)
    rng = np.random.default_rng(5)
    data = O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2]) + 0.02 * rng.standard_normal(n)
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, 0., data.reshape(-1, 1))
    scale = float(np.max(np.abs(yo)))
    y32 = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=torch.float32)
    got = {}
    want = {"pair4": b"fused_pair4_kernel", "pair": b"fused_pair_kernel", "single": b"fused_substep_kernel", "direct": b"direct_substep_kernel"}
    for name in ("pair4", "pair", "single", "direct"):
        monkeypatch.setenv("HJ_FORCE_DIRECT", "1" if name == "direct" else "0")
        monkeypatch.setenv("HJ_PAIR", "0" if name == "single" else "2")
        monkeypatch.setenv("HJ_PAIR4", "1" if name == "pair4" else "0")
        g.__dict__.pop("_hj_device", None)
        yd, sb, _ = L.termLaxFriedrichs(0., y32, sdata(g, L.DoublePendulum4D(g, 1.0), DERIV[scheme]))
        assert _last_kernel(g) == want[name], (name, _last_kernel(g))
        assert abs(sb - sbo) <= 1e-5 * sbo, (name, sb, sbo)
        got[name] = yd.cpu().numpy().astype(np.float64)
    g.__dict__.pop("_hj_device", None)
    rel = np.abs(got["pair4"] - yo) / scale
    if scheme.startswith("WENO"):
        assert rel.max() <= 1e-4, rel.max()
    else:       # fp32 ENO2: a selector whose margin is below fp32 rounding may go the other way (masked comparison)
        assert np.mean(rel > 1e-4) <= 2e-3 and rel.max() <= 0.2, (float(np.mean(rel > 1e-4)), rel.max())
    for name in ("pair", "single", "direct"):
        assert np.array_equal(got["pair4"], got[name]), (name, float(np.max(np.abs(got["pair4"] - got[name]))))


@pytest.mark.parametrize("sel", [0, 1, 2])
@pytest.mark.parametrize("pd", [(0, 1, 2, 3), (2,), None])
def test_pair4_every_built_tile_bitwise_vs_direct(sel, pd, monkeypatch):
    """Each compile-time tile of HJ_TILE4 (hj_inst.hip: 5x6x66 in 512 threads, 3x5x66 and 5x6x34 in 256) on a grid all of them
    fit, periodic / mixed / extrapolated axes: the term equals the direct kernel's bit for bit and the oracle's to 1e-4."""
    n = (6, 7, 9, 72)
    g, og = pendulum_grid(n, pd)
    rng = np.random.default_rng(11)
    data = O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2]) + 0.02 * rng.standard_normal(n)
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), "WENO5_ASSHIPPED", 0., data.reshape(-1, 1))
    y32 = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=torch.float32)
    got = {}
    for name in ("pair4", "direct"):
        monkeypatch.setenv("HJ_FORCE_DIRECT", "1" if name == "direct" else "0")
        monkeypatch.setenv("HJ_PAIR", "2")
        monkeypatch.setenv("HJ_TILE4_SEL", str(sel))
        g.__dict__.pop("_hj_device", None)
        yd, sb, _ = L.termLaxFriedrichs(0., y32, sdata(g, L.DoublePendulum4D(g, 1.0), DERIV["WENO5_ASSHIPPED"]))
        assert _last_kernel(g) == (b"fused_pair4_kernel" if name == "pair4" else b"direct_substep_kernel")
        if name == "pair4":
            dg = g.__dict__["_hj_device"]
            dg = dg[next(iter(dg))] if isinstance(dg, dict) else dg
            e = (C.c_int * 4)()
            _ffi.check(dg.lib.hj_last_tile(dg.ctx, e))
            assert tuple(e)[1:] == {0: (5, 6, 66), 1: (3, 5, 66), 2: (5, 6, 34)}[sel], tuple(e)
        assert abs(sb - sbo) <= 1e-5 * sbo
        got[name] = yd.cpu().numpy().astype(np.float64)
    g.__dict__.pop("_hj_device", None)
    assert np.array_equal(got["pair4"], got["direct"]), float(np.max(np.abs(got["pair4"] - got["direct"])))
    assert (np.abs(got["pair4"] - yo) / float(np.max(np.abs(yo)))).max() <= 1e-4


def test_pair4_rk3_steps_clamp_ranges_and_bound(monkeypatch):
    """One RK3 step through hj_rk_substep with the new kernel: the three stage instantiations (Euler / with y0 / general with
    the termRestrictUpdate clamp) equal the direct kernel bit for bit; a stage computed as three plane ranges equals one launch;
    the in-kernel CFL maxima equal the definition."""
    n = (37, 10, 12, 68)
    g, og = pendulum_grid(n)
    rng = np.random.default_rng(7)
    d0 = torch.as_tensor(O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2]) + 0.02 * rng.standard_normal(n),
                         device="cuda", dtype=torch.float32).contiguous()
    par = [1.0, 0., 0., 0.]
    dt = 2e-3
    outs = {}
    for name, force in (("pair4", "0"), ("direct", "1")):
        monkeypatch.setenv("HJ_FORCE_DIRECT", force)
        monkeypatch.setenv("HJ_PAIR", "2")
        dg = DeviceGrid(g, "float32")
        dg.bind_stream()
        a, b, c, r = dg.empty(), dg.empty(), dg.empty(), dg.empty()
        _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_EULER, dt, d0, None, a)
        _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_RK3_HALF, dt, a, d0, b)
        _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_RK3_FULL, dt, b, d0, c)
        assert dg.lib.hj_last_kernel(dg.ctx) == (b"fused_pair4_kernel" if force == "0" else b"direct_substep_kernel")
        _substep_rs(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_RK3_FULL, dt, b, d0, r, -1)
        if force == "0":
            c2 = torch.zeros_like(c)
            for k, (p0, p1) in enumerate([(0, 5), (5, 30), (30, n[0])]):
                _substep(dg, "WENO5_ASSHIPPED", _ffi.HAM_DOUBLE_PENDULUM, par, _ffi.STAGE_RK3_FULL, dt, b, d0, c2, p0, p1, slot=4 + k)
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
        assert torch.equal(outs["pair4"][k], outs["direct"][k]), (k, float((outs["pair4"][k] - outs["direct"][k]).abs().max()))
    assert float((outs["pair4"][2] - outs["pair4"][3]).abs().max()) > 0      # the clamp did something


def test_pair4_long_axis0_chunks_the_row_table():
    """An axis 0 longer than one LDS row table holds: several chunks per tile column, each with its own table."""
    n = (300, 5, 6, 34)
    g, og = pendulum_grid(n)
    data = O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2])
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), "WENO5_ASSHIPPED", 0., data.reshape(-1, 1))
    import os
    os.environ["HJ_PAIR"] = "2"
    try:
        g.__dict__.pop("_hj_device", None)
        y32 = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=torch.float32)
        yd, sb, _ = L.termLaxFriedrichs(0., y32, sdata(g, L.DoublePendulum4D(g, 1.0), DERIV["WENO5_ASSHIPPED"]))
        assert _last_kernel(g) == b"fused_pair4_kernel"
    finally:
        os.environ.pop("HJ_PAIR", None)
        g.__dict__.pop("_hj_device", None)
    rel = np.abs(yd.cpu().numpy().astype(np.float64) - yo) / float(np.max(np.abs(yo)))
    assert rel.max() <= 1e-4, rel.max()
    assert abs(sb - sbo) <= 1e-5 * sbo


# ------------------------------------------------------------------------------ alpha depending on the costate range (general GLF)
from test_gpu_parity import mk, close, SCHEMES  # noqa: E402


def _is_t(a):
    return type(a).__module__.startswith("torch")


class BurgersDrift(object):
    """H = sum_d p_d^2 / 2 + c x_0 p_1; alpha_d = max(|derivMin_d|, |derivMax_d|) (+ |c x_0| for d = 1): the partial of a convex H
    bounded over the costate RANGE, which is what the reference's protocol hands to partialFunc (artificial_diss_glf.py:80-99).
    Array callbacks as the reference would write them (NumPy or torch)."""

    def __init__(self, grid, c):
        self.grid, self.c = grid, c

    def hamiltonian(self, t, data, p, sd=None):
        x0 = np.asarray(self.grid.xs[0])
        if _is_t(p[0]):
            x0 = torch.as_tensor(x0, device=p[0].device)
        h = 0
        for d in range(len(p)):
            h = h + 0.5 * p[d] * p[d]
        return h + self.c * x0 * p[1]

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        a = max(abs(float(dmin[dim])), abs(float(dmax[dim])))
        if dim != 1:
            return a
        x0 = np.asarray(self.grid.xs[0])
        arr = a + np.abs(self.c * x0)
        return torch.as_tensor(np.ascontiguousarray(arr), device=data.device) if _is_t(data) else arr


def _burgers_src(dim):
    s = "H = par[0] * x[0] * p[1];\n"
    for d in range(dim):
        s += "H += 0.5 * p[%d] * p[%d];  alpha[%d] = fmax(fabs(dmin[%d]), fabs(dmax[%d]));\n" % (d, d, d, d, d)
    return s + "alpha[1] += fabs(par[0] * x[0]);\n"


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((40, 38), 1), ((23, 21, 26), 2), ((19, 24, 22), None)])
def test_range_dependent_alpha_term_vs_oracle_and_split(scheme, n, pd):
    """termLaxFriedrichs with a partialFunc that uses derivMin / derivMax: the fused two-launch path (range pass + substep with
    the in-kernel max(alpha)) against the oracle's artificial_dissipation_glf semantics and against the split path (the same
    object's Python callbacks between the library's derivative and dissipation kernels)."""
    dim = len(n)
    g, og = mk([-1.0] * dim, [1.0 - (2.0 / n[d] if pd == d else 0) for d in range(dim)], n, pd)
    rng = np.random.default_rng(3)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[dim - 1]) + 0.02 * rng.standard_normal(n)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    sys_ = BurgersDrift(g, 0.7)
    split, sb_s, _ = L.termLaxFriedrichs(0., y, sdata(g, sys_, DERIV[scheme]))            # not registered yet: the split path
    reg = L.register_native_hamiltonian("burgers_drift_%dd" % dim, dim, _burgers_src(dim), nparams=1)
    assert reg.uses_range
    reg.attach(sys_, params=lambda o: [o.c])
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sdata(g, sys_, DERIV[scheme]))
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    yo, sbo = O.term_lax_friedrichs(og, BurgersDrift(og, 0.7), scheme, 0., d0.reshape(-1, 1))
    close(fused.cpu().numpy(), yo, 1e-11, what="fused vs oracle")
    close(fused.cpu().numpy(), split.cpu().numpy(), 1e-11, what="fused vs split")
    assert abs(sb_f - sbo) <= 1e-12 * sbo and abs(sb_s - sbo) <= 1e-12 * sbo, (sb_f, sb_s, sbo)


@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "WENO5", "ENO3"])
@pytest.mark.parametrize("order", [1, 2, 3])
def test_range_dependent_alpha_through_odecfl_vs_oracle(scheme, order):
    """odeCFL1/2/3 with a data-dependent alpha: deltaT comes from the FIRST stage's reduced stepBound (ode_cfl_3.py:142), the later
    stages reduce their own ranges; four steps against the oracle's integrators within 1e-11, t to 1e-13."""
    n = (26, 22, 24)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    sys_ = L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1)(
        g, [0.7], hamiltonian=lambda s, t, data, p, sd: BurgersDrift(g, 0.7).hamiltonian(t, data, p, sd),
        dissipation=lambda s, t, data, dmin, dmax, sd, dim: BurgersDrift(g, 0.7).dissipation(t, data, dmin, dmax, sd, dim))
    sd = sdata(g, sys_, DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    ode = {1: L.odeCFL1, 2: L.odeCFL2, 3: L.odeCFL3}[order]
    oode = {1: O.ode_cfl_1, 2: O.ode_cfl_2, 3: O.ode_cfl_3}[order]
    term = lambda tt, yy: O.term_lax_friedrichs(og, BurgersDrift(og, 0.7), scheme, tt, yy)  # noqa: E731
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    yo, t, to = d0.reshape(-1, 1), 0., 0.
    for _ in range(4):
        t, y, _ = ode(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        to, yo = oode(term, [to, 10.], yo, 0.8, single_step=True)
    assert _last_kernel(g).endswith(b"(hipRTC)")
    assert abs(t - to) <= 1e-13 * to, (t, to)
    if scheme == "ENO3":
        diff = np.abs(y.cpu().numpy() - yo)
        assert np.mean(diff > 1e-11) <= 2e-3 and diff.max() <= 1e-3, (float(np.mean(diff > 1e-11)), diff.max())
    else:
        close(y.cpu().numpy(), yo, 1e-11, what="4 steps, order %d" % order)
    # a whole span in one call (several steps inside odeCFLn's loop) lands on the same state as stepping it
    t2, y2, _ = ode(L.termLaxFriedrichs, [0., float(t)], torch.as_tensor(d0.reshape(-1, 1), device="cuda"),
                    L.odeCFLset(L.Bundle(dict(factorCFL=.8))), sd)
    assert abs(t2 - t) <= 1e-13
    if scheme != "ENO3":
        close(y2.cpu().numpy(), y.cpu().numpy(), 1e-11, what="span vs single steps")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "ENO3"])
def test_runtime_hamiltonian_4d_and_fp32_vs_oracle(scheme, dtype):
    """Run-time Hamiltonians beyond fp64 3-D (round 5): the double pendulum of BASELINE C5 written as a device expression on a 4-D
    grid, fp64 (one-cell-per-lane kernel) and fp32 (pair kernel for the light stencil), against the fp64 oracle."""
    n = (9, 11, 13, 12)
    g, og = pendulum_grid(n)
    src = """
        const T s1 = sin(x[0]), c1 = cos(x[0]), s2 = sin(x[2]), c2 = cos(x[2]);
        const T sd = s2 * c1 - c2 * s1, cd = c2 * c1 + s2 * s1, den = T(2) - cd * cd;
        const T f1 = (x[1] * x[1] * sd * cd + T(9.8) * s2 * cd + x[3] * x[3] * sd - T(19.6) * s1) / den;
        const T f3 = (-x[3] * x[3] * sd * cd + T(19.6) * s1 * cd - T(2) * x[1] * x[1] * sd - T(19.6) * s2) / den;
        H = p[0] * x[1] + p[1] * f1 + p[2] * x[3] + p[3] * f3 + par[0] * (fabs(p[1]) + fabs(p[3]));
        alpha[0] = fabs(x[1]); alpha[1] = fabs(f1) + par[0]; alpha[2] = fabs(x[3]); alpha[3] = fabs(f3) + par[0];
    """
    user = L.register_native_hamiltonian("pendulum_rt", 4, src, nparams=1)(g, [1.0])
    rng = np.random.default_rng(2)
    data = O.shape_sphere(og, None, 1.5) + 0.05 * np.sin(2 * og.xs[0]) * np.cos(og.xs[2]) + 0.02 * rng.standard_normal(n)
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, 0., data.reshape(-1, 1))
    y = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=getattr(torch, dtype))
    yd, sb, _ = L.termLaxFriedrichs(0., y, sdata(g, user, DERIV[scheme]))
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    assert yd.dtype == y.dtype
    rel = np.abs(yd.cpu().numpy().astype(np.float64) - yo) / float(np.max(np.abs(yo)))
    if dtype == "float64":
        assert rel.max() <= 1e-11 and abs(sb - sbo) <= 1e-12 * sbo, (rel.max(), sb, sbo)
    else:
        assert abs(sb - sbo) <= 1e-5 * sbo
        if scheme.startswith("WENO"):
            assert rel.max() <= 1e-4, rel.max()
        else:
            assert np.mean(rel > 1e-4) <= 2e-3 and rel.max() <= 0.2
    # and through an integrator step
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t1, y1, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y, op, sdata(g, user, DERIV[scheme]))
    term = lambda tt, yy: O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, tt, yy)  # noqa: E731
    to, y1o = O.ode_cfl_3(term, [0., 10.], data.reshape(-1, 1), 0.8, single_step=True)
    assert abs(t1 - to) <= (1e-12 if dtype == "float64" else 1e-5) * to
    rel = np.abs(y1.cpu().numpy().astype(np.float64) - y1o) / float(np.max(np.abs(y1o)))
    assert (rel.max() <= 1e-11) if dtype == "float64" else (np.mean(rel > 1e-4) <= 2e-3)


@pytest.mark.parametrize("world,periodic0", [(2, True), (3, False)])
def test_range_dependent_alpha_virtual_ranks_equal_undivided(world, periodic0):
    """dist.SlabIntegrator(dynamic=True) + HipSlabBackend on one card (threads as ranks): before every substep the slabs' costate
    ranges are reduced (hj_range_pass) and all-reduced, the launches read the reduced range (hj_ctx_set_range_source), deltaT comes
    from the all-reduced max(alpha) (hj_range_alpha_max).  Two RK3 steps equal the undivided grid through odeCFL3."""
    import threading
    from test_gpu_round4 import ThreadRing
    from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
    n = (30, 22, 24)
    pd = [0, 2] if periodic0 else 2
    g, og = mk([-1.0] * 3, [1.0 - (2.0 / n[0] if periodic0 else 0.0), 1.0, 1.0 - 2.0 / n[2]], n, pd)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    reg = L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1)
    sys_ = reg(g, [0.7])
    full = torch.as_tensor(d0, device="cuda")
    sd = sdata(g, sys_, DERIV["WENO5_ASSHIPPED"])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y, t_ref = full.reshape(-1, 1), 0.
    for _ in range(2):
        t_ref, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t_ref, 10.], y, op, sd)
    ref = y.reshape(n)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    tr = ThreadRing(world)
    out, errs = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                slab = SlabDecomposition(n[0], world, rank, periodic0, self_exchange=periodic0)
                be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, [0.7], "float64")
                integ = SlabIntegrator(slab, be, dxs, 3, 0.8, exchanger=tr.exchanger(slab), allreduce_max=tr.allreduce_max(rank), dynamic=True)
                integ.set_state(full[slab.begin:slab.end])
                t = 0.
                for _ in range(2):
                    t, dt = integ.step(t)
                be.sync()
                out[rank] = (slab.begin, slab.end, t, integ.state().clone())
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
    for r in range(world):
        b, e, t, ys = out[r]
        assert abs(t - t_ref) <= 1e-14 * t_ref, (t, t_ref)
        assert float((ys - ref[b:e]).abs().max()) <= 1e-13, (r, float((ys - ref[b:e]).abs().max()))


@pytest.mark.parametrize("kind,scheme,world,periodic0", [("llf", "WENO5_ASSHIPPED", 2, True), ("lllf", "WENO5_ASSHIPPED", 3, False),
                                                         ("llf", "WENO5", 3, False), ("lllf", "ENO2", 2, True)])
def test_local_lax_friedrichs_range_reading_virtual_ranks_equal_undivided(kind, scheme, world, periodic0):
    """dist.SlabIntegrator(dynamic=True, diss='llf' | 'lllf') + HipSlabBackend: the local variants of a range-reading Hamiltonian on a
    decomposed grid.  LLF all-reduces the range before every substep as GLF does, LLLF exchanges no range at all; deltaT comes from ONE
    all-reduced scalar, max over the slabs of max_x sum_i alpha_i(x) / dx_i (hj_bound_pass).  Two RK3 steps equal the undivided grid
    through odeCFL3 with artificialDissipationLLF / LLLF."""
    import threading
    from test_gpu_round4 import ThreadRing
    from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
    n = (30, 22, 24)
    pd = [0, 2] if periodic0 else 2
    g, og = mk([-1.0] * 3, [1.0 - (2.0 / n[0] if periodic0 else 0.0), 1.0, 1.0 - 2.0 / n[2]], n, pd)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    reg = L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1)
    sys_ = BurgersDriftLocal(g, 0.7)
    reg.attach(sys_, params=lambda o: [o.c])
    full = torch.as_tensor(d0, device="cuda")
    diss = {"llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=diss, CoStateCalc=DERIV[scheme]))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y, t_ref = full.reshape(-1, 1), 0.
    for _ in range(2):
        t_ref, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t_ref, 10.], y, op, sd)
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    ref = y.reshape(n)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    tr = ThreadRing(world)
    out, errs = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                slab = SlabDecomposition(n[0], world, rank, periodic0, self_exchange=periodic0)
                be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS[scheme], reg.ham_id, [0.7], "float64")
                integ = SlabIntegrator(slab, be, dxs, 3, 0.8, exchanger=tr.exchanger(slab), allreduce_max=tr.allreduce_max(rank),
                                       needs_eps=(scheme == "WENO5"), dynamic=True, diss=kind)
                integ.set_state(full[slab.begin:slab.end])
                t = 0.
                for _ in range(2):
                    t, dt = integ.step(t)
                be.sync()
                out[rank] = (slab.begin, slab.end, t, integ.state().clone())
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
    tol = 1e-3 if scheme.startswith("ENO") else 1e-12
    for r in range(world):
        b, e, t, ys = out[r]
        assert abs(t - t_ref) <= 1e-13 * t_ref, (t, t_ref)
        diff = (ys - ref[b:e]).abs()
        assert float(diff.max()) <= tol and float((diff > 1e-12).double().mean()) <= 2e-3, (r, float(diff.max()))


def test_bound_pass_needs_a_local_kind_and_a_range_reading_hamiltonian():
    n = (24, 20, 22)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0], n, None)
    dg = DeviceGrid(g, "float64")
    reg = L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1)
    y = torch.as_tensor(O.shape_sphere(og, None, 0.5), device="cuda")
    sb = C.c_double()
    par = _ffi.darr([0.7])
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, _ffi.DISS_GLF))
    assert dg.lib.hj_bound_pass(dg.ctx, sid, reg.ham_id, par, C.c_void_p(y.data_ptr()), C.byref(sb)) == -4      # HJ_ESTATE
    _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, _ffi.DISS_LLF))
    assert dg.lib.hj_bound_pass(dg.ctx, sid, _ffi.HAM_DUBINS_REL, _ffi.darr([1., 1., 1., 2.]), C.c_void_p(y.data_ptr()), C.byref(sb)) == -1   # HJ_EINVAL
    # a single ctx without an external range: LLF runs its own range pass; the bound equals termLaxFriedrichs's
    _ffi.check(dg.lib.hj_bound_pass(dg.ctx, sid, reg.ham_id, par, C.c_void_p(y.data_ptr()), C.byref(sb)))
    _, sbo = O.term_lax_friedrichs(og, BurgersDriftLocal(og, 0.7), "WENO5_ASSHIPPED", 0., O.shape_sphere(og, None, 0.5).reshape(-1, 1), diss="llf")
    assert abs(sb.value - sbo) <= 1e-12 * sbo, (sb.value, sbo)
    _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, _ffi.DISS_GLF))


def test_run_time_kernels_first_compiled_by_several_threads_at_once():
    """Kernels of a run-time Hamiltonian are compiled at their first launch.  Six host threads (virtual ranks, each with a context of its own)
    meeting in that FIRST launch must all get complete kernels: the per-Hamiltonian kernel tables are guarded (hj_rtc.hip, g_rtc_mu; round 5 --
    tests/fuzz_slabs.py met an 'invalid argument' launch when eight threads raced through the first compile).  The expression is new to the
    process (its own name and constant), the undivided reference runs AFTER the threads."""
    import threading
    from test_gpu_round4 import ThreadRing
    from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
    n, world = (48, 20, 22), 6
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    src = _burgers_src(3).replace("0.5 * p[0] * p[0]", "0.501 * p[0] * p[0]")
    reg = L.register_native_hamiltonian("burgers_drift_race_3d", 3, src, nparams=1)
    full = torch.as_tensor(d0, device="cuda")
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    tr = ThreadRing(world)
    out, errs = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                slab = SlabDecomposition(n[0], world, rank, False)
                be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, [0.7], "float64")
                integ = SlabIntegrator(slab, be, dxs, 2, 0.8, exchanger=tr.exchanger(slab), allreduce_max=tr.allreduce_max(rank), dynamic=True)
                integ.set_state(full[slab.begin:slab.end])
                t, dt = integ.step(0.)
                be.sync()
                out[rank] = (slab.begin, slab.end, t, integ.state().clone())
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
    dg = DeviceGrid(g, "float64")
    cur, nxt, w0, w1 = full.clone(), torch.empty_like(full), torch.empty_like(full), torch.empty_like(full)
    tout, dtout = C.c_double(), C.c_double()
    _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, _ffi.DISS_GLF))
    _ffi.check(dg.lib.hj_rk_step(dg.ctx, 2, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, _ffi.darr([0.7]), 0., 1e9, 0.8, 1e300, 0,
                                 dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
    torch.cuda.synchronize()
    for r in range(world):
        b, e, t, ys = out[r]
        assert abs(t - tout.value) <= 1e-14 * tout.value
        assert float((ys - nxt[b:e]).abs().max()) <= 1e-13


# ------------------------------------------------------------------------------ alpha_i that reads the range of ANOTHER dimension
class CoupledBurgers(object):
    """H = sum_d p_d^2 / 2 + c p_0 p_1:  dH/dp_0 = p_0 + c p_1, so alpha_0 = max|p_0| + |c| max|p_1| and alpha_1 = max|p_1| + |c| max|p_0| over the
    ranges handed in -- the case in which the three Lax-Friedrichs variants really differ: GLF takes both maxima over the grid, LLF the node's own
    range in dimension i and the GRID's in the other (diss_local_laxfried.py:108-111), LLLF the node's own in both (diss_localsq_laxfried.py:87-90).
    The entries of derivMin / derivMax arrive as scalars, arrays, or a mixture (NumPy for the oracle, torch on the split path)."""

    def __init__(self, grid, c):
        self.grid, self.c = grid, c

    def hamiltonian(self, t, data, p, sd=None):
        h = self.c * p[0] * p[1]
        for d in range(len(p)):
            h = h + 0.5 * p[d] * p[d]
        return h

    @staticmethod
    def _amax(lo, hi, like):
        if _is_t(lo) or _is_t(hi):
            lo = lo if _is_t(lo) else torch.as_tensor(float(lo), device=like.device, dtype=like.dtype)
            hi = hi if _is_t(hi) else torch.as_tensor(float(hi), device=like.device, dtype=like.dtype)
            return torch.maximum(lo.abs(), hi.abs())
        return np.maximum(np.abs(lo), np.abs(hi))

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        a = self._amax(dmin[dim], dmax[dim], data)
        if dim > 1:
            return a
        other = 1 - dim
        return a + abs(self.c) * self._amax(dmin[other], dmax[other], data)


def _coupled_src(dim):
    s = "H = par[0] * p[0] * p[1];\n"
    for d in range(dim):
        s += "H += 0.5 * p[%d] * p[%d];  alpha[%d] = fmax(fabs(dmin[%d]), fabs(dmax[%d]));\n" % (d, d, d, d, d)
    s += "alpha[0] += fabs(par[0]) * fmax(fabs(dmin[1]), fabs(dmax[1]));\n"
    return s + "alpha[1] += fabs(par[0]) * fmax(fabs(dmin[0]), fabs(dmax[0]));\n"


@pytest.mark.parametrize("scheme", ["ENO2", "WENO5_ASSHIPPED", "WENO5"])
@pytest.mark.parametrize("n,pd", [((44, 37), 1), ((21, 23, 26), 2), ((9, 8, 10, 11), None)])
def test_cross_dimension_alpha_under_glf_llf_lllf_fused_vs_oracle_and_split(scheme, n, pd):
    """alpha_0 reads the range of dimension 1 and vice versa: under LLF the OTHER dimension's range must be the grid's, under LLLF the node's own --
    the three variants give three different terms and bounds (asserted), each equal to the oracle's and to the split path (the same object's
    callbacks on device arrays).  BurgersDrift above cannot tell LLF from LLLF (its alpha_i reads dimension i only)."""
    dim = len(n)
    g, og = mk([-1.0] * dim, [1.0 - (2.0 / n[d] if pd == d else 0) for d in range(dim)], n, pd)
    rng = np.random.default_rng(7)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[dim - 1]) + 0.02 * rng.standard_normal(n)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    diss_of = {"glf": L.artificialDissipationGLF, "llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}
    sys_ = CoupledBurgers(g, 0.6)

    def sd_of(kind):
        return L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=diss_of[kind], CoStateCalc=DERIV[scheme]))
    split = {k: L.termLaxFriedrichs(0., y, sd_of(k))[:2] for k in diss_of}                 # not attached yet: the split path
    L.register_native_hamiltonian("coupled_burgers_%dd" % dim, dim, _coupled_src(dim), nparams=1).attach(sys_, params=lambda o: [o.c])
    fused, bounds = {}, {}
    for kind in diss_of:
        f, sb_f, _ = L.termLaxFriedrichs(0., y, sd_of(kind))
        assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
        yo, sbo = O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), scheme, 0., d0.reshape(-1, 1), diss=kind)
        close(f.cpu().numpy(), yo, 1e-11, what="fused vs oracle, " + kind)
        close(split[kind][0].cpu().numpy(), yo, 1e-11, what="split vs oracle, " + kind)
        assert abs(sb_f - sbo) <= 1e-12 * sbo and abs(split[kind][1] - sbo) <= 1e-12 * sbo, (kind, sb_f, split[kind][1], sbo)
        fused[kind], bounds[kind] = f, sb_f
    # the three variants are three different terms here
    assert float((fused["llf"] - fused["lllf"]).abs().max()) > 1e-6 and float((fused["glf"] - fused["llf"]).abs().max()) > 1e-6
    assert bounds["glf"] < bounds["llf"] < bounds["lllf"]
    # ... and through the integrators (deltaT from the variant's own bound)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    for kind in ("llf", "lllf"):
        yy, t, yo2, to = y, 0., d0.reshape(-1, 1), 0.
        for _ in range(2):
            t, yy, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], yy, op, sd_of(kind))
            to, yo2 = O.ode_cfl_3(lambda tt, v: O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), scheme, tt, v, diss=kind), [to, 10.], yo2, 0.8, single_step=True)
        assert abs(t - to) <= 1e-12 * to, (kind, t, to)
        if scheme.startswith("ENO"):
            diff = np.abs(yy.cpu().numpy() - yo2)
            assert np.mean(diff > 1e-11) <= 2e-3 and diff.max() <= 1e-3
        else:
            close(yy.cpu().numpy(), yo2, 1e-11, what="2 steps, " + kind)


@pytest.mark.parametrize("kind,world,periodic0", [("llf", 3, False), ("llf", 2, True), ("lllf", 3, False), ("glf", 2, True)])
def test_cross_dimension_alpha_virtual_ranks_equal_undivided(kind, world, periodic0):
    """The same Hamiltonian across slabs: under LLF a rank whose slab does not hold the grid's extreme costates of dimension 1 gets them only from the
    all-reduced range (hj_ctx_set_range_source) -- with a rank-local range alpha_0 would come out too small there.  Two RK3 steps, every rank equal to
    the undivided grid through odeCFL3."""
    import threading
    from test_gpu_round4 import ThreadRing
    from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
    n = (33, 22, 24)
    pd = [0, 2] if periodic0 else 2
    g, og = mk([-1.0] * 3, [1.0 - (2.0 / n[0] if periodic0 else 0.0), 1.0, 1.0 - 2.0 / n[2]], n, pd)
    # steep in dimension 1 near x0 = -1 only: the first slab holds the extreme of dimension 1, the others do not
    d0 = O.shape_sphere(og, None, 0.5) + 0.3 * np.exp(-8 * (og.xs[0] + 1.0) ** 2) * np.sin(5 * og.xs[1]) + 0.1 * np.cos(2 * og.xs[2])
    reg = L.register_native_hamiltonian("coupled_burgers_3d", 3, _coupled_src(3), nparams=1)
    sys_ = CoupledBurgers(g, 0.6)
    reg.attach(sys_, params=lambda o: [o.c])
    full = torch.as_tensor(d0, device="cuda")
    diss = {"glf": L.artificialDissipationGLF, "llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=diss, CoStateCalc=DERIV["WENO5_ASSHIPPED"]))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y, t_ref = full.reshape(-1, 1), 0.
    for _ in range(2):
        t_ref, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t_ref, 10.], y, op, sd)
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    ref = y.reshape(n)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    tr = ThreadRing(world)
    out, errs = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                slab = SlabDecomposition(n[0], world, rank, periodic0, self_exchange=periodic0)
                be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, [0.6], "float64")
                integ = SlabIntegrator(slab, be, dxs, 3, 0.8, exchanger=tr.exchanger(slab), allreduce_max=tr.allreduce_max(rank), dynamic=True, diss=kind)
                integ.set_state(full[slab.begin:slab.end])
                t = 0.
                for _ in range(2):
                    t, dt = integ.step(t)
                be.sync()
                out[rank] = (slab.begin, slab.end, t, integ.state().clone())
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
    for r in range(world):
        b, e, t, ys = out[r]
        assert abs(t - t_ref) <= 1e-13 * t_ref, (t, t_ref)
        assert float((ys - ref[b:e]).abs().max()) <= 1e-12, (r, float((ys - ref[b:e]).abs().max()))


@pytest.mark.parametrize("kind", ["glf", "llf", "lllf"])
@pytest.mark.parametrize("n,pd", [((40, 36), 1), ((21, 23, 26), 2), ((9, 8, 10, 11), None)])
def test_range_reading_hamiltonian_in_fp32_vs_fp64_oracle(kind, n, pd):
    """The range path in single precision (run-time kernels are built per dtype; the range keys stay fp64 keys of fp32 costates): the cross-dimension
    Hamiltonian under the three Lax-Friedrichs variants, term and one odeCFL3 step, against the fp64 oracle at fp32 tolerances."""
    dim = len(n)
    g, og = mk([-1.0] * dim, [1.0 - (2.0 / n[d] if pd == d else 0) for d in range(dim)], n, pd)
    d0 = (O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[dim - 1])).astype(np.float32)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    assert y.dtype == torch.float32
    diss = {"glf": L.artificialDissipationGLF, "llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sys_ = CoupledBurgers(g, 0.6)
    L.register_native_hamiltonian("coupled_burgers_%dd" % dim, dim, _coupled_src(dim), nparams=1).attach(sys_, params=lambda o: [o.c])
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=diss, CoStateCalc=DERIV["WENO5_ASSHIPPED"]))
    f, sb, _ = L.termLaxFriedrichs(0., y, sd)
    assert _last_kernel(g).endswith(b"(hipRTC)") and f.dtype == torch.float32, _last_kernel(g)
    d64 = d0.astype(np.float64).reshape(-1, 1)
    yo, sbo = O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), "WENO5_ASSHIPPED", 0., d64, diss=kind)
    scale = float(np.abs(yo).max())
    assert float(np.abs(f.cpu().numpy().astype(np.float64) - yo).max()) <= 2e-4 * scale
    assert abs(sb - sbo) <= 1e-4 * sbo, (sb, sbo)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t, y1, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y, op, sd)
    to, yo1 = O.ode_cfl_3(lambda tt, v: O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), "WENO5_ASSHIPPED", tt, v, diss=kind), [0., 10.], d64, 0.8, single_step=True)
    assert y1.dtype == torch.float32 and abs(t - to) <= 1e-4 * to, (t, to)
    assert float(np.abs(y1.cpu().numpy().astype(np.float64) - yo1).max()) <= 1e-5 * max(1.0, float(np.abs(yo1).max()))


@pytest.mark.parametrize("kind", ["glf", "llf", "lllf"])
@pytest.mark.parametrize("positive", [0, 1])
def test_restrict_update_around_a_range_reading_hamiltonian(kind, positive):
    """termRestrictUpdate(termLaxFriedrichs) with the cross-dimension Hamiltonian: the flag-carrying instantiation of the run-time kernels (MODE 0: the
    clamp of ydot) together with the range pass and the local evaluation; term and two odeCFL2 steps against the oracle."""
    n = (22, 25, 24)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    y = torch.as_tensor(d0.reshape(-1), device="cuda")                 # (N,) vectors, as the air3D-style drivers call it
    diss = {"glf": L.artificialDissipationGLF, "llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sys_ = CoupledBurgers(g, 0.6)
    L.register_native_hamiltonian("coupled_burgers_3d", 3, _coupled_src(3), nparams=1).attach(sys_, params=lambda o: [o.c])
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=diss, CoStateCalc=DERIV["WENO5_ASSHIPPED"]))
    sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sd, positive=positive))
    inner = lambda tt, v: O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), "WENO5_ASSHIPPED", tt, v, diss=kind)  # noqa: E731
    oterm = O.term_restrict_update(inner, positive=bool(positive))
    f, sb, _ = L.termRestrictUpdate(0., y, sdr)
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    yo, sbo = oterm(0., d0.reshape(-1))
    close(f.cpu().numpy().reshape(-1), np.asarray(yo).reshape(-1), 1e-11, what="restricted term")
    assert abs(sb - sbo) <= 1e-12 * sbo
    assert (f >= 0).all() if positive else (f <= 0).all()
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    yy, t, yo2, to = y, 0., d0.reshape(-1), 0.
    for _ in range(2):
        t, yy, _ = L.odeCFL2(L.termRestrictUpdate, [t, 10.], yy, op, sdr)
        to, yo2 = O.ode_cfl_2(oterm, [to, 10.], yo2, 0.8, single_step=True)
    assert abs(t - to) <= 1e-12 * to
    close(yy.cpu().numpy().reshape(-1), np.asarray(yo2).reshape(-1), 1e-11, what="2 restricted RK2 steps")


@pytest.mark.parametrize("mode", ["minVOverTime", "maxVOverTime"])
def test_hjipde_solve_native_loop_with_a_cross_dimension_alpha_equals_the_step_loop_and_the_oracle(mode, monkeypatch):
    """HJIPDE_solve over two tau intervals with the cross-dimension Hamiltonian (the solver installs artificialDissipationGLF itself, as the
    reference does: hji_solver.py:433-434): the native loop (hj_rk_integrate: range pass, bound kernel, deltaT on the device, the min / max with the
    step's start folded into the last stage) against the Python step loop, and against an oracle loop written out here (ode_cfl_3 single steps + the
    reference's post-step min / max)."""
    n = (24, 22, 20)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    sys_ = CoupledBurgers(g, 0.6)
    L.register_native_hamiltonian("coupled_burgers_3d", 3, _coupled_src(3), nparams=1).attach(sys_, params=lambda o: [o.c])
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation))
    tau = [0., 0.004, 0.009]
    outs = []
    for stepwise in ("0", "1"):
        monkeypatch.setenv("HJ_HJIPDE_STEPWISE", stepwise)
        data, _, _ = L.HJIPDE_solve(d0, tau, sd, mode, L.Bundle(dict(quiet=True, keepLast=True)))
        outs.append(np.asarray(data))
    assert np.array_equal(outs[0], outs[1])
    term = lambda tt, v: O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), "WENO5_ASSHIPPED", tt, v)  # noqa: E731
    yo = d0.reshape(-1, 1)
    for i in range(1, len(tau)):
        t = tau[i - 1]
        while t < tau[i] - 1e-4:                      # hji_solver.py:185,536 (small = 1e-4)
            y_prev = yo
            t, yo = O.ode_cfl_3(term, [t, tau[i]], yo, 0.8, single_step=True)
            yo = np.minimum(yo, y_prev) if mode == "minVOverTime" else np.maximum(yo, y_prev)
    close(outs[0].reshape(-1), yo.reshape(-1), 1e-11, what="HJIPDE_solve vs oracle loop")


@pytest.mark.parametrize("kind", ["glf", "llf", "lllf"])
@pytest.mark.parametrize("order", [2, 3])
def test_multi_step_span_with_a_cross_dimension_alpha_vs_oracle(kind, order):
    """odeCFLn over a whole span (singleStep off: the native multi-step loop, hj_rk_integrate -> one dynamic step after the other, each with its
    range pass / bound pass and deltaT formed on the device, the last one shortened to land on tspan[1]) under each Lax-Friedrichs variant."""
    n = (24, 22, 20)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    diss = {"glf": L.artificialDissipationGLF, "llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sys_ = CoupledBurgers(g, 0.6)
    L.register_native_hamiltonian("coupled_burgers_3d", 3, _coupled_src(3), nparams=1).attach(sys_, params=lambda o: [o.c])
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=diss, CoStateCalc=DERIV["WENO5_ASSHIPPED"]))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8)))
    ode, oode = {2: (L.odeCFL2, O.ode_cfl_2), 3: (L.odeCFL3, O.ode_cfl_3)}[order]
    t, y, _ = ode(L.termLaxFriedrichs, [0., 0.011], torch.as_tensor(d0.reshape(-1, 1), device="cuda"), op, sd)
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    to, yo = oode(lambda tt, v: O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), "WENO5_ASSHIPPED", tt, v, diss=kind), [0., 0.011], d0.reshape(-1, 1), 0.8)
    t = float(np.asarray(t).ravel()[-1])
    to = float(np.asarray(to).ravel()[-1])
    assert abs(t - to) <= 1e-13 and abs(t - 0.011) <= 1e-12, (t, to)
    close(y.cpu().numpy().reshape(-1), np.asarray(yo).reshape(-1), 1e-11, what="multi-step span")


def test_strided_inputs_are_read_in_logical_order():
    """Callers hand over whatever layout they have: a strided view of a larger tensor, a column of a 2-column array, a Fortran-ordered NumPy array, an
    fp64 array for an fp64 problem stored with a byte offset.  The reference reads them through y.reshape(grid.shape) -- logical order -- and so must the
    drop-in (context.to_device makes a contiguous copy); results equal the contiguous call bit for bit, and the caller's array is not written."""
    from test_gpu_parity import dubins
    g, og = dubins([26, 24, 22])
    d0 = O.shape_cylinder(og, 2, None, .5) + 0.05 * np.sin(3 * og.xs[0])
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    ref_term = L.termLaxFriedrichs(0., y, sd)[0]
    t_ref, ref_step, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y, op, sd)
    wide = torch.zeros((y.numel(), 3), device="cuda", dtype=torch.float64)
    wide[:, 1] = y.reshape(-1)
    col = wide[:, 1:2]                                            # (N, 1) with stride 3
    assert not col.is_contiguous()
    every_other = torch.zeros(2 * y.numel(), device="cuda", dtype=torch.float64)
    every_other[::2] = y.reshape(-1)
    keep = wide.clone()
    for view in (col, every_other[::2].reshape(-1, 1)):
        assert torch.equal(L.termLaxFriedrichs(0., view, sd)[0].reshape(-1), ref_term.reshape(-1))
        t, y1, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], view, op, sd)
        assert t == t_ref and torch.equal(y1.reshape(-1), ref_step.reshape(-1))
    assert torch.equal(wide, keep)
    # NumPy: Fortran order, and a view into a larger buffer
    f = np.asfortranarray(d0)
    assert not f.flags.c_contiguous
    buf = np.zeros((d0.size, 2))
    buf[:, 0] = d0.reshape(-1)
    for arr in (f, buf[:, 0:1]):
        t, y1, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], arr.reshape(-1, 1) if arr.ndim == 3 else arr, op, sd)
        assert t == t_ref and np.array_equal(np.asarray(y1).reshape(-1), ref_step.cpu().numpy().reshape(-1))
    assert np.array_equal(buf[:, 0], d0.reshape(-1)) and np.array_equal(f, d0)


def test_native_slab_steppers_refuse_a_range_reading_hamiltonian():
    """hj_slab_rk_step / hj_slab_rk_step_deep on a slab with neighbours and a Hamiltonian whose alpha reads the costate range: every launch would reduce
    the range of its own planes -- a rank-local range, silently different from the undivided grid's.  Both refuse (dist.SlabIntegrator(dynamic=True)
    all-reduces the range); the deep stepper did not until late in round 5."""
    n = (40, 20, 22)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    reg = L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1)
    dg = DeviceGrid(g, "float64", None, (10, 30, True, True))
    bufs = [torch.zeros((20 + 18,) + n[1:], device="cuda", dtype=torch.float64) for _ in range(4)]
    ptr = [C.c_void_p(b[9:].data_ptr()) for b in bufs]
    par = _ffi.darr([0.7])
    for fn in (dg.lib.hj_slab_rk_step_deep, dg.lib.hj_slab_rk_step):
        rc = fn(dg.ctx, 3, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, par, 1e-3, 0, ptr[0], ptr[1], ptr[2], ptr[3])
        assert rc == -3, rc                                   # HJ_EUNSUPPORTED
        assert b"range" in dg.lib.hj_last_error()
    # ... and a plain substep / an LLF bound pass on such a slab before the all-reduced range has been set (dist.SlabIntegrator always sets it)
    rc = dg.lib.hj_rk_substep(dg.ctx, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, par, 0., _ffi.STAGE_EULER, 1e-3, 0, ptr[0], C.c_void_p(0), ptr[1], 0, 0, 20)
    assert rc == -4 and b"WHOLE grid" in dg.lib.hj_last_error(), (rc, dg.lib.hj_last_error())       # HJ_ESTATE
    _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, _ffi.DISS_LLF))
    sb = C.c_double()
    rc = dg.lib.hj_bound_pass(dg.ctx, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, par, ptr[0], C.byref(sb))
    assert rc == -4 and b"WHOLE grid" in dg.lib.hj_last_error(), (rc, dg.lib.hj_last_error())
    _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, _ffi.DISS_GLF))
    keys = torch.zeros(8, dtype=torch.int64, device="cuda")
    _ffi.check(dg.lib.hj_range_pass(dg.ctx, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, par, ptr[0], C.c_void_p(keys.data_ptr())))      # the pass itself runs
    _ffi.check(dg.lib.hj_ctx_set_range_source(dg.ctx, C.c_void_p(keys.data_ptr())))
    _ffi.check(dg.lib.hj_rk_substep(dg.ctx, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], reg.ham_id, par, 0., _ffi.STAGE_EULER, 1e-3, 0, ptr[0], C.c_void_p(0), ptr[1], 0, 0, 20))
    _ffi.check(dg.lib.hj_ctx_set_range_source(dg.ctx, C.c_void_p(0)))
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------ opt-in fast ENO arithmetic (set_eno_mode('fast'))
def _dilate(mask, r):
    """cells within r of a marked cell along any axis (box dilation: an upper bound of the domain of dependence of a substep)"""
    out = mask.copy()
    for ax in range(mask.ndim):
        acc = out.copy()
        for k in range(1, r + 1):
            acc |= np.roll(out, k, axis=ax) | np.roll(out, -k, axis=ax)
        out = acc
    return out


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3"])
def test_fast_eno_mode_masked_parity_with_the_reference_golden(scheme, golden):
    """set_eno_mode('fast'): ENO2 / ENO3 substeps in the lean arithmetic against the goldens of the UNMODIFIED reference on noisy data,
    by the rule of SURVEY 8(c): cells whose stencil selectors have a margin below 1e-12 (in any dimension, at any of the 15 substeps,
    together with everything their values can have reached since) are excluded -- at most 1e-4 of the grid -- and every other cell
    agrees within 1e-11 after five RK3 steps; t within 1e-13.  The default ('exact') stays bit for bit (test_eno_paths_bitwise...)."""
    from test_gpu_parity import dubins
    G = golden("ode.npz")
    g, og = dubins(G["dubn_data"].shape)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    sd = sdata(g, sys_, DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    # the oracle's own run (pinned to the goldens elsewhere), recording the selector margins of every substep
    taint = np.zeros(og.shape, dtype=bool)
    osys = O.DubinsRel(og, 1, 1)

    def term(tt, yy):
        nonlocal taint
        taint = _dilate(taint, 3)
        data = yy.reshape(og.shape)
        for d in range(3):
            taint |= O.eno_selector_margin(og, data, d, scheme) < 1e-12
        return O.term_lax_friedrichs(og, osys, scheme, tt, yy)
    yo, to = G["dubn_data"].reshape(-1, 1), 0.
    for _ in range(5):
        to, yo = O.ode_cfl_3(term, [to, 10.], yo, 0.8, single_step=True)
    assert np.max(np.abs(yo - G["rk3n_%s_y5" % scheme])) <= 1e-12      # the oracle is the reference here
    L.set_eno_mode('fast')
    try:
        y, t = torch.as_tensor(G["dubn_data"].reshape(-1, 1), device="cuda"), 0.
        for _ in range(5):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    finally:
        L.set_eno_mode('exact')
    assert abs(t - float(G["rk3n_%s_t5" % scheme])) <= 1e-13
    diff = np.abs(y.cpu().numpy().reshape(og.shape) - G["rk3n_%s_y5" % scheme].reshape(og.shape))
    excluded = float(np.mean(taint))
    assert excluded <= 1e-4, excluded
    assert float(diff[~taint].max()) <= 1e-11, float(diff[~taint].max())
    assert float(diff.max()) > 0                      # it IS another arithmetic (the exact mode gives 0)


def test_fast_eno3_double_integrator_c3_shape_masked_parity(golden):
    """The C3 system (double integrator, ENO3) in fast mode on the reference's noisy 2-D golden: same rule."""
    G = golden("ode.npz")
    g2, og2 = mk([-1, -1], [1, 1], [32, 32], None)
    sd2 = sdata(g2, L.DoubleIntegrator(g2, 1), L.upwindFirstENO3)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    taint = np.zeros(og2.shape, dtype=bool)
    osys = O.DoubleIntegrator(og2, 1)

    def term(tt, yy):
        nonlocal taint
        taint = _dilate(taint, 3)
        for d in range(2):
            taint |= O.eno_selector_margin(og2, yy.reshape(og2.shape), d, "ENO3") < 1e-12
        return O.term_lax_friedrichs(og2, osys, "ENO3", tt, yy)
    yo, to = G["din_data"].reshape(-1, 1), 0.
    for _ in range(5):
        to, yo = O.ode_cfl_3(term, [to, 10.], yo, 0.8, single_step=True)
    L.set_eno_mode('fast')
    try:
        y, t = torch.as_tensor(G["din_data"].reshape(-1, 1), device="cuda"), 0.
        for _ in range(5):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd2)
    finally:
        L.set_eno_mode('exact')
    diff = np.abs(y.cpu().numpy().reshape(og2.shape) - G["din_rk3_ENO3_y5"].reshape(og2.shape))
    assert float(np.mean(taint)) <= 1e-4 and float(diff[~taint].max()) <= 1e-11, (float(np.mean(taint)), float(diff.max()))


# ------------------------------------------------------------------------------ NumPy callers, ndarray-subclass mode (lazy.DeviceArray)
def test_numpy_loop_with_real_ndarray_results_equals_the_tensor_loop():
    """set_lazy("ndarray"): the driver loop of the reference (hji_solver.py:542) gets genuine np.ndarray subclass instances back
    (isinstance / np.save / pickle as with ode_cfl_3.py:241-272's returns), values present, and the next call consumes the device
    tensor behind them; results equal the tensor-in loop bit for bit."""
    from levelsetpy_amd import lazy
    from test_gpu_parity import dubins
    g, og = dubins([33, 31, 29])
    d0 = O.shape_cylinder(og, 2, None, .5) + 0.02 * np.random.default_rng(2).standard_normal(og.shape)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5)
    sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sd, positive=0))
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    old = lazy.LAZY
    lazy.set_lazy("ndarray")
    try:
        y_np = d0.flatten()
        y_t = torch.as_tensor(y_np, device="cuda")
        t1 = t2 = 0.
        for k in range(3):
            t1, y_np, _ = L.odeCFL3(L.termRestrictUpdate, [t1, 10.], y_np, op, sdr)
            t2, y_t, _ = L.odeCFL3(L.termRestrictUpdate, [t2, 10.], y_t, op, sdr)
            assert isinstance(y_np, np.ndarray) and type(y_np) is lazy.DeviceArray and y_np.device_tensor() is not None
            assert np.array_equal(y_np, y_t.cpu().numpy()) and not y_np.flags.writeable
            if k == 1:
                y_np = y_np.reshape(og.shape).reshape(-1)       # the drivers reshape between calls: still attached
                assert y_np.device_tensor() is not None
        assert t1 == t2
        y_np[0] = 7.0                                           # a write detaches; the next call uploads the modified values
        assert y_np.device_tensor() is None and y_np.flags.writeable
        t3, y3, _ = L.odeCFL3(L.termRestrictUpdate, [t1, 10.], y_np, op, sdr)
        y_t[0] = 7.0
        t4, y4, _ = L.odeCFL3(L.termRestrictUpdate, [t2, 10.], y_t, op, sdr)
        assert t3 == t4 and np.array_equal(y3, y4.cpu().numpy())
    finally:
        lazy.set_lazy(old)


def test_foreign_write_to_the_shared_ctx_is_noticed():
    """term._Plan.bind skips the hj_ctx_set_* calls when the state it last wrote is unchanged -- another user of the cached ctx
    (here: this test, through the C ABI) may have written since.  The library counts the writes (hj_ctx_state_generation)."""
    from levelsetpy_amd.context import device_grid
    from test_gpu_parity import dubins
    g, og = dubins([21, 19, 17])
    d0 = O.shape_cylinder(og, 2, None, .5)
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstENO2)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    yd1, sb1, _ = L.termLaxFriedrichs(0., y, sd)
    dg = device_grid(g)
    gen = dg.lib.hj_ctx_state_generation(dg.ctx)
    _ffi.check(dg.lib.hj_ctx_set_dissipation(dg.ctx, 1))            # local Lax-Friedrichs: a different stepBound
    _ffi.check(dg.lib.hj_ctx_set_post_step(dg.ctx, 1))
    assert dg.lib.hj_ctx_state_generation(dg.ctx) == gen + 2
    yd2, sb2, _ = L.termLaxFriedrichs(0., y, sd)
    assert sb1 == sb2 and torch.equal(yd1, yd2)


def test_range_dependent_step_bounds_arrive_late_but_equal_and_hjipde_solve_runs_the_native_loop(monkeypatch):
    """hj_rk_step with a range-dependent alpha returns while its last stage runs; the later stages' stepBounds reach the host
    asynchronously.  hj_rk_prev_bounds (no wait, one step late) and hj_rk_last_bounds (waits) must report the same numbers as a run that
    waits after every step, and the first stage's bound must be the deltaT / factorCFL the step used.  HJIPDE_solve drives such a system
    through hj_rk_integrate (the native loop) and lands where the step-by-step loop lands."""
    from levelsetpy_amd.context import device_grid
    n = (28, 22, 24)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    reg = L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1)
    dg = device_grid(g)
    dg.bind_stream()
    lib, ctx, sid, par = dg.lib, dg.ctx, _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], _ffi.darr([0.7])
    _ffi.check(lib.hj_ctx_set_dissipation(ctx, 0))
    _ffi.check(lib.hj_ctx_set_post_step(ctx, 0))
    _ffi.check(lib.hj_ctx_set_post_arrays(ctx, 0, None, 0, None))

    def run(wait_each):
        bufs = [torch.as_tensor(d0, device="cuda").clone() for _ in range(3)]
        w = torch.empty_like(bufs[0])
        t, rec = 0.0, []
        tout, dtout = C.c_double(), C.c_double()
        for k in range(3):
            cur, nxt = bufs[k % 2], bufs[(k + 1) % 2]
            _ffi.check(lib.hj_rk_step(ctx, 3, sid, reg.ham_id, par, t, 10.0, 0.8, 1e300, 0, C.c_void_p(cur.data_ptr()), C.c_void_p(nxt.data_ptr()),
                                      C.c_void_p(nxt.data_ptr()), C.c_void_p(w.data_ptr()), C.byref(tout), C.byref(dtout)))
            t = tout.value
            sbs, nsb, dt = (C.c_double * 3)(), C.c_int(), C.c_double()
            if wait_each:
                _ffi.check(lib.hj_rk_last_bounds(ctx, sbs, C.byref(nsb)))
                rec.append((dtout.value, [sbs[i] for i in range(nsb.value)]))
            else:
                _ffi.check(lib.hj_rk_prev_bounds(ctx, sbs, C.byref(nsb), C.byref(dt)))
                rec.append((dt.value if nsb.value else None, [sbs[i] for i in range(nsb.value)]))
        if not wait_each:
            sbs, nsb = (C.c_double * 3)(), C.c_int()
            _ffi.check(lib.hj_rk_last_bounds(ctx, sbs, C.byref(nsb)))
            rec.append((dtout.value, [sbs[i] for i in range(nsb.value)]))
        torch.cuda.synchronize()
        return t, bufs[3 % 2].clone(), rec
    ta, ya, ra = run(True)
    tb, yb, rb = run(False)
    assert ta == tb and torch.equal(ya, yb)
    assert all(len(b) == 3 for _, b in ra) and all(abs(dt - 0.8 * b[0]) <= 1e-15 * dt for dt, b in ra)
    assert rb[0] == (None, [])                       # nothing has arrived right after the first step
    assert rb[1] == ra[0] and rb[2] == ra[1]         # one step late, same numbers, each with its own deltaT
    assert rb[3] == ra[2]                            # asked for: waits
    # HJIPDE_solve: the native loop (hj_rk_integrate -> hj_rk_step per step) against the Python step loop
    sys_ = reg(g, [0.7], hamiltonian=lambda s, t, data, p, sd: BurgersDrift(g, 0.7).hamiltonian(t, data, p, sd),
               dissipation=lambda s, t, data, dmin, dmax, sd, dim: BurgersDrift(g, 0.7).dissipation(t, data, dmin, dmax, sd, dim))
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation))
    outs = []
    for stepwise in ("0", "1"):
        monkeypatch.setenv("HJ_HJIPDE_STEPWISE", stepwise)
        data, tau, _ = L.HJIPDE_solve(d0, [0., 0.01, 0.025], sd, 'minVOverTime', L.Bundle(dict(quiet=True, keepLast=True)))
        outs.append(np.asarray(data))
    assert np.array_equal(outs[0], outs[1])


def test_small_grids_take_the_direct_kernel_by_default(monkeypatch):
    """Below ~52^3 cells a 3-D grid runs direct_substep_kernel unless HJ_DIRECT_BELOW says otherwise (this suite sets it to 0 to keep its
    small grids on the tiled kernels); the two kernels give the same bits."""
    from levelsetpy_amd.context import device_grid
    from test_gpu_parity import dubins
    res = {}
    for below in (None, "0"):
        if below is None:
            monkeypatch.delenv("HJ_DIRECT_BELOW", raising=False)
        else:
            monkeypatch.setenv("HJ_DIRECT_BELOW", below)
        for n in (51, 60):
            g, og = dubins(n)                       # a new grid object: a new context, which reads the knob
            d0 = O.shape_cylinder(og, 2, None, .5)
            sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5)
            op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], torch.as_tensor(d0.reshape(-1, 1), device="cuda"), op, sd)
            dg = device_grid(g)
            res[(below, n)] = (t, y.clone(), dg.lib.hj_last_kernel(dg.ctx))
    assert res[(None, 51)][2] == b"direct_substep_kernel" and res[("0", 51)][2] == b"fused_substep_kernel"
    assert res[(None, 60)][2] == b"fused_substep_kernel"
    for n in (51, 60):
        assert res[(None, n)][0] == res[("0", n)][0] and torch.equal(res[(None, n)][1], res[("0", n)][1])


def test_deep_halo_stepper_refuses_the_intended_weno5_on_an_external_transport():
    """The intended WENO5's epsilon is a maximum over the WHOLE grid per stage; an external transport moves planes only.  Before round 5 the
    deep-halo stepper ran such a case with rank-local epsilons (5e-6 off the undivided run, silently: tests/fuzz_slabs.py found it)."""
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    from test_gpu_parity import dubins
    g, og = dubins([40, 12, 14])
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    full = torch.as_tensor(O.shape_cylinder(og, 2, None, .5), device="cuda")
    st = NativeSlabStepper(g, SlabDecomposition(40, 2, 0, False), _ffi.SCHEME_IDS["WENO5"], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.], dxs,
                           "float64", order=3, deep=True, external=lambda s: None)
    st.set_state(full[0:20])
    with pytest.raises(ValueError, match="all-reduce"):
        st.step(0.)
    st.close()
    # the as-shipped arithmetic has no grid-wide quantity: the same set-up steps
    st = NativeSlabStepper(g, SlabDecomposition(40, 2, 0, False), _ffi.SCHEME_IDS["WENO5_ASSHIPPED"], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.], dxs,
                           "float64", order=3, deep=True, external=lambda s: None)
    st.set_state(full[0:20])
    t, dt = st.step(0.)
    assert t > 0
    st.close()


def test_attach_after_first_use_is_picked_up_by_an_existing_schemedata(monkeypatch):
    """A schemeData that ran on the split path (its system object not yet attached) must take the fused path once the object IS attached:
    the classification cached per schemeData is revisited when attach() has been called since (round 5: examples/custom_hamiltonian.py
    ran both of its legs on the split path).  (HJ_TRACE=0: since round 6 the library would trace these callbacks by itself.)"""
    monkeypatch.setenv("HJ_TRACE", "0")
    n = (22, 20, 24)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0], n, None)
    d0 = O.shape_sphere(og, None, 0.5) + 0.05 * np.sin(3 * og.xs[0])
    sys_ = BurgersDrift(g, 0.7)
    sd = sdata(g, sys_, DERIV["WENO5_ASSHIPPED"])
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    a, sba, _ = L.termLaxFriedrichs(0., y, sd)
    assert not _last_kernel(g).endswith(b"(hipRTC)")
    L.register_native_hamiltonian("burgers_drift_3d", 3, _burgers_src(3), nparams=1).attach(sys_, params=lambda o: [o.c])
    b, sbb, _ = L.termLaxFriedrichs(0., y, sd)                 # the SAME Bundle
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    close(b.cpu().numpy(), a.cpu().numpy(), 1e-11)
    assert abs(sba - sbb) <= 1e-12 * sba


# ------------------------------------------------------------------------------ local Lax-Friedrichs variants with a range-reading Hamiltonian
class BurgersDriftLocal(BurgersDrift):
    """The same system with a partialFunc that takes derivMin / derivMax entries as scalars OR arrays (the local variants hand it the node's own
    range as arrays: diss_local_laxfried.py:108-111, diss_localsq_laxfried.py:87-90)."""

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        lo, hi = dmin[dim], dmax[dim]
        if _is_t(lo) or _is_t(hi):
            lo = lo if _is_t(lo) else torch.as_tensor(float(lo), device=data.device, dtype=data.dtype)
            hi = hi if _is_t(hi) else torch.as_tensor(float(hi), device=data.device, dtype=data.dtype)
            a = torch.maximum(lo.abs(), hi.abs())
        else:
            a = np.maximum(np.abs(lo), np.abs(hi))
        if dim != 1:
            return a
        x0 = np.abs(self.c * np.asarray(self.grid.xs[0]))
        return a + (torch.as_tensor(np.ascontiguousarray(x0), device=data.device) if _is_t(data) else x0)


@pytest.mark.parametrize("kind", ["llf", "lllf"])
@pytest.mark.parametrize("scheme", ["ENO2", "WENO5_ASSHIPPED", "WENO5"])
@pytest.mark.parametrize("n,pd", [((44, 37), 1), ((21, 23, 26), 2), ((9, 8, 10, 11), None)])
def test_local_lax_friedrichs_with_a_range_reading_hamiltonian_fused_vs_oracle_and_split(kind, scheme, n, pd):
    """VERDICT r04 missing 3: artificialDissipationLLF / LLLF with a partialFunc that USES the costate range -- per-node ranges
    [min(p-, p+), max(p-, p+)] in dimension i (LLF: grid-wide range in the others; LLLF: per-node everywhere), inside the fused kernel
    (HJ_DISS_LLF: range pass + substep; HJ_DISS_LLLF: one launch), bound 1 / max_x sum_i alpha_i(x) / dx_i reduced in the kernel.
    Against the oracle's artificial_dissipation_local and against the split path (the same object's Python callbacks)."""
    dim = len(n)
    g, og = mk([-1.0] * dim, [1.0 - (2.0 / n[d] if pd == d else 0) for d in range(dim)], n, pd)
    rng = np.random.default_rng(5)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[dim - 1]) + 0.02 * rng.standard_normal(n)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    diss = {"llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sys_ = BurgersDriftLocal(g, 0.7)

    def sd_of(s):
        return L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=diss, CoStateCalc=DERIV[scheme]))
    split, sb_s, _ = L.termLaxFriedrichs(0., y, sd_of(sys_))                   # not attached yet: the split path
    L.register_native_hamiltonian("burgers_drift_%dd" % dim, dim, _burgers_src(dim), nparams=1).attach(sys_, params=lambda o: [o.c])
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sd_of(sys_))
    assert _last_kernel(g).endswith(b"(hipRTC)"), _last_kernel(g)
    yo, sbo = O.term_lax_friedrichs(og, BurgersDriftLocal(og, 0.7), scheme, 0., d0.reshape(-1, 1), diss=kind)
    close(fused.cpu().numpy(), yo, 1e-11, what="fused vs oracle")
    close(split.cpu().numpy(), yo, 1e-11, what="split vs oracle")
    assert abs(sb_f - sbo) <= 1e-12 * sbo and abs(sb_s - sbo) <= 1e-12 * sbo, (sb_f, sb_s, sbo)
    # two steps of every order through the integrators: deltaT from the first stage's LOCAL bound (a pass of its own before the stage)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    for order, ode, oode in ((1, L.odeCFL1, O.ode_cfl_1), (3, L.odeCFL3, O.ode_cfl_3)):
        yy, t, yo2, to = y, 0., d0.reshape(-1, 1), 0.
        for _ in range(2):
            t, yy, _ = ode(L.termLaxFriedrichs, [t, 10.], yy, op, sd_of(sys_))
            to, yo2 = oode(lambda tt, v: O.term_lax_friedrichs(og, BurgersDriftLocal(og, 0.7), scheme, tt, v, diss=kind), [to, 10.], yo2, 0.8, single_step=True)
        assert abs(t - to) <= 1e-12 * to, (order, t, to)
        if scheme.startswith("ENO"):
            diff = np.abs(yy.cpu().numpy() - yo2)
            assert np.mean(diff > 1e-11) <= 2e-3 and diff.max() <= 1e-3
        else:
            close(yy.cpu().numpy(), yo2, 1e-11, what="2 steps, order %d" % order)
