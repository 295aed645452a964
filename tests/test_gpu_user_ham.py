"""GPU: Hamiltonians compiled at run time with hipRTC (C ABI hj_ham_register; levelsetpy_amd.user_ham).

A user's hamFunc / partialFunc pair written once more as a device expression runs through the SAME fused substep kernel
as the built-in systems.  Checked against (1) the built-in kernel of the same system (Dubins relative, double integrator:
agreement to rounding -- the run-time expression calls the device's sin / cos where the built-in kernel reads NumPy's
tables), (2) the CPU oracle with the user's own Python callbacks as the system (termLaxFriedrichs, odeCFL1/2/3, every
scheme), (3) the split path (the same callbacks on device arrays between the derivative and dissipation kernels)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd import _ffi  # noqa: E402
from levelsetpy_amd.context import device_grid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

from test_gpu_parity import mk, sdata, DERIV, SCHEMES, close, dubins  # noqa: E402

DUBINS_REL_COL = "col[0] = cos(x[2]); col[1] = sin(x[2]);"      # once per grid column, outside the march
DUBINS_REL_SRC = """
    const T c3 = col[0], s3 = col[1];
    H = p[0] * (par[0] - par[1] * c3) - p[1] * (par[1] * s3) - par[2] * fabs(p[0] * x[1] - p[1] * x[0] - p[2]) + par[2] * fabs(p[2]);
    alpha[0] = fabs(par[0] - par[1] * c3) + fabs(par[2] * x[1]);
    alpha[1] = fabs(par[1] * s3) + fabs(par[2] * x[0]);
    alpha[2] = par[3];
"""

# Dubins car in absolute coordinates, worst-case turn rate (the reference's DynamicalSystems/dubins_absolute.py:150-170 has this
# dissipation; its Hamiltonian is the standard one): H = v (p1 cos th + p2 sin th) + w |p3|
DUBINS_ABS_SRC = """
    H = par[0] * (p[0] * cos(x[2]) + p[1] * sin(x[2])) + par[1] * fabs(p[2]);
    alpha[0] = fabs(par[0] * cos(x[2]));
    alpha[1] = fabs(par[0] * sin(x[2]));
    alpha[2] = par[1];
"""


def _is_t(a):
    return type(a).__module__.startswith("torch")


class DubinsAbsPy(object):
    """The same system as the reference would write it: array callbacks (NumPy or torch arrays alike)."""

    def __init__(self, grid, v, w):
        self.grid, self.v, self.w = grid, v, w

    def hamiltonian(self, t, data, p, sd=None):
        x3 = np.asarray(self.grid.xs[2])
        if _is_t(p[0]):
            x3 = torch.as_tensor(x3, device=p[0].device)
            return self.v * (p[0] * torch.cos(x3) + p[1] * torch.sin(x3)) + self.w * p[2].abs()
        return self.v * (p[0] * np.cos(x3) + p[1] * np.sin(x3)) + self.w * np.abs(p[2])

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        x3 = np.asarray(self.grid.xs[2])
        if dim == 0:
            a = np.abs(self.v * np.cos(x3))
        elif dim == 1:
            a = np.abs(self.v * np.sin(x3))
        else:
            return self.w
        a = np.broadcast_to(a, tuple(int(v) for v in np.asarray(self.grid.N).ravel())) if hasattr(self.grid, "N") else a
        return torch.as_tensor(np.ascontiguousarray(a), device=data.device) if _is_t(data) else a


@pytest.mark.parametrize("scheme", SCHEMES)
def test_runtime_dubins_relative_equals_builtin_and_oracle(scheme):
    g, og = dubins([23, 21, 19])
    d0 = O.shape_cylinder(og, 2, None, .5) + 0.03 * np.random.default_rng(2).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    reg = L.register_native_hamiltonian("dubins_rel_rt", 3, DUBINS_REL_SRC, nparams=4, column_src=DUBINS_REL_COL, ncol=2)
    user = reg(g, [1.0, 1.0, 1.0, 2.0])
    builtin = L.DubinsVehicleRel(g, 1, 1)
    yd_u, sb_u, _ = L.termLaxFriedrichs(0., y, sdata(g, user, DERIV[scheme]))
    dg = device_grid(g)
    assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel (hipRTC)"
    yd_b, sb_b, _ = L.termLaxFriedrichs(0., y, sdata(g, builtin, DERIV[scheme]))
    assert dg.lib.hj_last_kernel(dg.ctx) != b"fused_pair_kernel (hipRTC)"
    assert abs(sb_u - sb_b) <= 1e-14 * sb_b
    close(yd_u.cpu().numpy(), yd_b.cpu().numpy(), 1e-12, what="run-time vs built-in term")
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), scheme, 0., d0.reshape(-1, 1))
    close(yd_u.cpu().numpy(), yo, 1e-11, what="run-time vs oracle")
    assert abs(sb_u - sbo) <= 1e-13 * sbo
    # five RK3 steps, then one RK2 and one RK1 step: the integrators take the new id like any other
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    yu, yb, tu, tb = y, y, 0., 0.
    for _ in range(5):
        tu, yu, _ = L.odeCFL3(L.termLaxFriedrichs, [tu, 10.], yu, op, sdata(g, user, DERIV[scheme]))
        tb, yb, _ = L.odeCFL3(L.termLaxFriedrichs, [tb, 10.], yb, op, sdata(g, builtin, DERIV[scheme]))
    assert abs(tu - tb) <= 1e-14
    if scheme.startswith("WENO"):
        close(yu.cpu().numpy(), yb.cpu().numpy(), 1e-11, what="5 RK3 steps")
    else:   # an ENO stencil choice may flip where the two arithmetics differ in the last bit: bound the fraction
        diff = (yu - yb).abs()
        assert float((diff > 1e-11).double().mean()) <= 2e-3 and float(diff.max()) <= 1e-3
    for ode in (L.odeCFL2, L.odeCFL1):
        t2, y2, _ = ode(L.termLaxFriedrichs, [0., 10.], y, op, sdata(g, user, DERIV[scheme]))
        t3, y3, _ = ode(L.termLaxFriedrichs, [0., 10.], y, op, sdata(g, builtin, DERIV[scheme]))
        assert abs(t2 - t3) <= 1e-14
        close(y2.cpu().numpy(), y3.cpu().numpy(), 1e-12)


@pytest.mark.parametrize("scheme", ["ENO3", "WENO5_ASSHIPPED"])
def test_runtime_dubins_absolute_fused_equals_split_path_and_oracle(scheme):
    """A system the library was NOT built with: attached to the caller's own object, selected by the identity of its bound
    methods; the fused run-time kernel against the split path (the same object's Python callbacks) and the CPU oracle."""
    n = (22, 20, 24)
    g, og = mk([-2., -2., -np.pi], [2., 2., np.pi * (1 - 2 / n[2])], n, 2)
    d0 = O.shape_sphere(og, None, 1.0) + 0.03 * np.random.default_rng(5).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    veh = DubinsAbsPy(g, 1.3, 0.7)
    sd = sdata(g, veh, DERIV[scheme])
    split, sb_s, _ = L.termLaxFriedrichs(0., y, sd)                      # not attached yet: the split path
    dg = device_grid(g)
    reg = L.register_native_hamiltonian("dubins_abs_rt", 3, DUBINS_ABS_SRC, nparams=2)
    reg.attach(veh, params=lambda o: [o.v, o.w])
    sd = sdata(g, veh, DERIV[scheme])                                    # (a fresh Bundle: the old one's plan says "split")
    fused, sb_f, _ = L.termLaxFriedrichs(0., y, sd)
    assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel (hipRTC)"
    assert abs(sb_f - sb_s) <= 1e-13 * sb_s
    close(fused.cpu().numpy(), split.cpu().numpy(), 1e-11, what="fused vs split")
    ov = DubinsAbsPy(og, 1.3, 0.7)
    yo, sbo = O.term_lax_friedrichs(og, ov, scheme, 0., d0.reshape(-1, 1))
    close(fused.cpu().numpy(), yo, 1e-11, what="fused vs oracle")
    assert abs(sb_f - sbo) <= 1e-13 * sbo
    # a parameter changed in place is picked up (the cached plan is re-validated)
    veh.v = 2.0
    f2, sb2, _ = L.termLaxFriedrichs(0., y, sd)
    yo2, sbo2 = O.term_lax_friedrichs(og, DubinsAbsPy(og, 2.0, 0.7), scheme, 0., d0.reshape(-1, 1))
    close(f2.cpu().numpy(), yo2, 1e-11)
    assert abs(sb2 - sbo2) <= 1e-13 * sbo2
    # three RK3 steps over a time span (hj_rk_integrate: the native loop) vs the oracle
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.7)))
    tf = 3.2 * 0.7 * sbo2
    t, yn, _ = L.odeCFL3(L.termLaxFriedrichs, [0., tf], y, op, sd)
    to, yoo = O.ode_cfl_3(lambda tt, yy: O.term_lax_friedrichs(og, DubinsAbsPy(og, 2.0, 0.7), scheme, tt, yy), [0., tf], d0.reshape(-1, 1), 0.7)
    assert abs(float(t) - to) <= 1e-13
    if scheme.startswith("WENO"):
        close(yn.cpu().numpy(), yoo, 1e-11, what="span")
    else:
        diff = np.abs(yn.cpu().numpy() - yoo)
        assert float(np.mean(diff > 1e-11)) <= 2e-3 and diff.max() <= 1e-3


def test_runtime_2d_hamiltonian_restrict_update_and_hjipde_solve():
    """2-D: the double integrator as a run-time expression; termRestrictUpdate (the clamp is a run-time flag of the MODE 0
    instantiation) and HJIPDE_solve('minVOverTime') take it; compared with the built-in kernel."""
    g, og = mk([-1, -1], [1, 1], [70, 64], None)
    d0 = O.shape_sphere(og, None, .35)
    reg = L.register_native_hamiltonian("dint_rt", 2, "H = -(p[0] * x[1] - fabs(p[1]) * par[0]); alpha[0] = fabs(x[1]); alpha[1] = fabs(par[0]);", nparams=1)
    user, builtin = reg(g, [1.0]), L.DoubleIntegrator(g, 1)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    res = []
    for sys_ in (user, builtin):
        sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sdata(g, sys_, L.upwindFirstWENO5), positive=0))
        y, t = torch.as_tensor(d0.reshape(-1), device="cuda"), 0.
        for _ in range(4):
            t, y, _ = L.odeCFL3(L.termRestrictUpdate, [t, 10.], y, op, sdr)
        res.append((t, y.cpu().numpy()))
    assert abs(res[0][0] - res[1][0]) <= 1e-15
    close(res[0][1], res[1][1], 1e-12, what="restrict update")
    outs = []
    for sys_ in (user, builtin):
        sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation))
        data, tau, _ = L.HJIPDE_solve(d0, [0., 0.02, 0.05], sd, 'minVOverTime', L.Bundle(dict(quiet=True, keepLast=True)))
        outs.append(np.asarray(data))
    close(outs[0], outs[1], 1e-12, what="HJIPDE_solve")


def test_runtime_hamiltonian_through_the_slab_stepper_virtual_ranks():
    """The deep-halo slab stepper with a run-time Hamiltonian id: three virtual ranks, pad planes moved by the test, bitwise
    equal to the undivided grid run with the same kernel."""
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    from levelsetpy_amd.context import DeviceGrid
    n, world = (61, 20, 22), 3
    g, og = mk([-2., -2., -np.pi], [2., 2., np.pi * (1 - 2 / n[2])], n, 2)
    reg = L.register_native_hamiltonian("dubins_abs_rt", 3, DUBINS_ABS_SRC, nparams=2)
    par = [1.3, 0.7]
    full = torch.as_tensor(O.shape_sphere(og, None, 1.0) + 0.02 * np.random.default_rng(8).standard_normal(og.shape), device="cuda")
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    steppers = [NativeSlabStepper(g, SlabDecomposition(n[0], world, r, False), sid, reg.ham_id, par, dxs, order=3, deep=True,
                                  external=lambda st: None) for r in range(world)]
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
    cur, nxt, w0, w1 = full.clone(), torch.empty_like(full), torch.empty_like(full), torch.empty_like(full)
    tout, dtout = C.c_double(), C.c_double()
    t = tr = 0.
    for _ in range(3):
        ts = [st.step(t) for st in steppers]
        move_pads()
        t, dt = ts[0]
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, 3, sid, reg.ham_id, _ffi.darr(par), tr, 1e9, 0.8, dt, 0, dg.ptr(cur), dg.ptr(nxt),
                                     dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
        cur, nxt = nxt, cur
        tr = float(tout.value)
        assert dtout.value == dt
    torch.cuda.synchronize()
    for st in steppers:
        got, ref = st.state(), cur[st.slab.begin:st.slab.end]
        assert torch.equal(got, ref), "rank %d differs by %g" % (st.slab.rank, float((got - ref).abs().max()))
        st.close()


_CACHE_SCRIPT = r'''
import sys, time, json, hashlib
import numpy as np, torch
import levelsetpy_amd as L
reg = L.register_native_hamiltonian("cache_probe", 2, "H = -(p[0] * x[1]) + par[0] * fabs(p[1]); alpha[0] = fabs(x[1]); alpha[1] = par[0];", nparams=1)
g = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), np.array([[70], [90]], dtype=np.int64), None)
s = reg(g, [1.5])
sd = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO3))
y0 = torch.as_tensor(L.shapeSphere(g, np.zeros((2, 1)), .4), device="cuda").reshape(-1, 1)
t0 = time.perf_counter()
t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 0.05], y0, L.odeCFLset(L.Bundle(dict(factorCFL=.8))), sd)
torch.cuda.synchronize()
sec = time.perf_counter() - t0
print(json.dumps({"stats": L.kernel_cache_stats(), "sec": sec, "t": t, "sha": hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()}))
'''


def test_compiled_kernels_are_cached_on_disk(tmp_path):
    """A registered expression is compiled once per (source, headers, options), not once per process: the second process
    loads every kernel from $HJ_RTC_CACHE (no hipRTC compile, the cache files untouched) and computes the same bits;
    HJ_RTC_CACHE=0 compiles again and leaves the directory alone."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = tmp_path / "rtc"

    def run(cache_env):
        env = dict(os.environ, HJ_RTC_CACHE=cache_env, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        p = subprocess.run([sys.executable, "-c", _CACHE_SCRIPT], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])
    first = run(str(cache))
    assert first["stats"][0] >= 2 and first["stats"][1] == 0           # substep kernel(s) + the alpha-bound kernel, all compiled
    files = sorted(cache.glob("*.hjco"))
    assert len(files) == first["stats"][0]
    stamp = [(f.name, f.stat().st_mtime_ns, f.stat().st_size) for f in files]
    second = run(str(cache))
    assert second["stats"] == [0, first["stats"][0]]                   # nothing compiled, everything loaded
    assert second["sha"] == first["sha"] and second["t"] == first["t"]
    assert [(f.name, f.stat().st_mtime_ns, f.stat().st_size) for f in sorted(cache.glob("*.hjco"))] == stamp
    off = run("0")
    assert off["stats"] == [first["stats"][0], 0] and off["sha"] == first["sha"]
    assert [(f.name, f.stat().st_mtime_ns, f.stat().st_size) for f in sorted(cache.glob("*.hjco"))] == stamp


# ------------------------------------------------------------------------------ the big shape of the run-time kernels (grids of 6.5 M cells and more)
def _tile_cells(dg):
    e = (C.c_int * 4)()
    _ffi.check(dg.lib.hj_last_tile(dg.ctx, e))
    return int(e[1]) * max(1, int(e[2]))


@pytest.mark.parametrize("scheme,dtype", [("WENO5_ASSHIPPED", "float64"), ("ENO2", "float64"), ("WENO5_ASSHIPPED", "float32")])
def test_runtime_kernels_in_the_big_shape_equal_the_builtin_kernels(scheme, dtype):
    """From 6.5 M cells the run-time kernels of a light stencil take the shape the built-in ones run there (512 threads x 2 pairs + the parked halo
    ring; hj_rtc.hip, shape 1) -- every other test of this suite is below that size.  190 x 186 x 188 Dubins grid: the term and one odeCFL3 step of the
    run-time expression against the built-in system (the same arithmetic in the same kernel template)."""
    g, og = dubins([190, 186, 188])
    d0 = O.shape_cylinder(og, 2, None, .5) + 0.03 * np.random.default_rng(2).standard_normal(og.shape)
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda", dtype=getattr(torch, dtype))
    tol = 1e-12 if dtype == "float64" else 2e-6          # (fp32: the same template and arithmetic, the compiler may contract differently)
    user = L.register_native_hamiltonian("dubins_rel_rt", 3, DUBINS_REL_SRC, nparams=4, column_src=DUBINS_REL_COL, ncol=2)(g, [1.0, 1.0, 1.0, 2.0])
    builtin = L.DubinsVehicleRel(g, 1, 1)
    dg = device_grid(g, dtype)
    yd_u, sb_u, _ = L.termLaxFriedrichs(0., y, sdata(g, user, DERIV[scheme]))
    assert yd_u.dtype == y.dtype
    assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel (hipRTC)" and _tile_cells(dg) > 1024, (dg.lib.hj_last_kernel(dg.ctx), _tile_cells(dg))
    yd_b, sb_b, _ = L.termLaxFriedrichs(0., y, sdata(g, builtin, DERIV[scheme]))
    assert abs(sb_u - sb_b) <= (1e-14 if dtype == "float64" else 1e-6) * sb_b
    assert float((yd_u - yd_b).abs().max()) <= tol * float(yd_b.abs().max())
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    tu, yu, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y, op, sdata(g, user, DERIV[scheme]))
    tb, yb, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 10.], y, op, sdata(g, builtin, DERIV[scheme]))
    assert abs(tu - tb) <= (1e-14 if dtype == "float64" else 1e-9)
    diff = (yu - yb).abs()
    if scheme.startswith("WENO"):
        assert float(diff.max()) <= tol * 10
    else:
        assert float((diff > 1e-11).double().mean()) <= 2e-3 and float(diff.max()) <= 1e-3


class WavyPy(object):
    """A deliberately heavy expression (double-precision sines, cosines and exponentials of the coordinates in every coefficient): at two pairs per
    thread it may not fit the register file, in which case the library falls back to the small shape for it (hj_rtc.hip, big_spills)."""

    def __init__(self, grid, c):
        self.grid, self.c = grid, c

    def coeffs(self):
        x0, x1, x2 = (np.asarray(v) for v in self.grid.xs)
        a = np.sin(x0) * np.cos(x1) + np.exp(-x2 * x2)
        b = np.cos(x0 + 0.5 * x1) * np.sin(2 * x2) - 0.3 * np.exp(-x0 * x0)
        w = np.sin(x1 - x2) * np.cos(0.7 * x0) + 0.2
        return a, b, w

    def hamiltonian(self, t, data, p, sd=None):
        a, b, w = self.coeffs()
        return a * p[0] + b * p[1] + w * p[2] + self.c * (np.abs(p[0]) + np.abs(p[2]))

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        a, b, w = self.coeffs()
        return [np.abs(a) + self.c, np.abs(b), np.abs(w) + self.c][dim]


WAVY_SRC = """
    const T a = sin(x[0]) * cos(x[1]) + exp(-x[2] * x[2]);
    const T b = cos(x[0] + T(0.5) * x[1]) * sin(T(2) * x[2]) - T(0.3) * exp(-x[0] * x[0]);
    const T w = sin(x[1] - x[2]) * cos(T(0.7) * x[0]) + T(0.2);
    H = a * p[0] + b * p[1] + w * p[2] + par[0] * (fabs(p[0]) + fabs(p[2]));
    alpha[0] = fabs(a) + par[0]; alpha[1] = fabs(b); alpha[2] = fabs(w) + par[0];
"""


def test_a_heavy_runtime_expression_on_a_big_grid_vs_oracle():
    """The same size with an expression that is expensive in registers: whichever shape the library ends up with (the big one, or the small one after the
    big one was found to need scratch), the term and its bound equal the oracle's; the shape taken is reported."""
    n = (190, 186, 188)
    g, og = mk([-1.0, -1.2, -0.9], [1.0, 1.2, 0.9 * (1 - 2 / n[2])], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    user = L.register_native_hamiltonian("wavy_3d", 3, WAVY_SRC, nparams=1)(g, [0.4])
    dg = device_grid(g)
    yd, sb, _ = L.termLaxFriedrichs(0., y, sdata(g, user, DERIV["WENO5_ASSHIPPED"]))
    assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel (hipRTC)"
    print("heavy expression at %d cells: tile of %d cells (%s shape)" % (int(np.prod(n)), _tile_cells(dg), "big" if _tile_cells(dg) > 1024 else "small"))
    yo, sbo = O.term_lax_friedrichs(og, WavyPy(og, 0.4), "WENO5_ASSHIPPED", 0., d0.reshape(-1, 1))
    close(yd.cpu().numpy(), yo, 1e-11, what="heavy expression vs oracle")
    assert abs(sb - sbo) <= 1e-12 * sbo
    # a second call takes the remembered shape; one RK2 step lands where the oracle lands
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    t, y1, _ = L.odeCFL2(L.termLaxFriedrichs, [0., 10.], y, op, sdata(g, user, DERIV["WENO5_ASSHIPPED"]))
    to, yo1 = O.ode_cfl_2(lambda tt, v: O.term_lax_friedrichs(og, WavyPy(og, 0.4), "WENO5_ASSHIPPED", tt, v), [0., 10.], d0.reshape(-1, 1), 0.8, single_step=True)
    assert abs(t - to) <= 1e-13 * to
    close(y1.cpu().numpy(), yo1, 1e-11, what="heavy expression, one RK2 step")


@pytest.mark.parametrize("kind", ["llf", "lllf"])
def test_a_range_reading_expression_on_a_big_grid_vs_oracle(kind):
    """... and the range path at that size (range pass, local evaluation, in-kernel bound in the big shape): the cross-dimension Hamiltonian of
    test_gpu_round5.py, term and bound against the oracle."""
    from test_gpu_round5 import CoupledBurgers, _coupled_src
    n = (190, 186, 188)
    g, og = mk([-1.0] * 3, [1.0, 1.0, 1.0 - 2.0 / n[2]], n, 2)
    d0 = O.shape_sphere(og, None, 0.5) + 0.1 * np.sin(3 * og.xs[0]) * np.cos(2 * og.xs[2])
    y = torch.as_tensor(d0.reshape(-1, 1), device="cuda")
    sys_ = CoupledBurgers(g, 0.6)
    L.register_native_hamiltonian("coupled_burgers_3d", 3, _coupled_src(3), nparams=1).attach(sys_, params=lambda o: [o.c])
    diss = {"llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, dissFunc=diss, CoStateCalc=DERIV["WENO5_ASSHIPPED"]))
    dg = device_grid(g)
    yd, sb, _ = L.termLaxFriedrichs(0., y, sd)
    assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel (hipRTC)" and _tile_cells(dg) > 1024
    yo, sbo = O.term_lax_friedrichs(og, CoupledBurgers(og, 0.6), "WENO5_ASSHIPPED", 0., d0.reshape(-1, 1), diss=kind)
    close(yd.cpu().numpy(), yo, 1e-11, what="range path, big shape")
    assert abs(sb - sbo) <= 1e-12 * sbo
