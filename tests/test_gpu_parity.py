"""GPU parity tests: the HIP path (through the C ABI / the reference-named Python API) against
the CPU oracle and against the golden vectors generated from the reference.

Tolerances (fp64): ghost cells bit-exact; derivatives / ydot abs <= 1e-11*max(1,|ref|_inf)
(the kernels contract a*b+c into FMAs and evaluate WENO quotients with one division, so they
are not bit-identical to NumPy); stepBound / t relative 1e-13; states after 5 RK3 steps 1e-11.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import levelsetpy_amd as L  # noqa: E402
from levelsetpy_amd import _ffi  # noqa: E402
from levelsetpy_amd.context import DeviceGrid, device_grid  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SCHEMES = ["ENO2", "ENO3", "WENO5_ASSHIPPED", "WENO5"]
DERIV = {"ENO2": L.upwindFirstENO2, "ENO3": L.upwindFirstENO3, "WENO5_ASSHIPPED": L.upwindFirstWENO5,
         "WENO5": L.upwindFirstWENO5Intended}


def close(a, ref, tol=1e-11, what=""):
    a, ref = np.asarray(a), np.asarray(ref)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    scale = max(1.0, float(np.max(np.abs(ref))))
    err = float(np.max(np.abs(a - ref)))
    assert err <= tol * scale, "%s err %.3e > %.1e*%.3g" % (what, err, tol, scale)


ENO_STATS = []      # (what, fraction of cells beyond the strict tolerance, max error / scale): read by the report test


def close_eno(a, ref, tol=1e-11, loose=1e-4, what="", frac=0.0):
    """Multi-step ENO2/ENO3 on TIE-PRONE data: the signed-distance cylinder is symmetric, many |D2| / |D3|
    comparisons are exact or near ties, and a stencil choice that flips against the reference propagates over the
    following substeps (rounds 1-2: up to 2.6 % of the cells beyond 1e-11 after five RK3 steps, hence a masked
    comparison).  Since round 3 the fused ENO substep is evaluated in the reference's operation order with contraction
    off (hj_device.h, np_order): on the native path the results equal the reference's BIT FOR BIT
    (test_eno_paths_bitwise_equal_the_reference_golden), so the default here is STRICT -- frac = 0: every cell within
    `tol`.  `frac` remains for comparisons against paths that are not NumPy-ordered (the oracle's own runs through a
    foreign schemeFunc are; nothing in this file passes a non-zero value any more)."""
    a, ref = np.asarray(a), np.asarray(ref)
    assert a.shape == ref.shape
    scale = max(1.0, float(np.max(np.abs(ref))))
    diff = np.abs(a - ref)
    err = float(np.max(diff))
    beyond = float(np.mean(diff > tol * scale))
    ENO_STATS.append((what, beyond, err / scale))
    assert err <= loose * scale, "%s max err %.3e" % (what, err)
    if frac is not None:
        assert beyond <= frac, "%s: %.3e of the cells differ by more than %.1e (allowed %.1e)" % (what, beyond, tol, frac)


def mk(gmin, gmax, N, pd):
    """(product grid Bundle, oracle Grid) with identical parameters."""
    N = [int(n) for n in N]
    g = L.createGrid(np.asarray(gmin, dtype=np.float64).reshape(-1, 1), np.asarray(gmax, dtype=np.float64).reshape(-1, 1),
                     np.asarray(N, dtype=np.int64).reshape(-1, 1), pd if pd is not None else None)
    if isinstance(pd, (list, tuple)):
        og = O.Grid(gmin, gmax, N, list(pd))
    else:
        og = O.Grid(gmin, gmax, N, [pd] if pd else [])
    return g, og


def dubins(n, pd=2):
    n = np.atleast_1d(n)
    if n.size == 1:
        n = np.repeat(n, 3)
    return mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)


def sdata(g, sys_, fn):
    return L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                         dissFunc=L.artificialDissipationGLF, CoStateCalc=fn))


# ------------------------------------------------------------------------------ ghosts
@pytest.mark.parametrize("dim", [0, 1, 2])
@pytest.mark.parametrize("w", [1, 2, 3])
def test_ghost_bit_exact_vs_reference(golden, dim, w):
    G = golden("ghost.npz")
    x = G["x"]
    assert np.array_equal(L.addGhostPeriodic(x, dim, w, None), G["per_d%d_w%d" % (dim, w)])
    assert np.array_equal(L.addGhostExtrapolate(x, dim, w, None), G["ext_d%d_w%d_tz0" % (dim, w)])
    assert np.array_equal(L.addGhostExtrapolate(x, dim, w, L.Bundle(dict(towardZero=True))),
                          G["ext_d%d_w%d_tz1" % (dim, w)])


def test_ghost_dtype_shapes_errors(golden):
    G = golden("ghost.npz")
    out = L.addGhostExtrapolate(G["x"].astype(np.float32), 1, 2, None)
    assert out.dtype == np.float64 and np.array_equal(out, G["ext_f32in_d1_w2"])
    x = np.arange(12.0).reshape(3, 4)
    assert L.addGhostPeriodic(x, 1, None).shape == (3, 6)          # width None -> 1
    with pytest.raises(ValueError):
        L.addGhostExtrapolate(x, 0, 4)
    t = torch.as_tensor(x, device="cuda")
    o = L.addGhostPeriodic(t, 0, 2)
    assert torch.is_tensor(o) and o.is_cuda and np.array_equal(o.cpu().numpy(), O.add_ghost_periodic(x, 0, 2))
    # 1-D and 5-D arrays pad too
    v = np.linspace(-1, 1, 7)
    assert np.array_equal(L.addGhostExtrapolate(v, 0, 3), O.add_ghost_extrapolate(v, 0, 3))
    z = np.random.default_rng(0).standard_normal((2, 3, 4, 3, 2))
    assert np.array_equal(L.addGhostPeriodic(z, 2, 2), O.add_ghost_periodic(z, 2, 2))


def test_add_ghost_all_dims():
    g, og = dubins([6, 7, 8])
    x = np.random.default_rng(3).standard_normal(g.shape)
    assert np.array_equal(L.addGhostAllDims(g, x, 2), O.add_ghost_all_dims(og, x, 2))


# ------------------------------------------------------------------------------ derivatives
CASES = [("g2", 2), ("g3", 3), ("g3s", 3), ("g4", 4)]


def _grids_for(G, tag):
    base = "g3" if tag == "g3s" else tag
    data = G[tag + "_data"]
    pd = [i for i, b in enumerate(G[tag + "_bc"]) if b]
    g, og = mk(G[base + "_min"], G[base + "_max"], data.shape, pd if len(pd) != 1 else pd[0])
    return g, og, data


@pytest.mark.parametrize("tag,nd", CASES)
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_upwind_vs_reference_golden(golden, tag, nd, scheme):
    G = golden("deriv.npz")
    g, og, data = _grids_for(G, tag)
    for dim in range(nd):
        dL, dR = DERIV[scheme](g, data, dim)
        close(dL, G["%s_%s_L_d%d" % (tag, scheme, dim)], what="L d%d" % dim)
        close(dR, G["%s_%s_R_d%d" % (tag, scheme, dim)], what="R d%d" % dim)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape,pd", [((37, 29), None), ((23, 41), 1), ((19, 17, 23), 2),
                                       ((9, 8, 7, 6), (0, 1, 2, 3)), ((6, 7, 8, 9), None)])
def test_upwind_vs_oracle_random(scheme, shape, pd):
    nd = len(shape)
    g, og = mk([-1.0] * nd, [1.0 + 0.1 * i for i in range(nd)], shape, pd)
    rng = np.random.default_rng(5)
    data = np.cos(3 * og.xs[0]) + 0.2 * rng.standard_normal(shape)
    for dim in range(nd):
        dL, dR = DERIV[scheme](g, data, dim)
        oL, oR = O.SCHEMES[scheme](og, data, dim)
        close(dL, oL, what="%s L d%d" % (scheme, dim))
        close(dR, oR, what="%s R d%d" % (scheme, dim))


def test_upwind_minmax_and_generate_all():
    g, og = dubins([11, 12, 13])
    data = np.random.default_rng(1).standard_normal(g.shape)
    dg = device_grid(g)
    dg.bind_stream()
    phi = dg.to_device(data)
    a, b = dg.empty(), dg.empty()
    mm = (C.c_double * 4)()
    _ffi.check(dg.lib.hj_upwind(dg.ctx, _ffi.ENO3, 1, dg.ptr(phi), dg.ptr(a), dg.ptr(b), mm))
    oL, oR = O.upwind_first_eno3(og, data, 1)
    ref = [oL.min(), oL.max(), oR.min(), oR.max()]
    for k in range(4):
        assert abs(mm[k] - ref[k]) <= 1e-11 * max(1, abs(ref[k]))
    cl, cr = L.upwindFirstENO3(g, data, 1, True)
    ol, orr, _ = O.eno3_helper(og, data, 1)
    for k in range(3):
        close(cl[k], ol[k])
        close(cr[k], orr[k])
    with pytest.raises(ValueError):
        L.upwindFirstENO3(g, data, 3)
    with pytest.raises(ValueError):
        L.upwindFirstENO3(g, data[:-1], 0)


# ------------------------------------------------------------------------------ LF term
@pytest.mark.parametrize("ub,wb", [(1, 1), (5, 5), (2, 3)])
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_term_dubins_vs_reference_golden(golden, ub, wb, scheme):
    G = golden("term.npz")
    g, og = mk(G["dub_min"], G["dub_max"], G["dub_N"], 2)
    sys_ = L.DubinsVehicleRel(g, ub, wb)
    y = G["dub_data"].reshape(-1, 1)
    yd, sb, _ = L.termLaxFriedrichs(0.3, y, sdata(g, sys_, DERIV[scheme]))
    assert yd.shape == y.shape and isinstance(sb, float)
    close(yd, G["dub_u%d_w%d_%s_ydot" % (ub, wb, scheme)])
    ref = float(G["dub_u%d_w%d_%s_sb" % (ub, wb, scheme)])
    assert abs(sb - ref) <= 1e-13 * ref


@pytest.mark.parametrize("ub", [1, 2.5])
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_term_double_integrator_vs_reference_golden(golden, ub, scheme):
    G = golden("term.npz")
    g, og = mk(G["di_min"], G["di_max"], G["di_N"], None)
    sys_ = L.DoubleIntegrator(g, ub)
    yd, sb, _ = L.termLaxFriedrichs(0., G["di_data"].reshape(-1, 1), sdata(g, sys_, DERIV[scheme]))
    close(yd, G["di_u%s_%s_ydot" % (ub, scheme)])
    ref = float(G["di_u%s_%s_sb" % (ub, scheme)])
    assert abs(sb - ref) <= 1e-13 * ref


def _term_both_kernels(g, sys_, scheme, y, monkeypatch):
    """ydot from the tiled kernel and from the direct kernel (fresh ctx each)."""
    outs = []
    for force in ("0", "1"):
        monkeypatch.setenv("HJ_FORCE_DIRECT", force)
        g.__dict__.pop("_hj_device", None)
        outs.append(L.termLaxFriedrichs(0., y, sdata(g, sys_, DERIV[scheme])))
    g.__dict__.pop("_hj_device", None)
    return outs


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n", [(33, 21, 19), (12, 70, 16), (40, 9, 130)])
def test_term_tiled_and_direct_vs_oracle_3d(scheme, n, monkeypatch):
    g, og = dubins(n)
    rng = np.random.default_rng(7)
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * rng.standard_normal(g.shape)
    y = data.reshape(-1, 1)
    (yt, sbt, _), (yd, sbd, _) = _term_both_kernels(g, L.DubinsVehicleRel(g, 2, 3), scheme, y, monkeypatch)
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 2, 3), scheme, 0., y)
    close(yt, yo, what="tiled")
    close(yd, yo, what="direct")
    assert np.array_equal(yt, yd) and sbt == sbd        # round 3: the two kernels agree bit for bit
    assert abs(sbt - sbo) <= 1e-13 * sbo and abs(sbd - sbo) <= 1e-13 * sbo


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((50, 45), None), ((20, 3000), 1), ((300, 12), 0)])
def test_term_tiled_and_direct_vs_oracle_2d(scheme, n, pd, monkeypatch):
    if pd == 0:
        pd = (0,)
    g, og = mk([-1, -1], [1, 1], n, pd)
    rng = np.random.default_rng(8)
    data = O.shape_sphere(og, None, .25) + 0.02 * rng.standard_normal(g.shape)
    y = data.reshape(-1, 1)
    (yt, sbt, _), (yd, sbd, _) = _term_both_kernels(g, L.DoubleIntegrator(g, 1.5), scheme, y, monkeypatch)
    yo, sbo = O.term_lax_friedrichs(og, O.DoubleIntegrator(og, 1.5), scheme, 0., y)
    close(yt, yo, what="tiled")
    close(yd, yo, what="direct")
    assert np.array_equal(yt, yd) and sbt == sbd        # round 3: the two kernels agree bit for bit
    assert abs(sbt - sbo) <= 1e-13 * sbo and abs(sbd - sbo) <= 1e-13 * sbo


@pytest.mark.parametrize("scheme", ["ENO3", "WENO5"])
def test_term_4d_pendulum_vs_oracle(scheme):
    n = (9, 8, 10, 7)
    gmin = [-np.pi, -8, -np.pi, -8]
    gmax = [np.pi * (1 - 2 / n[0]), 8 * (1 - 2 / n[1]), np.pi * (1 - 2 / n[2]), 8 * (1 - 2 / n[3])]
    g, og = mk(gmin, gmax, n, (0, 1, 2, 3))
    data = O.shape_sphere(og, None, .5) + 0.01 * np.random.default_rng(2).standard_normal(n)
    y = data.reshape(-1, 1)
    yd, sb, _ = L.termLaxFriedrichs(0., y, sdata(g, L.DoublePendulum4D(g, 1.0), DERIV[scheme]))
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, 0., y)
    close(yd, yo, 1e-10)
    assert abs(sb - sbo) <= 1e-12 * sbo


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n,pd", [((14, 13, 17, 15), (0, 2)), ((8, 25, 9, 30), (1, 3)), ((21, 7, 8, 33), None)])
def test_term_tiled_and_direct_vs_oracle_4d(scheme, n, pd, monkeypatch):
    """4-D grids tile three plane axes (halo cross on each): periodic and extrapolated axes mixed,
    extents that are not multiples of the tile, several tiles per axis."""
    gmin = [-np.pi, -8, -np.pi, -8]
    gmax = [np.pi * (1 - 2 / n[0]), 8 * (1 - 2 / n[1]), np.pi * (1 - 2 / n[2]), 8 * (1 - 2 / n[3])]
    g, og = mk(gmin, gmax, n, pd)
    data = O.shape_sphere(og, None, 1.5) + 0.02 * np.random.default_rng(12).standard_normal(n)
    y = data.reshape(-1, 1)
    (yt, sbt, _), (yd, sbd, _) = _term_both_kernels(g, L.DoublePendulum4D(g, 1.0), scheme, y, monkeypatch)
    yo, sbo = O.term_lax_friedrichs(og, O.DoublePendulum4D(og, 1.0), scheme, 0., y)
    close(yt, yo, 1e-10, what="tiled")
    close(yd, yo, 1e-10, what="direct")
    close(yt, yd, 1e-12, what="tiled vs direct")
    assert abs(sbt - sbo) <= 1e-12 * sbo and abs(sbd - sbo) <= 1e-12 * sbo


def test_term_split_path_with_foreign_callbacks_matches_fused():
    """A user hamFunc/partialFunc (plain functions -> split path) gives the fused result."""
    g, og = dubins([14, 13, 12])
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    y = (O.shape_cylinder(og, 2, None, .5)).reshape(-1, 1)
    fused, sbf, _ = L.termLaxFriedrichs(0., y, sdata(g, sys_, L.upwindFirstENO3))

    def ham(t, data, derivs, sd):
        return sys_.hamiltonian(t, data, derivs, sd)

    def part(t, data, dmin, dmax, sd, dim):
        return sys_.dissipation(t, data, dmin, dmax, sd, dim)
    sd = L.Bundle(dict(grid=g, hamFunc=ham, partialFunc=part, dissFunc=L.artificialDissipationGLF,
                       derivFunc=L.upwindFirstENO3))        # derivFunc spelling
    split, sbs, _ = L.termLaxFriedrichs(0., y, sd)
    close(split, fused, 1e-12)
    assert abs(sbs - sbf) <= 1e-14
    # device tensors in -> device tensors out, same numbers
    yt = torch.as_tensor(y, device="cuda")
    ft, sbt, _ = L.termLaxFriedrichs(0., yt, sdata(g, sys_, L.upwindFirstENO3))
    assert torch.is_tensor(ft) and ft.is_cuda and ft.shape == yt.shape
    assert np.array_equal(ft.cpu().numpy(), fused)
    st, _, _ = L.termLaxFriedrichs(0., yt, sd)
    close(st.cpu().numpy(), fused, 1e-12)


def test_term_restrict_update_and_missing_fields():
    g, og = dubins([12, 11, 10])
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    y = O.shape_cylinder(og, 2, None, .5).flatten()
    inner = sdata(g, sys_, L.upwindFirstENO2)
    sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=inner, positive=0))
    yd, sb, _ = L.termRestrictUpdate(0., y, sdr)
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), "ENO2", 0., y)
    assert yd.shape == (y.size,)
    close(yd, np.minimum(yo, 0))
    sdr.positive = 1
    yd, _, _ = L.termRestrictUpdate(0., y, sdr)
    close(yd, np.maximum(yo, 0))
    bad = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, CoStateCalc=L.upwindFirstENO2))
    with pytest.raises(AssertionError):
        L.termLaxFriedrichs(0., y.reshape(-1, 1), bad)


# ------------------------------------------------------------------------------ integrators
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_ode_cfl_vs_reference_golden(golden, scheme):
    G = golden("ode.npz")
    g, og = mk(G["dub_min"], G["dub_max"], G["dub_N"], 2)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    sd = sdata(g, sys_, DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y = G["dub_data"].reshape(-1, 1)
    t = 0.
    for k in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
        if k in (0, 4):
            ref_t = float(G["rk3_%s_t%d" % (scheme, k + 1)])
            assert isinstance(t, np.float64) and abs(t - ref_t) <= 1e-13 * ref_t
            if scheme.startswith("WENO"):
                close(y, G["rk3_%s_y%d" % (scheme, k + 1)], 1e-11)
            else:       # 15 substeps from exactly symmetric data: ENO3's flipped ties have spread (see close_eno)
                close_eno(y, G["rk3_%s_y%d" % (scheme, k + 1)], 1e-11, what="rk3 golden %s step %d" % (scheme, k + 1))
    # strict comparison on the noisy initial data (no exact ENO ties), every scheme
    y = G["dubn_data"].reshape(-1, 1)
    t = 0.
    for k in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    assert abs(t - float(G["rk3n_%s_t5" % scheme])) <= 1e-13
    close(y, G["rk3n_%s_y5" % scheme], 1e-11)
    op2 = L.odeCFLset(L.Bundle(dict(factorCFL=.95, singleStep='off')))
    t, y, _ = L.odeCFL2(L.termLaxFriedrichs, [0., 0.02], G["dub_data"].reshape(-1, 1), op2, sd)
    assert abs(t - float(G["rk2_%s_t" % scheme])) <= 1e-13
    cmp = close if scheme.startswith("WENO") else close_eno
    cmp(y, G["rk2_%s_y" % scheme], 1e-11)
    sdr = L.Bundle(dict(innerFunc=L.termLaxFriedrichs, innerData=sd, positive=0))
    t, y, _ = L.odeCFL2(L.termRestrictUpdate, [0., 0.02], G["dub_data"].flatten(), op2, sdr)
    assert y.shape == G["rk2r_%s_y" % scheme].shape
    assert abs(t - float(G["rk2r_%s_t" % scheme])) <= 1e-13
    cmp(y, G["rk2r_%s_y" % scheme], 1e-11)


@pytest.mark.parametrize("ring", ["0", "1"])
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_reference_goldens_through_the_pair_kernel(golden, scheme, ring, monkeypatch):
    """The dominant kernel of the large grids, fused_pair_kernel (two cells per lane; with ring = "1" its 5-plane LDS halo
    ring), only engages by itself from 2.5 M cells up, so the golden comparisons above run fused_substep_kernel.
    Here a slice of them -- termLaxFriedrichs on the Dubins and double-integrator goldens, five odeCFL3 steps and the
    odeCFL2 run -- is repeated with HJ_PAIR=2 (the pair kernel at any size; fresh contexts), and the kernel that ran is
    asserted."""
    monkeypatch.setenv("HJ_PAIR", "2")
    monkeypatch.setenv("HJ_PAIR_RING", ring)
    T, Gd = golden("term.npz"), golden("ode.npz")

    def ran_pair(g):
        dg = device_grid(g, "float64")
        nbuf, ahead = C.c_int(), C.c_int()
        _ffi.check(dg.lib.hj_last_launch(dg.ctx, C.byref(nbuf), C.byref(ahead)))
        assert dg.lib.hj_last_kernel(dg.ctx) == b"fused_pair_kernel"
        # ring: 2 + HJ_PAIR_AH plane buffers (3 ahead by default, 2 on grids below 12 M cells); else the double buffer
        assert (nbuf.value in (4, 5) and ahead.value == nbuf.value - 2) if ring == "1" else (nbuf.value, ahead.value) == (2, 0), (nbuf.value, ahead.value)

    g, og = mk(T["dub_min"], T["dub_max"], T["dub_N"], 2)
    for ub, wb in ((1, 1), (2, 3)):
        yd, sb, _ = L.termLaxFriedrichs(0.3, T["dub_data"].reshape(-1, 1), sdata(g, L.DubinsVehicleRel(g, ub, wb), DERIV[scheme]))
        close(yd, T["dub_u%d_w%d_%s_ydot" % (ub, wb, scheme)])
        ref = float(T["dub_u%d_w%d_%s_sb" % (ub, wb, scheme)])
        assert abs(sb - ref) <= 1e-13 * ref
    ran_pair(g)
    g2, _ = mk(T["di_min"], T["di_max"], T["di_N"], None)
    yd, sb, _ = L.termLaxFriedrichs(0., T["di_data"].reshape(-1, 1), sdata(g2, L.DoubleIntegrator(g2, 2.5), DERIV[scheme]))
    close(yd, T["di_u2.5_%s_ydot" % scheme])
    ran_pair(g2)
    g3, _ = mk(Gd["dub_min"], Gd["dub_max"], Gd["dub_N"], 2)
    sd = sdata(g3, L.DubinsVehicleRel(g3, 1, 1), DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y, t = Gd["dub_data"].reshape(-1, 1), 0.
    for k in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    assert abs(t - float(Gd["rk3_%s_t5" % scheme])) <= 1e-13 * t
    if scheme.startswith("ENO"):
        assert np.array_equal(y, Gd["rk3_%s_y5" % scheme])        # NumPy-order ENO path: bit for bit
    else:
        close(y, Gd["rk3_%s_y5" % scheme], 1e-11)
    op2 = L.odeCFLset(L.Bundle(dict(factorCFL=.95, singleStep='off')))
    t, y, _ = L.odeCFL2(L.termLaxFriedrichs, [0., 0.02], Gd["dub_data"].reshape(-1, 1), op2, sd)
    close(y, Gd["rk2_%s_y" % scheme], 1e-11)
    ran_pair(g3)


def test_eno_paths_bitwise_equal_the_reference_golden(golden):
    """Round 3: with ENO2 / ENO3 the WHOLE fused substep is evaluated in the reference's operation order, contraction
    off (hj_device.h np_order: divided-difference tables and selectors, chosen candidate, Hamiltonian, dissipation sum,
    -(H - diss), the stage expressions of odeCFLn) and the trig tables are NumPy's: the states after five RK3 steps,
    the RK2 runs and t itself equal the golden outputs of the unmodified reference BIT FOR BIT -- on the exactly
    symmetric (tie-prone) data too, where any rounding difference would flip stencil choices.  Recorded in
    gpurun_out/eno_bitwise.txt: the number of differing cells per comparison (0 expected)."""
    G = golden("ode.npz")
    g, og = dubins(G["dub_data"].shape)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    report = []
    for scheme in ("ENO2", "ENO3"):
        sd = sdata(g, sys_, DERIV[scheme])
        op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
        for tag, data in (("rk3", "dub_data"), ("rk3n", "dubn_data")):
            y = G[data].reshape(-1, 1)
            t = 0.
            for k in range(5):
                t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
                key = "%s_%s_y%d" % (tag, scheme, k + 1)
                if key in G:
                    report.append((key, int(np.sum(y != G[key])), float(np.max(np.abs(y - G[key])))))
            tk = "%s_%s_t5" % (tag, scheme)
            report.append((tk, int(t != float(G[tk])), abs(t - float(G[tk]))))
        op2 = L.odeCFLset(L.Bundle(dict(factorCFL=.95, singleStep='off')))
        t, y, _ = L.odeCFL2(L.termLaxFriedrichs, [0., 0.02], G["dub_data"].reshape(-1, 1), op2, sd)
        report.append(("rk2_%s_y" % scheme, int(np.sum(y != G["rk2_%s_y" % scheme])), float(np.max(np.abs(y - G["rk2_%s_y" % scheme])))))
        report.append(("rk2_%s_t" % scheme, int(t != float(G["rk2_%s_t" % scheme])), abs(t - float(G["rk2_%s_t" % scheme]))))
    # the double integrator (2-D, ENO3), symmetric and noisy data
    g2, _ = mk([-1, -1], [1, 1], [32, 32], None)
    sd2 = sdata(g2, L.DoubleIntegrator(g2, 1), L.upwindFirstENO3)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    for data, key in (("di_data", "di_rk3_ENO3_y5"), ("din_data", "din_rk3_ENO3_y5")):
        y = G[data].reshape(-1, 1)
        t = 0.
        for _ in range(5):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd2)
        report.append((key, int(np.sum(y != G[key])), float(np.max(np.abs(y - G[key])))))
    outdir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "eno_bitwise.txt"), "w") as f:
        for name, ndiff, err in report:
            f.write("%-24s differing=%d max_abs=%.3e\n" % (name, ndiff, err))
    bad = [r for r in report if r[1] != 0]
    assert not bad, bad


def test_ode_cfl3_double_integrator_vs_reference_golden(golden):
    G = golden("ode.npz")
    g, og = mk([-1, -1], [1, 1], [32, 32], None)
    sd = sdata(g, L.DoubleIntegrator(g, 1), L.upwindFirstENO3)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y = G["di_data"].reshape(-1, 1)
    t = 0.
    for _ in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    assert abs(t - float(G["di_rk3_ENO3_t5"])) <= 1e-13
    close_eno(y, G["di_rk3_ENO3_y5"], 1e-11)
    y = G["din_data"].reshape(-1, 1)
    t = 0.
    for _ in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    close(y, G["din_rk3_ENO3_y5"], 1e-11)


def test_ode_generic_path_equals_device_path_and_hooks():
    g, og = dubins([13, 12, 11])
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    sd = sdata(g, sys_, L.upwindFirstENO3)
    y0 = O.shape_cylinder(og, 2, None, .5).reshape(-1, 1)
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='off')))
    t1, y1, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 0.03], y0, op, sd)

    def wrapped(t, y, s):        # a foreign schemeFunc -> generic integrator loop
        return L.termLaxFriedrichs(t, y, s)
    t2, y2, _ = L.odeCFL3(wrapped, [0., 0.03], y0, op, sd)
    assert abs(t1 - t2) <= 1e-14 and abs(t1 - 0.03) <= 100 * L.eps * 0.03
    close_eno(y1, y2, 1e-12)
    # RK1 (the reference's is broken; compare with the oracle's intended Euler)
    term = lambda tt, yy: O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), "ENO3", tt, yy)  # noqa: E731
    to, yo = O.ode_cfl_1(term, [0., 0.01], y0, 0.5)
    t3, y3, _ = L.odeCFL1(L.termLaxFriedrichs, [0., 0.01], y0, L.odeCFLset(factorCFL=.5), sd)
    assert abs(t3 - to) <= 1e-14
    close_eno(y3, yo, 1e-11)
    # postTimeStep hook is called once per step with (t, y, schemeData) and may edit y
    calls = []

    def post(t, y, s):
        calls.append(float(t))
        return np.minimum(y, 2.0), s
    opp = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='off', postTimeStep=post)))
    t4, y4, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 0.1], y0, opp, sd)
    assert len(calls) >= 2 and calls[-1] == t4 and y4.max() <= 2.0
    # mixed shapes are rejected instead of silently broadcasting to (N,N)
    with pytest.raises(ValueError):
        L.odeCFL3(L.termLaxFriedrichs, [0., 0.01], y0.flatten(), op, sd)
    # multi-entry tspan returns one row per time
    tt, yy, _ = L.odeCFL2(L.termLaxFriedrichs, [0., 0.005, 0.01], y0, op, sd)
    assert tt.shape == (3, 1) and yy.shape == (3, y0.size)
    # inputs are never mutated
    assert np.array_equal(y0, O.shape_cylinder(og, 2, None, .5).reshape(-1, 1))


def test_cfl_violation_warning_on_device_and_generic_paths(caplog):
    # ode_cfl_3.py:173-175,215-217: warn when deltaT > min(1, 1.2 factorCFL)*stepBound at substeps 2, 3
    import logging
    g, og = dubins([13, 12, 11])
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstENO3)
    y0 = O.shape_cylinder(og, 2, None, .5).reshape(-1, 1)
    for fn in (L.termLaxFriedrichs, lambda t, y, s: L.termLaxFriedrichs(t, y, s)):
        for factor, expect in ((0.8, 0), (1.5, 2)):
            caplog.clear()
            with caplog.at_level(logging.WARNING):
                L.odeCFL3(fn, [0., 1.], y0, L.odeCFLset(L.Bundle(dict(factorCFL=factor, singleStep='on'))), sd)
            n = sum('violated CFL' in r.getMessage() for r in caplog.records)
            assert n == expect, (factor, n)


def test_known_answers_51cubed_all_schemes():
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        KA = json.load(f)
    g, og = dubins(51)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    for scheme in ("ENO2", "ENO3", "WENO5_ASSHIPPED"):
        ka = KA["dubins51_" + scheme]
        sd = sdata(g, sys_, DERIV[scheme])
        y = d0.reshape(-1, 1)
        yd, sb, _ = L.termLaxFriedrichs(0., y, sd)
        assert abs(sb - ka["stepBound0"]) <= 1e-13 * sb
        assert abs(np.linalg.norm(yd) - ka["ydot0_l2"]) <= 1e-11 * ka["ydot0_l2"]
        op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
        t = 0.
        for k in range(5):
            t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
            if k == 0:
                assert abs(t - ka["t1"]) <= 1e-14
        assert abs(t - ka["t5"]) <= 1e-14
        assert abs(y.sum() - ka["sum5"]) <= 1e-10 * abs(ka["sum5"])
        assert abs(np.linalg.norm(y) - ka["l2_5"]) <= 1e-11 * ka["l2_5"]
        assert abs(y.min() - ka["min5"]) <= 1e-11 and abs(y.max() - ka["max5"]) <= 1e-11
    g2, og2 = mk([-1, -1], [1, 1], [128, 128], None)
    ka = KA["dint128_ENO3"]
    sd = sdata(g2, L.DoubleIntegrator(g2, 1), L.upwindFirstENO3)
    y = L.shapeSphere(g2, np.zeros((2, 1)), .25).reshape(-1, 1)
    t = 0.
    for _ in range(5):
        t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t, 10.], y, op, sd)
    assert abs(t - ka["t5"]) <= 1e-14 and abs(np.linalg.norm(y) - ka["l2_5"]) <= 1e-11 * ka["l2_5"]


# ------------------------------------------------------------------------------ HJIPDE_solve
@pytest.mark.parametrize("comp", ["minVOverTime", "maxVOverTime", "set"])
def test_hjipde_solve_vs_reference_golden(golden, comp):
    G = golden("hjipde.npz")
    g, og = mk(G["dub_min"], G["dub_max"], G["dub_N"], 2)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                       CoStateCalc=L.upwindFirstWENO5, uMode='min', dMode='max'))
    extra = L.Bundle(dict(keepLast=True, quiet=True))
    data, tau, _ = L.HJIPDE_solve(G["hj_data0"].copy(), G["hj_tau"].copy(), sd, comp, extra)
    np.testing.assert_array_equal(tau, G["hj_%s_tau" % comp])
    close(data, G["hj_%s_data" % comp], 1e-11)


def test_hjipde_solve_store_all_targets_obstacles():
    g, og = dubins([15, 14, 13])
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    obst = L.shapeSphere(g, np.array([[1.5], [0.], [0.]]), .3)
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation,
                       derivFunc=L.upwindFirstENO3))
    tau = np.array([0., .02, .04])
    data, tau_o, _ = L.HJIPDE_solve(d0, tau, sd, 'minVWithTarget',
                                    L.Bundle(dict(quiet=True, targetFunction=d0, obstacleFunction=obst)))
    assert data.shape == (3,) + g.shape and np.array_equal(data[0], d0)
    # oracle replay of the same loop
    osys = O.DubinsRel(og, 1, 1)
    term = lambda tt, yy: O.term_lax_friedrichs(og, osys, "ENO3", tt, yy)  # noqa: E731
    y = d0.reshape(-1, 1)
    for i in (1, 2):
        tn = tau[i - 1]
        while tn < tau[i] - 1e-4:
            tn, y = O.ode_cfl_3(term, [tn, tau[i]], y, 0.8, single_step=True)
            y = np.minimum(y, d0.reshape(-1, 1))
            y = np.maximum(y, -obst.reshape(-1, 1))
        close_eno(data[i], y.reshape(g.shape), 1e-11)
    with pytest.raises(ValueError):
        L.HJIPDE_solve(d0, tau, sd, 'minVWithTarget', L.Bundle(dict(quiet=True)))
    with pytest.raises(ValueError):
        L.HJIPDE_solve(d0, tau, sd, 'nonsense', L.Bundle(dict(quiet=True)))


def test_nan_guard():
    g, og = dubins([9, 9, 9])
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    d0[4, 4, 4] = np.nan
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, derivFunc=L.upwindFirstENO2))
    with pytest.raises(ValueError, match="Nans"):
        L.HJIPDE_solve(d0, np.array([0., .01]), sd, 'set', L.Bundle(dict(quiet=True, keepLast=True)))


# ------------------------------------------------------------------------------ C ABI direct
def test_cabi_errors_and_rk_substep_ranges():
    g, og = dubins([20, 11, 12])
    dg = device_grid(g)
    dg.bind_stream()
    lib = dg.lib
    par = _ffi.darr([1, 1, 1, 2])
    y = dg.to_device(O.shape_cylinder(og, 2, None, .5))
    out, out2 = dg.empty(), dg.empty()
    # aliasing / bad arguments are refused with a message
    assert lib.hj_rk_substep(dg.ctx, _ffi.ENO3, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_EULER, 1e-3, 0,
                             dg.ptr(y), None, dg.ptr(y), 0, 0, 20) == -1
    assert b"alias" in lib.hj_last_error()
    assert lib.hj_rk_substep(dg.ctx, 9, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_EULER, 1e-3, 0,
                             dg.ptr(y), None, dg.ptr(out), 0, 0, 20) == -1
    assert lib.hj_rk_substep(dg.ctx, _ffi.ENO3, _ffi.HAM_DOUBLE_INTEGRATOR, par, 0., _ffi.STAGE_EULER, 1e-3, 0,
                             dg.ptr(y), None, dg.ptr(out), 0, 0, 20) == -1
    assert lib.hj_rk_substep(dg.ctx, _ffi.ENO3, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_RK3_HALF, 1e-3, 0,
                             dg.ptr(y), None, dg.ptr(out), 0, 0, 20) == -1
    # plane ranges: [0,7) + [7,20) == [0,20)
    _ffi.check(lib.hj_rk_substep(dg.ctx, _ffi.ENO3, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_EULER, 1e-3, 0,
                                 dg.ptr(y), None, dg.ptr(out), 0, 0, 20))
    out2.zero_()
    _ffi.check(lib.hj_rk_substep(dg.ctx, _ffi.ENO3, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_EULER, 1e-3, 0,
                                 dg.ptr(y), None, dg.ptr(out2), 1, 0, 7))
    _ffi.check(lib.hj_rk_substep(dg.ctx, _ffi.ENO3, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_EULER, 1e-3, 0,
                                 dg.ptr(y), None, dg.ptr(out2), 2, 7, 20))
    dg.sync()
    assert torch.equal(out, out2)
    sb, am = C.c_double(), (C.c_double * 4)()
    _ffi.check(lib.hj_read_step_bound(dg.ctx, 0, C.byref(sb), am))
    sbs = C.c_double()
    _ffi.check(lib.hj_static_step_bound(dg.ctx, _ffi.HAM_DUBINS_REL, par, C.byref(sbs), None))
    assert sb.value == sbs.value
    _, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), "ENO3", 0., O.shape_cylinder(og, 2, None, .5).reshape(-1, 1))
    assert abs(sb.value - sbo) <= 1e-13 * sbo


# ------------------------------------------------------------------------------ slab decomposition
@pytest.mark.parametrize("scheme", ["ENO3", "WENO5"])
@pytest.mark.parametrize("periodic0", [False, True])
def test_slab_halo_mode_equals_single_domain(scheme, periodic0):
    """Two slabs of axis 0 with exchanged ghost planes == the undivided grid (bitwise)."""
    n = (23, 14, 12)
    gmin, gmax = [-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])]
    pd = (0, 2) if periodic0 else 2
    g, og = mk(gmin, gmax, n, pd)
    rng = np.random.default_rng(11)
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * rng.standard_normal(n)
    sid = _ffi.SCHEME_IDS[scheme]
    par = _ffi.darr([1, 1, 1, 2])
    dg = device_grid(g)
    dg.bind_stream()
    y = dg.to_device(data)
    ref = dg.empty()
    _ffi.check(dg.lib.hj_rk_substep(dg.ctx, sid, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_EULER, 2e-3, 0,
                                    dg.ptr(y), None, dg.ptr(ref), 0, 0, n[0]))
    eps_dev = None
    if scheme == "WENO5":
        eps_dev = torch.empty(3, dtype=torch.float64, device="cuda")
        _ffi.check(dg.lib.hj_max_d1sq(dg.ctx, dg.ptr(y), dg.ptr(eps_dev)))
    dg.sync()
    cut = 10
    pieces = []
    for (b, e) in ((0, cut), (cut, n[0])):
        lo = (b > 0) or periodic0
        hi = (e < n[0]) or periodic0
        sdg = DeviceGrid(g, "float64", None, (b, e, lo, hi))
        sdg.bind_stream()
        buf = torch.zeros((e - b + 6,) + n[1:], dtype=torch.float64, device="cuda")
        buf[3:3 + e - b] = y[b:e]
        if lo:
            buf[0:3] = y[[(b - 3 + k) % n[0] for k in range(3)]]
        if hi:
            buf[3 + e - b:] = y[[(e + k) % n[0] for k in range(3)]]
        outb = torch.zeros_like(buf)
        if eps_dev is not None:
            _ffi.check(sdg.lib.hj_ctx_set_weno_eps_source(sdg.ctx, sdg.ptr(eps_dev)))
        _ffi.check(sdg.lib.hj_rk_substep(sdg.ctx, sid, _ffi.HAM_DUBINS_REL, par, 0., _ffi.STAGE_EULER, 2e-3, 0,
                                         C.c_void_p(buf[3:].data_ptr()), None, C.c_void_p(outb[3:].data_ptr()),
                                         0, 0, e - b))
        sdg.sync()
        pieces.append(outb[3:3 + e - b])
    got = torch.cat(pieces, 0)
    assert torch.equal(got, ref)
    yo, _ = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), scheme, 0., data.reshape(-1, 1))
    close(ref.cpu().numpy(), data + 2e-3 * yo.reshape(n), 1e-11)


# ------------------------------------------------------------------------------ fp32
def test_fp32_path_vs_fp64_oracle():
    g, og = dubins([24, 22, 20])
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    data = O.shape_cylinder(og, 2, None, .5)
    yt = torch.as_tensor(data.reshape(-1, 1), device="cuda", dtype=torch.float32)
    yd, sb, _ = L.termLaxFriedrichs(0., yt, sdata(g, sys_, L.upwindFirstWENO5Intended))
    assert yd.dtype == torch.float32
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), "WENO5", 0., data.reshape(-1, 1))
    rel = np.max(np.abs(yd.cpu().numpy().astype(np.float64) - yo)) / np.max(np.abs(yo))
    assert rel <= 1e-4, rel
    assert abs(sb - sbo) <= 1e-6 * sbo


# ------------------------------------------------------------------------------ slab integrator (HIP backend)
class _LocalTransport(object):
    """Two in-process 'ranks' (threads, one CUDA stream each) standing in for two GPUs: halo
    exchange and all-reduce through shared tensors + a thread barrier.  Exercises HipSlabBackend
    and SlabIntegrator (edge-first ordering, side-stream hand-off) on real kernels; the RCCL
    transport itself is covered by the driver's multi-GPU run and the gloo CPU test."""

    def __init__(self, world):
        import threading
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
            tr.bar.wait()
            m = torch.maximum(tr.box[(0, "ar")], tr.box[(1, "ar")])
            tr.bar.wait()
            t.copy_(m)
        return f


@pytest.mark.parametrize("scheme,periodic0", [("WENO5_ASSHIPPED", False), ("WENO5", True), ("ENO3", False)])
def test_slab_integrator_hip_backend_two_virtual_ranks(scheme, periodic0):
    import threading
    from levelsetpy_amd.dist import SlabDecomposition, SlabIntegrator, HipSlabBackend
    n = (26, 15, 14)
    pd = (0, 2) if periodic0 else 2
    g, og = mk([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * np.random.default_rng(13).standard_normal(n)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    sd = sdata(g, sys_, DERIV[scheme])
    op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    y, t_ref = data.reshape(-1, 1), 0.
    for _ in range(3):
        t_ref, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t_ref, 10.], y, op, sd)
    tr = _LocalTransport(2)
    out, errs = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                slab = SlabDecomposition(n[0], 2, rank, periodic0)
                be = HipSlabBackend(g, slab, _ffi.SCHEME_IDS[scheme], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.])
                integ = SlabIntegrator(slab, be, [float(v) for v in np.asarray(g.dx).ravel()], 3, 0.8,
                                       needs_eps=(scheme == "WENO5"), exchanger=tr.exchanger(slab),
                                       allreduce_max=tr.allreduce_max(rank))
                integ.set_state(torch.as_tensor(data[slab.begin:slab.end], device="cuda"))
                t = 0.
                for _ in range(3):
                    t, _dt = integ.step(t)
                be.sync()
                out[rank] = (slab.begin, slab.end, t, integ.state().cpu().numpy())
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            tr.bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join(120)
    assert not errs, errs
    got = np.zeros(n)
    for r in range(2):
        b, e, t, ys = out[r]
        got[b:e] = ys
        assert abs(t - t_ref) <= 1e-14
    (close if scheme.startswith("WENO") else close_eno)(got, y.reshape(n), 1e-12)


def test_native_rccl_slab_stepper_self_ring():
    """hj_slab_rk_step (ncclSend/ncclRecv inside the C library, edge planes + exchange on a second
    stream) on ONE GPU: a periodic axis 0 closed through a self send/recv must reproduce the
    in-kernel periodic wrap.  Exercises the real RCCL transport, streams and events."""
    import torch.distributed as dist
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29591")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        n = (40, 18, 16)
        g, og = mk([-2., -1.25, -np.pi], [2. * (1 - 2 / n[0]), 1.25, np.pi * (1 - 2 / n[2])], n, (0, 2))
        data = O.shape_cylinder(og, 2, None, .5) + 0.1 * np.sin(3 * og.xs[0])
        for scheme in ("WENO5_ASSHIPPED", "WENO5", "ENO3"):
            sys_ = L.DubinsVehicleRel(g, 1, 1)
            sd = sdata(g, sys_, DERIV[scheme])
            op = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
            y, t_ref = data.reshape(-1, 1), 0.
            for _ in range(4):
                t_ref, y, _ = L.odeCFL3(L.termLaxFriedrichs, [t_ref, 10.], y, op, sd)
            # one 9-plane exchange per step / one 3-plane exchange per substep in each of its three stream schedules
            # (HJ_SLAB_SCHEDULE is read when the stepper creates its context; the default depends on the slab thickness)
            for deep, sched in ((True, None), (False, "overlap"), (False, "overlap2"), (False, "serial"), (False, "gated")):
                if sched is None:
                    os.environ.pop("HJ_SLAB_SCHEDULE", None)
                else:
                    os.environ["HJ_SLAB_SCHEDULE"] = sched
                try:
                    slab = SlabDecomposition(n[0], 1, 0, True, self_exchange=True)
                    nat = NativeSlabStepper(g, slab, _ffi.SCHEME_IDS[scheme], _ffi.HAM_DUBINS_REL, [1., 1., 1., 2.],
                                            [float(v) for v in np.asarray(g.dx).ravel()], deep=deep)
                finally:
                    os.environ.pop("HJ_SLAB_SCHEDULE", None)
                nat.set_state(torch.as_tensor(data, device="cuda"))
                t = 0.
                for _ in range(4):
                    t, _dt = nat.step(t)
                got = nat.state().cpu().numpy()
                nat.close()
                assert abs(t - t_ref) <= 1e-14
                (close if scheme.startswith("WENO") else close_eno)(got, y.reshape(n), 1e-12, what="%s deep=%s %s" % (scheme, deep, sched))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("scheme,periodic0,world", [("WENO5_ASSHIPPED", False, 3), ("ENO3", False, 2),
                                                    ("WENO5_ASSHIPPED", True, 3), ("ENO2", True, 2),
                                                    ("ENO3", True, 5), ("WENO5_ASSHIPPED", False, 5)])
def test_deep_halo_stepper_virtual_ranks_bitwise(scheme, periodic0, world, order):
    """hj_slab_rk_step_deep with `world` virtual ranks in ONE process (hj_comm_init_external: the test moves
    the pad planes itself): end ranks with a single neighbour, middle ranks with two, the periodic ring.
    Stages recompute planes beyond the slab; the result must equal the undivided grid BITWISE."""
    from levelsetpy_amd.dist import SlabDecomposition, NativeSlabStepper
    n = (61 if world <= 3 else 97, 18, 16)      # 97 = 5*19 + 2: uneven slabs of 20 and 19 planes
    pd = (0, 2) if periodic0 else 2
    gmax0 = 2. * (1 - 2 / n[0]) if periodic0 else 2.
    g, og = mk([-2., -1.25, -np.pi], [gmax0, 1.25, np.pi * (1 - 2 / n[2])], n, pd)
    rng = np.random.default_rng(5)
    data = O.shape_cylinder(og, 2, None, .5) + 0.1 * np.sin(3 * og.xs[0]) + 0.01 * rng.standard_normal(n)
    dxs = [float(v) for v in np.asarray(g.dx).ravel()]
    par = [1., 1., 1., 2.]
    sid = _ffi.SCHEME_IDS[scheme]
    # undivided reference: the same fused kernels, whole grid, same dt
    dg = DeviceGrid(g)
    dg.bind_stream()
    steppers = []
    for r in range(world):
        slab = SlabDecomposition(n[0], world, r, periodic0)
        steppers.append(NativeSlabStepper(g, slab, sid, _ffi.HAM_DUBINS_REL, par, dxs, order=order, deep=True,
                                          external=lambda st: None))
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
    cur, nxt, w0, w1 = full.clone(), torch.empty_like(full), torch.empty_like(full), torch.empty_like(full)
    tout, dtout = C.c_double(), C.c_double()
    t = t_ref = 0.
    for _ in range(3):
        ts = [st.step(t) for st in steppers]
        move_pads()
        assert all(abs(a[0] - ts[0][0]) == 0 for a in ts)
        dt = ts[0][1]
        t = ts[0][0]
        _ffi.check(dg.lib.hj_rk_step(dg.ctx, order, sid, _ffi.HAM_DUBINS_REL, _ffi.darr(par), t_ref, 1e9, 0.8, dt, 0,
                                     dg.ptr(cur), dg.ptr(nxt), dg.ptr(w0), dg.ptr(w1), C.byref(tout), C.byref(dtout)))
        cur, nxt = nxt, cur
        t_ref = float(tout.value)
        assert dtout.value == dt and abs(t_ref - t) <= 1e-15
    torch.cuda.synchronize()
    for st in steppers:
        got = st.state()
        ref = cur[st.slab.begin:st.slab.end]
        assert torch.equal(got, ref), "rank %d differs by %g" % (st.slab.rank, float((got - ref).abs().max()))
        st.close()


# ------------------------------------------------------------------------------ BASELINE sizes: properties
def _substep(dg, scheme, ham, par, stage, dt, y, y0, out, p0=0, p1=None, slot=3):
    p1 = dg.shape[0] if p1 is None else p1
    _ffi.check(dg.lib.hj_rk_substep(dg.ctx, _ffi.SCHEME_IDS[scheme], ham, _ffi.darr(par), 0., stage, dt, 0,
                                    dg.ptr(y), dg.ptr(y0) if y0 is not None else None, dg.ptr(out), slot, p0, p1))


@pytest.mark.parametrize("scheme", ["WENO5_ASSHIPPED", "WENO5"])
def test_full_size_201_cubed_properties(scheme, monkeypatch):
    """BASELINE C2 (201^3 Dubins, fp64), too big for the NumPy oracle in a test: size-independent
    properties instead.  (1) the tiled kernel and the independent direct kernel agree to rounding;
    (2) computing the grid as two plane ranges equals one launch bitwise; (3) after 3 RK3 steps from
    the z-invariant cylinder the state is still z-invariant wherever the Hamiltonian is (it is not: the
    dynamics depend on x3) -- instead check the reflection symmetry x2 -> -x2, x3 -> -x3 of the Dubins
    problem on the symmetric grid rows; (4) stepBound equals the closed form 1/sum(max alpha/dx)."""
    n = 201
    g, og = dubins(n)
    d0 = L.shapeCylinder(g, 2, np.zeros((3, 1)), .5)
    par = [1., 1., 1., 2.]
    outs = {}
    for force in ("0", "1"):
        monkeypatch.setenv("HJ_FORCE_DIRECT", force)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        y = dg.to_device(d0)
        a, b, c = dg.empty(), dg.empty(), dg.empty()
        _substep(dg, scheme, _ffi.HAM_DUBINS_REL, par, _ffi.STAGE_EULER, 2e-3, y, None, a)
        _substep(dg, scheme, _ffi.HAM_DUBINS_REL, par, _ffi.STAGE_RK3_HALF, 2e-3, a, y, b)
        _substep(dg, scheme, _ffi.HAM_DUBINS_REL, par, _ffi.STAGE_RK3_FULL, 2e-3, b, y, c)
        if force == "0":
            # (2) plane-range split of the last substep
            c2 = torch.zeros_like(c)
            _substep(dg, scheme, _ffi.HAM_DUBINS_REL, par, _ffi.STAGE_RK3_FULL, 2e-3, b, y, c2, 0, 77, slot=4)
            _substep(dg, scheme, _ffi.HAM_DUBINS_REL, par, _ffi.STAGE_RK3_FULL, 2e-3, b, y, c2, 77, n, slot=5)
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
        outs[force] = c.cpu().numpy()
    assert np.array_equal(outs["0"], outs["1"]), np.max(np.abs(outs["0"] - outs["1"]))     # round 3: bit for bit
    assert np.isfinite(outs["0"]).all()
    # (3) H(x1, -x2, -x3; p1, -p2, -p3) = H(x; p) and the cylinder is even in x2: the solution stays
    # even under (x2, x3) -> (-x2, -x3).  x2 nodes are symmetric; x3 = -pi + k*dx3 maps k -> n-k (mod n).
    u = outs["0"]
    k = np.arange(n)
    mirror = u[:, ::-1, :][:, :, (n - k) % n]
    assert np.max(np.abs(u - mirror)) <= 1e-10


def test_full_size_4096_squared_tiled_vs_direct(monkeypatch):
    """BASELINE C3 (double integrator, 4096^2, ENO3): tiled vs direct kernel on one substep, and the
    CFL bound against its closed form 1/(max|x2|/dx1 + u/dx2)."""
    n = 4096
    g = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), n * np.ones((2, 1), dtype=np.int64), None, low_mem=True)
    d0 = L.shapeSphere(g, np.zeros((2, 1)), .25) + 0.01 * np.random.default_rng(0).standard_normal((n, n))
    outs = {}
    for force in ("0", "1"):
        monkeypatch.setenv("HJ_FORCE_DIRECT", force)
        dg = DeviceGrid(g, "float64")
        dg.bind_stream()
        y = dg.to_device(d0)
        a = dg.empty()
        _substep(dg, "ENO3", _ffi.HAM_DOUBLE_INTEGRATOR, [1.5, 0, 0, 0], _ffi.STAGE_EULER, 1e-4, y, None, a)
        sb = C.c_double()
        _ffi.check(dg.lib.hj_read_step_bound(dg.ctx, 3, C.byref(sb), None))
        dx = np.asarray(g.dx).ravel()
        assert abs(sb.value - 1 / (1.0 / dx[0] + 1.5 / dx[1])) <= 1e-13 * sb.value
        outs[force] = a.cpu().numpy()
    assert np.max(np.abs(outs["0"] - outs["1"])) <= 1e-11


def test_native_integrate_loop_equals_stepwise():
    """hj_rk_integrate (the odeCFLn loop of a whole span in one C call) against the same span stepped
    through Python with a no-op postTimeStep hook: identical t and state, inputs untouched."""
    g, og = dubins([24, 20, 18])
    sd = sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5)
    y0 = (O.shape_cylinder(og, 2, None, .5) + 0.01 * np.random.default_rng(3).standard_normal(g.shape)).reshape(-1, 1)
    keep = y0.copy()
    for order, fn in ((3, L.odeCFL3), (2, L.odeCFL2), (1, L.odeCFL1)):
        a = fn(L.termLaxFriedrichs, [0., 0.07], y0, L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='off'))), sd)
        b = fn(L.termLaxFriedrichs, [0., 0.07], y0,
               L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='off', postTimeStep=lambda t, y, s: (y, s)))), sd)
        assert a[0] == b[0] and abs(a[0] - 0.07) <= 100 * L.eps * 0.07
        assert np.array_equal(a[1], b[1]), order
    assert np.array_equal(y0, keep)
    # a span that is already over takes no step and hands back a copy of the input
    t, y, _ = L.odeCFL3(L.termLaxFriedrichs, [0.5, 0.5], y0, L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='off'))), sd)
    assert t == 0.5 and np.array_equal(y, y0)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("n", [(3, 3, 3), (4, 3, 6), (5, 7, 3), (3, 40, 4), (6, 5, 33)])
def test_term_tiny_extents_vs_oracle(scheme, n, monkeypatch):
    """Extents down to the stencil width (3 nodes): every stencil then reaches ghost cells on both sides
    (extrapolated: ghosts built from the same two edge nodes; periodic: wrapped more than once is not
    needed since width 3 <= N)."""
    g, og = dubins(n)
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * np.random.default_rng(21).standard_normal(g.shape)
    y = data.reshape(-1, 1)
    (yt, sbt, _), (yd, sbd, _) = _term_both_kernels(g, L.DubinsVehicleRel(g, 1, 2), scheme, y, monkeypatch)
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 2), scheme, 0., y)
    close(yt, yo, what="tiled")
    close(yd, yo, what="direct")
    assert np.array_equal(yt, yd) and sbt == sbd        # round 3: the two kernels agree bit for bit
    assert abs(sbt - sbo) <= 1e-13 * sbo and abs(sbd - sbo) <= 1e-13 * sbo


def test_grid_below_stencil_width_is_rejected():
    g, og = dubins((2, 5, 5))
    y = np.zeros((50, 1))
    with pytest.raises(ValueError):
        L.termLaxFriedrichs(0., y, sdata(g, L.DubinsVehicleRel(g, 1, 1), L.upwindFirstWENO5))


@pytest.mark.parametrize("scheme", ["ENO2", "WENO5_ASSHIPPED", "WENO5"])
def test_term_toward_zero_ghost_data_in_fused_path(scheme, monkeypatch):
    """grid.bdryData[i].towardZero (add_ghost_extrapolate.py:60-64) must reach the fused kernels' ghost
    synthesis: slopes point towards zero on axes 0 and 1."""
    n = (17, 14, 12)
    g, _ = dubins(n)
    g.bdryData[0] = L.Bundle(dict(towardZero=True))
    g.bdryData[1] = L.Bundle(dict(towardZero=True))
    og = O.Grid([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n[2])], n, [2], toward_zero=[True, True, False])
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * np.random.default_rng(4).standard_normal(n)
    y = data.reshape(-1, 1)
    (yt, sbt, _), (yd, sbd, _) = _term_both_kernels(g, L.DubinsVehicleRel(g, 1, 1), scheme, y, monkeypatch)
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 1), scheme, 0., y)
    close(yt, yo, what="tiled")
    close(yd, yo, what="direct")
    # and it differs from the default (away-from-zero) ghosts, i.e. the flag is not ignored
    g2, og2 = dubins(n)
    y2, _, _ = L.termLaxFriedrichs(0., y, sdata(g2, L.DubinsVehicleRel(g2, 1, 1), DERIV[scheme]))
    assert np.max(np.abs(np.asarray(y2) - np.asarray(yt))) > 1e-6


# ------------------------------------------------------------------------------ HJIPDE_solve: stopping sets, discounting
def _solve_setup(n=(21, 19, 16)):
    g, og = dubins(n)
    sys_ = L.DubinsVehicleRel(g, 1, 1)
    sd = L.Bundle(dict(grid=g, hamFunc=sys_.hamiltonian, partialFunc=sys_.dissipation, accuracy='high',
                       dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO2))
    data0 = O.shape_cylinder(og, 2, None, .5)
    return g, og, sd, data0


def test_hjipde_solve_stop_init_and_stop_sets():
    from levelsetpy_amd.hji_solver import _eval_point
    g, og, sd, data0 = _solve_setup()
    tau = np.linspace(0, 0.6, 13)
    ex = L.Bundle(dict(quiet=True))
    full, _, _ = L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', ex)
    # a state outside the initial set that the backward reachable tube swallows (picked from the run)
    swallowed = (full[0] > 0.02) & (full[-1] < -0.02)
    assert swallowed.any(), "the tube did not grow: min change %g" % float((full[-1] - full[0]).min())
    node = np.unravel_index(int(np.argmax(np.where(swallowed, full[0], -np.inf))), g.shape)
    x = np.array([float(np.asarray(g.vs[d]).ravel()[node[d]]) for d in range(3)])
    vals = [_eval_point(g, full[i], x) for i in range(len(tau))]
    assert vals[0] > 0 and vals[-1] < 0
    first = next(i for i, v in enumerate(vals) if v <= 0)
    d, t, out = L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, stopInit=x)))
    assert out.stoptau == tau[first] and len(t) == first + 1 and d.shape[0] == first + 1
    assert np.array_equal(d, full[:first + 1])
    # stop sets: the tube at tau[6] as the set; 'intersect' fires as soon as one of its nodes is inside
    # (at once: it contains the initial set), 'include' when all are (the tube is monotone: at tau[6])
    ball = np.where(full[6] <= 0, -1., 1.)
    di, ti, oi = L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, stopSetIntersect=ball)))
    da, ta, oa = L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, stopSetInclude=ball)))
    ki = next(i for i in range(1, len(tau)) if np.any((full[i] <= 0)[ball < 0]))
    ka = next(i for i in range(1, len(tau)) if np.all((full[i] <= 0)[ball < 0]))
    assert (ki, ka) == (1, 6)
    assert len(ti) == ki + 1 and len(ta) == ka + 1 and oi.stoptau == tau[ki] and oa.stoptau == tau[ka]
    assert np.array_equal(da, full[:ka + 1])
    # stopLevel moves the level that is tested
    kl = next(i for i in range(1, len(tau)) if np.all((full[i] <= 0.2)[ball < 0]))
    dl, tl, ol = L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, stopSetInclude=ball, stopLevel=0.2)))
    assert len(tl) == kl + 1 and kl < ka
    with pytest.raises(ValueError):
        L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, stopSetInclude=ball[:-1])))
    with pytest.raises(ValueError):
        L.HJIPDE_solve(data0, tau, sd, 'minVOverTime', L.Bundle(dict(quiet=True, stopInit=[0., 0.])))


def test_hjipde_solve_discounting_flip_and_anneal():
    g, og, sd, data0 = _solve_setup()
    target = data0 + 0.1 * np.sin(2 * og.xs[0])
    # tau steps shorter than one CFL step: exactly one integrator step (and one discount) per interval
    tau = np.array([0., 0.002, 0.004])
    plain, _, _ = L.HJIPDE_solve(data0, tau[:2], sd, 'set', L.Bundle(dict(quiet=True, keepLast=True)))
    gam = 0.9
    # default ('Jaime') mode: y <- gamma*y + (1-gamma)*data0, or the target when one is given
    d1, _, _ = L.HJIPDE_solve(data0, tau[:2], sd, 'set', L.Bundle(dict(quiet=True, keepLast=True, discountFactor=gam)))
    close(d1, gam * plain + (1 - gam) * data0, 1e-13)
    pl, _, _ = L.HJIPDE_solve(data0, tau[:2], sd, 'minVWithL', L.Bundle(dict(quiet=True, keepLast=True, targetFunction=target)))
    d2, _, _ = L.HJIPDE_solve(data0, tau[:2], sd, 'minVWithL',
                              L.Bundle(dict(quiet=True, keepLast=True, targetFunction=target, discountFactor=gam)))
    close(d2, gam * pl + (1 - gam) * target, 1e-13)
    # 'Kene' mode: shift below zero by max|l|, discount, min with the shifted target, shift back
    d3, _, _ = L.HJIPDE_solve(data0, tau[:2], sd, 'minVWithL',
                              L.Bundle(dict(quiet=True, keepLast=True, targetFunction=target, discountFactor=gam, discountMode='Kene')))
    M = np.max(np.abs(target))
    close(d3, np.minimum(gam * (plain - M), target - M) + M, 1e-13)
    with pytest.raises(ValueError):
        L.HJIPDE_solve(data0, tau[:2], sd, 'set', L.Bundle(dict(quiet=True, discountFactor=gam, discountMode='Kene')))
    # flipOutput reverses the stored time axis
    fa, _, _ = L.HJIPDE_solve(data0, tau, sd, 'set', L.Bundle(dict(quiet=True)))
    fb, _, _ = L.HJIPDE_solve(data0, tau, sd, 'set', L.Bundle(dict(quiet=True, flipOutput=True)))
    assert np.array_equal(fb, fa[::-1])
    # convergence stop; with annealing the first "convergence" only raises the discount factor
    big = L.Bundle(dict(quiet=True, stopConverge=True, convergeThreshold=1e9, discountFactor=gam))
    _, t1, o1 = L.HJIPDE_solve(data0, tau, sd, 'set', big)
    assert len(t1) == 2 and o1.stoptau == tau[1]
    big.discountAnneal = 'hard'
    _, t2, o2 = L.HJIPDE_solve(data0, tau, sd, 'set', big)
    assert len(t2) == 3 and o2.stoptau == tau[2]
    # ignoreBoundary: the change is measured 4 cells inside the grid
    ib = L.Bundle(dict(quiet=True, stopConverge=True, convergeThreshold=1e9, ignoreBoundary=True))
    _, t3, _ = L.HJIPDE_solve(data0, tau, sd, 'set', ib)
    assert len(t3) == 2


# ------------------------------------------------------------------------------ local Lax-Friedrichs variants
class _BurgersLike(object):
    """H = sum p_i^2 / 2 with alpha_i = max(|p_i| over the costate range handed in): a DATA-DEPENDENT
    alpha, so the three Lax-Friedrichs variants really differ (plain callbacks -> split path)."""

    @staticmethod
    def hamiltonian(t, data, derivs, sd):
        return sum(0.5 * p * p for p in derivs)

    @staticmethod
    def dissipation(t, data, derivMin, derivMax, sd, dim):
        lo, hi = derivMin[dim], derivMax[dim]
        if torch.is_tensor(lo) or torch.is_tensor(hi):
            lo = lo if torch.is_tensor(lo) else torch.full_like(hi, float(lo))
            hi = hi if torch.is_tensor(hi) else torch.full_like(lo, float(hi))
            return torch.maximum(lo.abs(), hi.abs())
        return np.maximum(np.abs(lo), np.abs(hi))


@pytest.mark.parametrize("kind", ["llf", "lllf"])
@pytest.mark.parametrize("scheme", ["ENO2", "WENO5_ASSHIPPED"])
def test_local_lax_friedrichs_split_path_vs_oracle(kind, scheme):
    g, og = dubins((19, 17, 15))
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * np.random.default_rng(9).standard_normal(g.shape)
    y = data.reshape(-1, 1)
    fn = {"llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sd = L.Bundle(dict(grid=g, hamFunc=_BurgersLike.hamiltonian, partialFunc=_BurgersLike.dissipation,
                       dissFunc=fn, CoStateCalc=DERIV[scheme]))
    yd, sb, _ = L.termLaxFriedrichs(0., y, sd)
    yo, sbo = O.term_lax_friedrichs(og, _BurgersLike, scheme, 0., y, diss=kind)
    close(yd, yo)
    assert abs(sb - sbo) <= 1e-13 * sbo
    # and they are not the global variant in disguise
    yg, sbg = O.term_lax_friedrichs(og, _BurgersLike, scheme, 0., y)
    assert np.max(np.abs(yg - yo)) > 1e-6 and sbo > sbg


@pytest.mark.parametrize("kind", ["llf", "lllf"])
def test_local_lax_friedrichs_fused_native(kind):
    """Native Hamiltonian: alpha ignores the costate range, so the term equals the GLF one and only the CFL
    bound changes (1/max_x sum_i alpha_i/dx_i); the integrator steps with it."""
    g, og = dubins((18, 16, 14))
    data = O.shape_cylinder(og, 2, None, .5) + 0.02 * np.random.default_rng(10).standard_normal(g.shape)
    y = data.reshape(-1, 1)
    fn = {"llf": L.artificialDissipationLLF, "lllf": L.artificialDissipationLLLF}[kind]
    sys_ = L.DubinsVehicleRel(g, 1, 2)
    sd = sdata(g, sys_, L.upwindFirstWENO5)
    sd.dissFunc = fn
    yd, sb, _ = L.termLaxFriedrichs(0., y, sd)
    yo, sbo = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 2), "WENO5_ASSHIPPED", 0., y, diss=kind)
    _, sbg = O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 2), "WENO5_ASSHIPPED", 0., y)
    close(yd, yo)
    assert abs(sb - sbo) <= 1e-13 * sbo and sbo > sbg
    term = lambda tt, yy: O.term_lax_friedrichs(og, O.DubinsRel(og, 1, 2), "WENO5_ASSHIPPED", tt, yy, diss=kind)  # noqa: E731
    to, yo3 = O.ode_cfl_3(term, [0., 0.05], y, 0.8)
    t3, y3, _ = L.odeCFL3(L.termLaxFriedrichs, [0., 0.05], y, L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='off'))), sd)
    assert abs(t3 - to) <= 1e-14
    close(y3, yo3)
    # the ctx goes back to the global bound for a GLF problem on the same grid
    sd2 = sdata(g, sys_, L.upwindFirstWENO5)
    _, sb2, _ = L.termLaxFriedrichs(0., y, sd2)
    assert abs(sb2 - sbg) <= 1e-13 * sbg


def test_compute_gradients_vs_oracle():
    """computeGradients (compute_gradients.py:11): central/left/right costates of a stored value function,
    single time and time-stacked, NaN/inf kept where they were, dims mask, NumPy and device input."""
    g, og = dubins((14, 13, 12))
    rng = np.random.default_rng(6)
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * rng.standard_normal(g.shape)
    dC, dL, dR = L.computeGradients(g, data)
    for i in range(3):
        oL, oR = O.upwind_first_weno5(og, data, i)
        close(dL[i], oL)
        close(dR[i], oR)
        close(dC[i], 0.5 * (oL + oR))
    # another scheme, a dims mask, a device tensor in -> tensors out
    dC2, dL2, dR2 = L.computeGradients(g, torch.as_tensor(data, device="cuda"), dims=[True, False, True],
                                       derivFunc=L.upwindFirstENO3)
    assert np.ndim(dC2[1]) == 0 and torch.is_tensor(dC2[0]) and dC2[0].is_cuda   # skipped dims stay empty cells
    oL, oR = O.upwind_first_eno3(og, data, 2)
    close(dC2[2].cpu().numpy(), 0.5 * (oL + oR))
    # time-stacked (time first), with a NaN and an inf: they come back in place, neighbours are finite
    stack = np.stack([data, data + 0.1, data * 1.5])
    stack[1, 3, 4, 5] = np.nan
    stack[2, 6, 7, 8] = np.inf
    keep = stack.copy()
    sC, sL, sR = L.computeGradients(g, stack)
    assert np.array_equal(np.isnan(stack), np.isnan(keep)) and np.array_equal(stack[np.isfinite(keep)], keep[np.isfinite(keep)])
    assert sC[0].shape == stack.shape
    assert np.isnan(sC[0][1, 3, 4, 5]) and np.isinf(sC[2][2, 6, 7, 8])
    assert np.isfinite(sC[0][1, 4, 4, 5]) and np.isfinite(sC[0][0]).all()
    oL, oR = O.upwind_first_weno5(og, stack[0], 1)
    close(sC[1][0], 0.5 * (oL + oR))
    with pytest.raises(ValueError):
        L.computeGradients(g, data[0])


@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_term_convection_vs_oracle_and_integration(scheme):
    """termConvection (term_convection.py:7): -V.grad(phi) upwinded by the sign of each component; scalar and
    array components, exact zeros, NumPy and device input; integrates through odeCFL3.  The shipped
    reference raises, so the oracle restates the documented formula (parity unpinned for this term)."""
    g, og = dubins((15, 14, 12))
    rng = np.random.default_rng(13)
    data = O.shape_cylinder(og, 2, None, .5) + 0.05 * rng.standard_normal(g.shape)
    v2 = np.sin(2 * og.xs[0]) * np.cos(og.xs[2]) + 0.3
    v2[np.abs(v2) < 0.05] = 0.0
    vel = [0.7, -0.4 * np.ones(g.shape), v2]
    y = data.reshape(-1, 1)
    sd = L.Bundle(dict(grid=g, velocity=vel, derivFunc=DERIV[scheme]))
    yd, sb, _ = L.termConvection(0., y, sd)
    yo, sbo = O.term_convection(og, vel, scheme, 0., y)
    assert yd.shape == y.shape and isinstance(sb, float)
    close(yd, yo)
    assert abs(sb - sbo) <= 1e-14 * sbo
    # device tensor in -> device tensor out; a callable velocity
    sd2 = L.Bundle(dict(grid=g, velocity=lambda t, d, s: vel, derivFunc=DERIV[scheme]))
    yd2, sb2, _ = L.termConvection(0., torch.as_tensor(y, device="cuda"), sd2)
    assert torch.is_tensor(yd2) and yd2.is_cuda and sb2 == sb
    close(yd2.cpu().numpy(), yo)
    # a few RK3 steps through the generic integrator loop
    term = lambda tt, yy: O.term_convection(og, vel, scheme, tt, yy)  # noqa: E731
    to, y3o = O.ode_cfl_3(term, [0., 0.05], y, 0.5)
    t3, y3, _ = L.odeCFL3(L.termConvection, [0., 0.05], y, L.odeCFLset(L.Bundle(dict(factorCFL=.5, singleStep='off'))), sd)
    assert abs(t3 - to) <= 1e-14
    (close if scheme.startswith("WENO") else close_eno)(y3, y3o, 1e-11)
    with pytest.raises(ValueError):
        L.termConvection(0., y, L.Bundle(dict(grid=g, velocity=[1., 2.], derivFunc=DERIV[scheme])))


# ------------------------------------------------------------------------------ randomized shapes / boundary mixes
def _fuzz_cases():
    rng = np.random.default_rng(2024)
    cases = []
    for k in range(28):
        dim = int(rng.choice([2, 3, 3, 3, 4]))
        hi = {2: 400, 3: 70, 4: 20}[dim]
        n = tuple(int(v) for v in rng.integers(3, hi, size=dim))
        if dim == 2 and k % 2:
            n = (int(rng.integers(3, 40)), int(rng.integers(200, 2500)))
        pd = tuple(int(d) for d in range(dim) if rng.random() < 0.4)
        tz = tuple(bool(rng.random() < 0.3) for _ in range(dim))
        scheme = SCHEMES[int(rng.integers(0, 4))]
        cases.append((k, dim, n, pd, tz, scheme))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "%d-%dd-%s-%s" % (c[0], c[1], "x".join(map(str, c[2])), c[5]))
def test_fuzz_shapes_boundaries_schemes(case, monkeypatch):
    """Random extents (down to the stencil width, primes, long thin grids), random periodic / extrapolated
    / towardZero mixes, every scheme and Hamiltonian: tiled kernel = direct kernel = oracle."""
    k, dim, n, pd, tz, scheme = case
    rng = np.random.default_rng(1000 + k)
    gmin = [-1.0 - 0.1 * d for d in range(dim)]
    gmax = [1.0 + 0.2 * d for d in range(dim)]
    g, _ = mk(gmin, gmax, n, list(pd) if pd else None)
    for d in range(dim):
        if tz[d] and d not in pd:
            g.bdryData[d] = L.Bundle(dict(towardZero=True))
    og = O.Grid(gmin, gmax, n, list(pd), toward_zero=[tz[d] and d not in pd for d in range(dim)])
    data = O.shape_sphere(og, None, .6) + 0.05 * rng.standard_normal(n)
    y = data.reshape(-1, 1)
    if dim == 2:
        sys_, osys = L.DoubleIntegrator(g, 1.3), O.DoubleIntegrator(og, 1.3)
    elif dim == 3:
        sys_, osys = L.DubinsVehicleRel(g, 1.5, 0.7), O.DubinsRel(og, 1.5, 0.7)
    else:
        sys_, osys = L.DoublePendulum4D(g, 1.0), O.DoublePendulum4D(og, 1.0)
    (yt, sbt, _), (yd, sbd, _) = _term_both_kernels(g, sys_, scheme, y, monkeypatch)
    yo, sbo = O.term_lax_friedrichs(og, osys, scheme, 0., y)
    tol = 1e-10 if dim == 4 else 1e-11
    close(yt, yo, tol, what="tiled")
    close(yd, yo, tol, what="direct")
    assert abs(sbt - sbo) <= 1e-12 * sbo and abs(sbd - sbo) <= 1e-12 * sbo


@pytest.mark.parametrize("comp,with_target,with_obstacle", [
    ("minVOverTime", False, False), ("maxVOverTime", False, True), ("minVWithV0", False, False),
    ("maxVWithV0", False, True), ("minVWithL", True, False), ("maxVWithL", True, True), ("set", False, True),
    ("minWithZero", False, False)])
def test_hjipde_solve_interval_at_once_equals_stepwise(comp, with_target, with_obstacle, monkeypatch):
    """The one-call-per-interval path (post-step operators fused into the last RK stage) against the
    step-by-step loop with separate min/max kernels: bitwise the same tube, time-varying obstacle included."""
    g, og, sd, data0 = _solve_setup((19, 17, 15))
    tau = np.linspace(0, 0.12, 5)
    ex = dict(quiet=True)
    if with_target:
        ex["targetFunction"] = data0 + 0.1 * np.cos(2 * og.xs[1])
    if with_obstacle:
        obs = np.sqrt((og.xs[0] - 1.5) ** 2 + og.xs[1] ** 2) - 0.4
        ex["obstacleFunction"] = np.stack([obs + 0.02 * k for k in range(len(tau))])
    outs = []
    for stepwise in ("1", "0"):
        monkeypatch.setenv("HJ_HJIPDE_STEPWISE", stepwise)
        sdk = L.Bundle(dict(grid=g, hamFunc=sd.hamFunc, partialFunc=sd.partialFunc, dissFunc=sd.dissFunc,
                            CoStateCalc=sd.CoStateCalc))
        d, t, _ = L.HJIPDE_solve(data0, tau, sdk, comp, L.Bundle(dict(ex)))
        outs.append(d)
    assert outs[0].shape == outs[1].shape == (len(tau),) + tuple(g.shape)
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0][-1], outs[0][0])


def test_zz_report_eno_tie_statistics():
    """Not a check: writes what the masked ENO comparisons saw (fraction of cells beyond the strict
    tolerance, max error) to gpurun_out/eno_tie_stats.txt so the bound quoted in DESIGN.md is measured."""
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "eno_tie_stats.txt"), "w") as f:
        for what, beyond, err in ENO_STATS:
            f.write("%-40s beyond_strict=%.3e max_err/scale=%.3e\n" % (what or "-", beyond, err))
