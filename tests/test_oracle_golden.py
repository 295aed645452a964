"""Pins the CPU oracle (oracle/hj_oracle.py) to golden vectors generated from the
unmodified reference (tests/golden/make_golden.py).  CPU only.

Tolerances: derivatives / ydot  abs <= 1e-12*max(1,|ref|_inf);
stepBound and t  rel 1e-14; ghost cells bit-exact.
"""
import json
import os

import numpy as np
import pytest

from oracle import hj_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def close(a, ref, tol=1e-12):
    ref = np.asarray(ref)
    a = np.asarray(a)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    scale = max(1.0, float(np.max(np.abs(ref))))
    err = float(np.max(np.abs(a - ref)))
    assert err <= tol * scale, "err %.3e > %.1e*%.3g" % (err, tol, scale)


class _G(object):
    pass


def mkgrid(gmin, gmax, N, bc):
    pd = [i for i, b in enumerate(bc) if b]
    return O.Grid(gmin, gmax, N, pd)


# ---------------------------------------------------------------- ghosts: bit-exact
@pytest.mark.parametrize("dim", [0, 1, 2])
@pytest.mark.parametrize("w", [1, 2, 3])
def test_ghost_bit_exact(golden, dim, w):
    G = golden("ghost.npz")
    x = G["x"]
    assert np.array_equal(O.add_ghost_periodic(x, dim, w), G["per_d%d_w%d" % (dim, w)])
    assert np.array_equal(O.add_ghost_extrapolate(x, dim, w, False), G["ext_d%d_w%d_tz0" % (dim, w)])
    assert np.array_equal(O.add_ghost_extrapolate(x, dim, w, True), G["ext_d%d_w%d_tz1" % (dim, w)])


def test_ghost_f32_input_returns_f64(golden):
    G = golden("ghost.npz")
    out = O.add_ghost_extrapolate(G["x"].astype(np.float32), 1, 2, False)
    assert out.dtype == np.float64
    assert np.array_equal(out, G["ext_f32in_d1_w2"])


def test_ghost_width_errors():
    x = np.zeros((4, 5))
    with pytest.raises(ValueError):
        O.add_ghost_extrapolate(x, 0, 5)
    assert O.add_ghost_periodic(x, 1, None).shape == (4, 7)


# ---------------------------------------------------------------- derivatives
CASES = [("g2", 2), ("g3", 3), ("g3s", 3), ("g4", 4)]


def _grid_for(G, tag):
    base = "g3" if tag == "g3s" else tag
    data = G[tag + "_data"]
    g = mkgrid(G[base + "_min"], G[base + "_max"], data.shape, G[tag + "_bc"])
    np.testing.assert_array_equal(g.dx.ravel(), G[tag + "_dx"])
    return g, data


@pytest.mark.parametrize("tag,nd", CASES)
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_upwind_vs_reference(golden, tag, nd, scheme):
    G = golden("deriv.npz")
    g, data = _grid_for(G, tag)
    for dim in range(nd):
        L, R = O.SCHEMES[scheme](g, data, dim)
        close(L, G["%s_%s_L_d%d" % (tag, scheme, dim)])
        close(R, G["%s_%s_R_d%d" % (tag, scheme, dim)])


@pytest.mark.parametrize("tag,nd", [("g2", 2), ("g3", 3)])
def test_eno3_helper_vs_reference(golden, tag, nd):
    G = golden("deriv.npz")
    g, data = _grid_for(G, tag)
    for dim in range(nd):
        dL, dR, DD = O.eno3_helper(g, data, dim)
        for k in range(3):
            close(dL[k], G["%s_helper_dL%d_d%d" % (tag, k, dim)])
            close(dR[k], G["%s_helper_dR%d_d%d" % (tag, k, dim)])
        for name in ("D1", "D2", "D3"):
            close(DD[name], G["%s_helper_%s_d%d" % (tag, name, dim)])


def test_asshipped_weno5_is_linear_weights(golden):
    """as shipped the weights collapse to (.1,.6,.3)."""
    G = golden("deriv.npz")
    g, data = _grid_for(G, "g3")
    for dim in range(3):
        dL, dR, _ = O.eno3_helper(g, data, dim)
        close(.1 * dL[0] + .6 * dL[1] + .3 * dL[2], G["g3_WENO5_ASSHIPPED_L_d%d" % dim], 1e-14)
        close(.3 * dR[0] + .6 * dR[1] + .1 * dR[2], G["g3_WENO5_ASSHIPPED_R_d%d" % dim], 1e-14)


def test_true_weno5_equals_asshipped_on_linear_data_and_converges():
    # on data linear along the axis every smoothness estimate is 0 -> optimal weights
    g = O.Grid([0, 0], [1, 1], [24, 20], None)
    data = 2.0 * g.xs[0] - 3.0 * g.xs[1]
    for dim in range(2):
        La, Ra = O.upwind_first_weno5(g, data, dim, 'asshipped')
        Lw, Rw = O.upwind_first_weno5(g, data, dim, 'weno5')
        # away from the (sign-aware, hence kinked) extrapolated ghosts
        close(Lw[3:-3, 3:-3], La[3:-3, 3:-3], 1e-12)
        close(Rw[3:-3, 3:-3], Ra[3:-3, 3:-3], 1e-12)
    # 5th-order convergence on smooth periodic data
    errs = []
    for n in (32, 64):
        g = O.Grid([0.0], [2 * np.pi * (1 - 1 / n)], [n], [0])
        g.xs = [g.vs[0].ravel()]
        g.shape = (n,)
        x = g.xs[0]
        L, R = O.upwind_first_weno5(g, np.sin(x), 0, 'weno5')
        errs.append(max(np.max(np.abs(L - np.cos(x))), np.max(np.abs(R - np.cos(x)))))
    order = np.log2(errs[0] / errs[1])
    assert order > 4.5, (errs, order)


# ---------------------------------------------------------------- GLF + LF term
def test_glf_vs_reference(golden):
    G = golden("term.npz")
    g = mkgrid(G["dub_min"], G["dub_max"], G["dub_N"], [0, 0, 1])
    sys_ = O.DubinsRel(g, 1, 1)
    dL = [G["glf_dL%d" % i] for i in range(3)]
    dR = [G["glf_dR%d" % i] for i in range(3)]
    diss, sb = O.artificial_dissipation_glf(g, sys_, 0., G["dub_data"], dL, dR)
    close(diss, G["glf_diss"])
    assert abs(sb - float(G["glf_sb"])) <= 1e-14 * abs(sb)


@pytest.mark.parametrize("ub,wb", [(1, 1), (5, 5), (2, 3)])
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_term_dubins_vs_reference(golden, ub, wb, scheme):
    G = golden("term.npz")
    g = mkgrid(G["dub_min"], G["dub_max"], G["dub_N"], [0, 0, 1])
    sys_ = O.DubinsRel(g, ub, wb)
    y = G["dub_data"].reshape(-1, 1)
    yd, sb = O.term_lax_friedrichs(g, sys_, scheme, 0.3, y)
    close(yd, G["dub_u%d_w%d_%s_ydot" % (ub, wb, scheme)])
    ref = float(G["dub_u%d_w%d_%s_sb" % (ub, wb, scheme)])
    assert abs(sb - ref) <= 1e-14 * abs(ref)


@pytest.mark.parametrize("ub", [1, 2.5])
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_term_double_integrator_vs_reference(golden, ub, scheme):
    G = golden("term.npz")
    g = mkgrid(G["di_min"], G["di_max"], G["di_N"], [0, 0])
    sys_ = O.DoubleIntegrator(g, ub)
    yd, sb = O.term_lax_friedrichs(g, sys_, scheme, 0., G["di_data"].reshape(-1, 1))
    close(yd, G["di_u%s_%s_ydot" % (ub, scheme)])
    ref = float(G["di_u%s_%s_sb" % (ub, scheme)])
    assert abs(sb - ref) <= 1e-14 * abs(ref)


# ---------------------------------------------------------------- integrators
@pytest.mark.parametrize("scheme", ["ENO2", "ENO3", "WENO5_ASSHIPPED"])
def test_ode_cfl_vs_reference(golden, scheme):
    G = golden("ode.npz")
    g = mkgrid(G["dub_min"], G["dub_max"], G["dub_N"], [0, 0, 1])
    sys_ = O.DubinsRel(g, 1, 1)
    term = lambda t, y: O.term_lax_friedrichs(g, sys_, scheme, t, y)  # noqa: E731
    y = G["dub_data"].reshape(-1, 1)
    t = 0.
    for k in range(5):
        t, y = O.ode_cfl_3(term, [t, 10.], y, 0.8, single_step=True)
        if k in (0, 4):
            ref_t = float(G["rk3_%s_t%d" % (scheme, k + 1)])
            assert abs(t - ref_t) <= 1e-14 * ref_t
            close(y, G["rk3_%s_y%d" % (scheme, k + 1)], 1e-12)
    # noisy initial data (no exact ENO ties)
    y = G["dubn_data"].reshape(-1, 1)
    t = 0.
    for k in range(5):
        t, y = O.ode_cfl_3(term, [t, 10.], y, 0.8, single_step=True)
    assert abs(t - float(G["rk3n_%s_t5" % scheme])) <= 1e-14
    close(y, G["rk3n_%s_y5" % scheme], 1e-12)
    t, y = O.ode_cfl_2(term, [0., 0.02], G["dub_data"].reshape(-1, 1), 0.95)
    assert abs(t - float(G["rk2_%s_t" % scheme])) <= 1e-14
    close(y, G["rk2_%s_y" % scheme], 1e-12)
    # termRestrictUpdate(positive=0) with a 1-D state, as the air3D notebooks call it
    rterm = O.term_restrict_update(term, positive=False)
    t, y = O.ode_cfl_2(rterm, [0., 0.02], G["dub_data"].flatten(), 0.95)
    assert y.shape == G["rk2r_%s_y" % scheme].shape
    assert abs(t - float(G["rk2r_%s_t" % scheme])) <= 1e-14
    close(y, G["rk2r_%s_y" % scheme], 1e-12)


def test_ode_cfl3_double_integrator_vs_reference(golden):
    G = golden("ode.npz")
    g = O.Grid([-1, -1], [1, 1], [32, 32], None)
    sys_ = O.DoubleIntegrator(g, 1)
    term = lambda t, y: O.term_lax_friedrichs(g, sys_, "ENO3", t, y)  # noqa: E731
    y = G["di_data"].reshape(-1, 1)
    t = 0.
    for _ in range(5):
        t, y = O.ode_cfl_3(term, [t, 10.], y, 0.8, single_step=True)
    assert abs(t - float(G["di_rk3_ENO3_t5"])) <= 1e-14
    close(y, G["di_rk3_ENO3_y5"], 1e-12)
    y = G["din_data"].reshape(-1, 1)
    t = 0.
    for _ in range(5):
        t, y = O.ode_cfl_3(term, [t, 10.], y, 0.8, single_step=True)
    close(y, G["din_rk3_ENO3_y5"], 1e-12)


# ---------------------------------------------------------------- known answers
def test_known_answers_51cubed():
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        KA = json.load(f)
    n = 51
    g = O.Grid([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / n)], [n] * 3, [2])
    d0 = O.shape_cylinder(g, 2, None, .5)
    sys_ = O.DubinsRel(g, 1, 1)
    for scheme in ("WENO5_ASSHIPPED",):
        ka = KA["dubins51_" + scheme]
        term = lambda t, y: O.term_lax_friedrichs(g, sys_, scheme, t, y)  # noqa: E731
        y = d0.reshape(-1, 1)
        yd, sb = term(0., y)
        assert abs(sb - ka["stepBound0"]) <= 1e-14 * sb
        assert abs(np.linalg.norm(yd) - ka["ydot0_l2"]) <= 1e-11 * ka["ydot0_l2"]
        t = 0.
        for k in range(5):
            t, y = O.ode_cfl_3(term, [t, 10.], y, 0.8, single_step=True)
            if k == 0:
                assert abs(t - ka["t1"]) <= 1e-14
        assert abs(t - ka["t5"]) <= 1e-14
        assert abs(y.sum() - ka["sum5"]) <= 1e-11 * abs(ka["sum5"])
        assert abs(np.linalg.norm(y) - ka["l2_5"]) <= 1e-12 * ka["l2_5"]
        assert abs(y.min() - ka["min5"]) <= 1e-12 and abs(y.max() - ka["max5"]) <= 1e-12


class _RangeProbe(object):
    """The partialFunc of tests/golden/make_golden.py:gen_extra: a scalar alpha built from the global range of
    the next dimension and the extrema of this dimension's per-node range."""

    @staticmethod
    def dissipation(t, data, derivMin, derivMax, sd, dim):
        j = (dim + 1) % 3
        glob = max(abs(float(np.asarray(derivMin[j]))), abs(float(np.asarray(derivMax[j]))))
        loc = max(abs(float(np.asarray(derivMin[dim]).min())), abs(float(np.asarray(derivMax[dim]).max())))
        return 0.25 * (dim + 1) + 0.5 * glob + 0.125 * loc


def test_llf_scalar_alpha_vs_reference(golden):
    """artificialDissipationLLF pinned for the one case the shipped function runs (0-d alpha for every
    dimension, diss_local_laxfried.py:116-121): diss, stepBound, and WHICH ranges partialFunc receives
    (per-node arrays in its own dimension, global scalars in the others)."""
    G = golden("extra.npz")
    g = mkgrid(G["g3_min"], G["g3_max"], G["g3_data"].shape, [0, 0, 1])
    dL = [G["llf_dL%d" % i] for i in range(3)]
    dR = [G["llf_dR%d" % i] for i in range(3)]
    diss, sb = O.artificial_dissipation_local(g, _RangeProbe, 0., G["g3_data"], dL, dR, False)
    close(diss, G["llf_diss"])
    assert abs(sb - float(G["llf_sb"])) <= 1e-14 * sb
    assert G["llf_range_ndims"].tolist() == [[3, 0, 0], [0, 3, 0], [0, 0, 3]]


def test_eno3_helper_fourth_candidate_vs_reference(golden):
    G = golden("extra.npz")
    g = mkgrid(G["g3_min"], G["g3_max"], G["g3_data"].shape, [0, 0, 1])
    for dim in range(3):
        dL, dR, _ = O.eno3_helper(g, G["g3_data"], dim, approx4=True)
        close(dL[3], G["g3_helper4_dL3_d%d" % dim])
        close(dR[3], G["g3_helper4_dR3_d%d" % dim])


def test_oracle_eval_u_is_the_reference_interpolant():
    """oracle.eval_u restates ValueFuncs/evaluate_u.py:64-120: scipy's RegularGridInterpolator ('linear') on the node
    vectors, periodic axes augmented by one wrapped node and the state shifted by whole periods.  Checked against
    scipy itself (the reference's own call, :104-113) on random states, inside and across the periodic seam."""
    from scipy.interpolate import RegularGridInterpolator
    og = O.Grid([-.75, -1.25, -np.pi], [3.25, 1.25, np.pi * (1 - 2 / 12)], [9, 11, 12], [2])
    rng = np.random.default_rng(3)
    data = rng.standard_normal(og.shape)
    vs = [v.ravel() for v in og.vs]
    vs[2] = np.concatenate([vs[2], [vs[2][-1] + og.dx[2, 0]]])
    aug = np.concatenate([data, data[:, :, :1]], axis=2)
    ref = RegularGridInterpolator(vs, aug)
    for _ in range(50):
        x = np.array([rng.uniform(-.75, 3.25), rng.uniform(-1.25, 1.25), rng.uniform(-3 * np.pi, 3 * np.pi)])
        xw = x.copy()
        period = vs[2][-1] - vs[2][0]
        while xw[2] > vs[2][-1]:
            xw[2] -= period
        while xw[2] < vs[2][0]:
            xw[2] += period
        assert abs(O.eval_u(og, data, x) - float(ref(xw)[0])) <= 1e-13
    assert np.isnan(O.eval_u(og, data, [4.0, 0.0, 0.0]))
