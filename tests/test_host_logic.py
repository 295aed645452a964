"""Host-side helpers of HJIPDE_solve that need no GPU: point evaluation (eval_u) and grid trimming."""
import numpy as np
import pytest
from scipy.interpolate import RegularGridInterpolator

import levelsetpy_amd as L
from levelsetpy_amd.hji_solver import _eval_point, _trim


def _grid():
    n = np.array([[9], [7], [8]], dtype=np.int64)
    gmin = np.array([[-1.], [-2.], [-np.pi]])
    gmax = np.array([[1.], [2.], [np.pi * (1 - 2 / 8)]])
    return L.createGrid(gmin, gmax, n, 2)


def test_eval_point_matches_scipy_with_periodic_augmentation():
    """ValueFuncs/evaluate_u.py:64-116: periodic axes get one wrapped node appended and the state is
    wrapped into the period; multilinear interpolation otherwise."""
    g = _grid()
    rng = np.random.default_rng(0)
    data = rng.standard_normal(g.shape)
    vs = [np.asarray(v).ravel() for v in g.vs]
    dx2 = float(np.asarray(g.dx).ravel()[2])
    aug = np.concatenate([data, data[:, :, :1]], axis=2)
    f = RegularGridInterpolator((vs[0], vs[1], np.append(vs[2], vs[2][-1] + dx2)), aug)
    period = 8 * dx2
    for _ in range(50):
        x = np.array([rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-3 * np.pi, 3 * np.pi)])
        xw = x.copy()
        xw[2] = vs[2][0] + ((x[2] - vs[2][0]) % period)
        assert abs(_eval_point(g, data, x) - float(f(xw)[0])) <= 1e-12
    # grid nodes are reproduced exactly, the far corner included
    assert _eval_point(g, data, [vs[0][3], vs[1][2], vs[2][5]]) == pytest.approx(data[3, 2, 5], abs=1e-14)
    assert _eval_point(g, data, [vs[0][-1], vs[1][-1], vs[2][-1]]) == pytest.approx(data[-1, -1, -1], abs=1e-14)
    # outside an extrapolated axis: NaN (MATLAB interpn semantics the caller tests for)
    assert np.isnan(_eval_point(g, data, [1.5, 0., 0.]))
    with pytest.raises(ValueError):
        _eval_point(g, data, [0., 0.])


def test_trim_keeps_nodes_strictly_inside_four_cells():
    g = _grid()          # 9 x 7 x 8 nodes: only axis 0 has nodes strictly between min+4dx and max-4dx... none
    a = np.arange(np.prod(g.shape), dtype=np.float64).reshape(g.shape)
    assert _trim(g, a).size == 0
    n = np.array([[15], [12], [11]], dtype=np.int64)
    g2 = L.createGrid(np.array([[-1.], [-2.], [0.]]), np.array([[1.], [2.], [1.]]), n, None)
    b = np.arange(np.prod(g2.shape), dtype=np.float64).reshape(g2.shape)
    t = _trim(g2, b)
    # node 4 sits at min + 4 dx up to rounding (the reference's comparison has the same edge): 4 or 5 nodes go
    for d, nd in enumerate((15, 12, 11)):
        assert nd - 10 <= t.shape[d] <= nd - 8
    vs = [np.asarray(v).ravel() for v in g2.vs]
    dx = np.asarray(g2.dx).ravel()
    keep = [np.nonzero((vs[d] > vs[d][0] + 4 * dx[d]) & (vs[d] < vs[d][-1] - 4 * dx[d]))[0] for d in range(3)]
    assert np.array_equal(t, b[np.ix_(*keep)])


def test_float_reciprocal_index_division_is_exact_below_2_pow_22():
    """hj_device.h fdivmod: q = (int)((float)a * rcp(d)) followed by one correction in each direction.  NumPy float32
    replica over every dividend below 2^22 for divisors of the sizes the kernels use (tile extents, tile counts, halo
    areas), with the hardware reciprocal's 1-ulp error taken in both directions."""
    a = np.arange(0, 1 << 22, dtype=np.int64)
    af = a.astype(np.float32)
    for d in (1, 3, 7, 15, 27, 30, 68, 84, 133, 513, 4096, 65521):
        r0 = np.float32(1.0) / np.float32(d)
        for r in (r0, np.nextafter(r0, np.float32(0)), np.nextafter(r0, np.float32(2))):
            q = (af * r).astype(np.int64)             # float32 product, truncated like v_cvt_i32_f32
            rem = a - q * d
            lo = rem < 0
            q = q - lo; rem = rem + lo * d
            hi = rem >= d
            q = q + hi; rem = rem - hi * d
            assert np.array_equal(q, a // d) and np.array_equal(rem, a % d), (d, float(r))
