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


# ---------------------------------------------------------------- HostView (lazy.py): ndarray-compatible handle
def _hv(a):
    import torch
    from levelsetpy_amd.lazy import HostView
    return HostView(torch.from_numpy(np.array(a, dtype=np.float64)))


def test_hostview_looks_like_the_ndarray_it_stands_for():
    from levelsetpy_amd.lazy import HostView
    a = np.arange(24, dtype=np.float64).reshape(6, 4) - 7.5
    v = _hv(a)
    assert v.shape == (6, 4) and v.ndim == 2 and v.size == 24 and v.dtype == np.float64 and len(v) == 6 and v.nbytes == 192
    assert v._h is None                                   # nothing copied by asking for metadata
    assert np.array_equal(np.asarray(v), a) and np.asarray(v) is np.asarray(v)     # exact, cached
    assert isinstance(v + 1, np.ndarray) and np.array_equal(v + 1, a + 1) and np.array_equal(2 * v - v, a)
    assert np.array_equal(np.abs(v), np.abs(a)) and np.array_equal(np.minimum(v, 0), np.minimum(a, 0))
    assert v.sum() == a.sum() and v.min() == a.min() and np.linalg.norm(v) == np.linalg.norm(a)
    assert np.array_equal(v[2:4, 1], a[2:4, 1]) and v[0, 0] == a[0, 0] and np.array_equal(v.T, a.T)
    assert np.array_equal(np.concatenate([v, v]), np.concatenate([a, a]))
    assert [float(x[0]) for x in v] == [float(x[0]) for x in a]
    r = v.reshape(-1, 1)
    assert isinstance(r, HostView) and r.shape == (24, 1) and np.array_equal(r, a.reshape(-1, 1))
    assert isinstance(v.flatten(), HostView) and v.flatten().shape == (24,) and v.reshape(4, 6).reshape((2, 12)).shape == (2, 12)
    assert isinstance(r.squeeze(), HostView) and r.squeeze().shape == (24,)
    assert np.array_equal(v.reshape(4, 6, order="F"), a.reshape(4, 6, order="F"))      # non-C orders: through the host
    assert float(_hv([2.5]).reshape(())) == 2.5
    import copy as _copy
    import pickle
    assert np.array_equal(_copy.copy(v), a) and np.array_equal(_copy.deepcopy(v), a)
    assert np.array_equal(pickle.loads(pickle.dumps(v)), a)
    assert "HostView(device" in repr(v)


def test_hostview_writes_detach_and_never_touch_the_tensor():
    a = np.linspace(-1, 1, 12).reshape(3, 4)
    v = _hv(a)
    t = v.device_tensor()
    host = np.asarray(v)
    assert not host.flags.writeable                      # the device copy is what the next call consumes
    with pytest.raises(ValueError):
        host[0, 0] = 9.0
    w = v.reshape(12)
    v[0, 1] = 5.0                                        # write through the view: private host copy, device let go
    assert v.device_tensor() is None and v[0, 1] == 5.0 and np.asarray(v).flags.writeable
    assert w.device_tensor() is not None and w.device_tensor().data_ptr() == t.data_ptr() and w[1] == a[0, 1]    # other views unaffected
    assert float(t[0, 1]) == a[0, 1]                     # the tensor itself was never written
    out = _hv(np.zeros(12))
    np.add(w, 1.0, out=out)
    assert out.device_tensor() is None and np.array_equal(out, a.reshape(12) + 1)


def test_hostview_is_what_numpy_callers_get_and_can_be_switched_off(monkeypatch):
    import torch
    from levelsetpy_amd import lazy
    from levelsetpy_amd.context import DeviceGrid, array_dtype_name

    class FakeDG(object):       # DeviceGrid.like without a GPU: only the branch taken for a host tensor matters here
        pass
    FakeDG.torch = torch
    t = torch.arange(6, dtype=torch.float64)
    got = DeviceGrid.like(FakeDG(), t, np.zeros(6), (6, 1), lazy=True)
    assert isinstance(got, np.ndarray) and got.shape == (6, 1)      # a CPU tensor is simply converted (no device to stay on)
    assert array_dtype_name(lazy.HostView(torch.zeros(3, dtype=torch.float32))) == "float32"
    assert array_dtype_name(lazy.HostView(torch.zeros(3, dtype=torch.float64))) == "float64"
    lazy.set_lazy(False)
    assert lazy.LAZY is False
    lazy.set_lazy(True)


def test_hostview_metadata_functions_do_not_copy():
    v = _hv(np.zeros((5, 1)))
    assert np.ndim(v) == 2 and np.shape(v) == (5, 1) and np.size(v) == 5 and v._h is None
    from levelsetpy_amd import integration
    import levelsetpy_amd as L
    integration._check_shape(L.termLaxFriedrichs, v)
    with pytest.raises(ValueError):
        integration._check_shape(L.termRestrictUpdate, v)
    assert v._h is None                                   # the integrators' own checks never look at the values


def test_devicearray_is_a_real_ndarray_that_remembers_its_tensor(tmp_path):
    """HJ_LAZY_NUMPY=ndarray (lazy.DeviceArray): isinstance / np.save / pickle / buffer protocol as with the reference's plain arrays
    (ode_cfl_3.py:241-272), the device tensor consumed when the array is passed back, writes detach."""
    import pickle
    import torch
    from levelsetpy_amd import lazy
    from levelsetpy_amd.context import DeviceGrid, array_dtype_name
    a = np.arange(24, dtype=np.float64).reshape(6, 4) - 7.5
    t = torch.from_numpy(a.copy())
    v = lazy.device_array(t)
    assert isinstance(v, np.ndarray) and type(v) is lazy.DeviceArray and v.shape == (6, 4) and v.dtype == np.float64
    assert v.device_tensor() is not None and v.device_tensor().data_ptr() == t.data_ptr()
    assert np.array_equal(np.asarray(v), a) and bytes(memoryview(v)) == a.tobytes()       # the values ARE there (no hook needed)
    assert type(v + 1) is np.ndarray and np.array_equal(np.abs(v), np.abs(a)) and v.sum() == a.sum()
    f = tmp_path / "y.npy"
    np.save(f, v)
    assert np.array_equal(np.load(f), a) and type(pickle.loads(pickle.dumps(v))) is np.ndarray
    # read-only while attached: a write through a plain view cannot silently diverge from the tensor
    assert not v.flags.writeable
    with pytest.raises(ValueError):
        np.asarray(v)[0, 0] = 1.0
    # views and copies are detached; C-order reshapes of the object itself keep the tensor
    assert v[1:3].device_tensor() is None and v.copy().device_tensor() is None and v.T.device_tensor() is None
    r = v.reshape(-1, 1)
    assert type(r) is lazy.DeviceArray and r.device_tensor() is not None and tuple(r.device_tensor().shape) == (24, 1)
    assert v.ravel().device_tensor() is not None and r.squeeze().device_tensor() is not None
    assert v.reshape(4, 6, order="F").device_tensor() is None
    # consumed on the device when passed back: to_device hands out the tensor, not an upload

    class FakeDG(object):
        pass
    FakeDG.torch, FakeDG.device, FakeDG.tdtype = torch, t.device, torch.float64
    assert DeviceGrid.to_device(FakeDG(), v).data_ptr() == t.data_ptr()
    assert array_dtype_name(lazy.device_array(torch.zeros(3, dtype=torch.float32))) == "float32"
    # writing through the array itself detaches it and makes it writable; the tensor is never written
    v[0, 1] = 5.0
    assert v.device_tensor() is None and v.flags.writeable and v[0, 1] == 5.0
    out = lazy.device_array(torch.zeros(24, dtype=torch.float64))
    np.add(np.arange(24.0), 1.0, out=out)
    assert out.device_tensor() is None and np.array_equal(out, np.arange(24.0) + 1)
    # aliases share ONE attachment: a write through a reshape / ravel / squeeze alias detaches the original too, and the
    # other way round -- both look at the same host memory, neither may go on standing for a tensor that no longer matches it
    y = lazy.device_array(torch.zeros(4, 3, dtype=torch.float64))
    z = y.reshape(-1)
    z[0] = 99.0
    assert z.device_tensor() is None and y.device_tensor() is None and y[0, 0] == 99.0
    y[1, 1] = 5.0                                   # ... and the original is writable as well now
    assert z[4] == 5.0
    y = lazy.device_array(torch.zeros(4, 3, dtype=torch.float64))
    z, q = y.ravel(), y.reshape(3, 4).squeeze()
    y[0, 0] = 7.0
    assert z.device_tensor() is None and q.device_tensor() is None and z[0] == 7.0 and q[0, 0] == 7.0
    y = lazy.device_array(torch.zeros(4, 3, dtype=torch.float64))
    z = y.reshape(12)
    np.add(z, 1.0, out=z)
    assert y.device_tensor() is None and y[3, 2] == 1.0
    # the switch
    old = lazy.LAZY
    try:
        lazy.set_lazy("ndarray")
        assert lazy.LAZY == "ndarray"
        lazy.set_lazy(False)
        assert lazy.LAZY is False
        lazy.set_lazy("1")
        assert lazy.LAZY is True
    finally:
        lazy.set_lazy(old)
