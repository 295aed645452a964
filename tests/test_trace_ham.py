"""CPU: the tracer that turns Python hamFunc / partialFunc callbacks into a device expression (levelsetpy_amd/trace_ham.py).

No GPU here: the traced graph is evaluated in NumPy (Traced.evaluate) against the callbacks themselves on real arrays, the
generated source is compiled with hipRTC (hj_ham_compile_check needs no device), and everything the tracer must refuse is
checked to be refused -- a refused pair keeps the split path, a wrongly accepted one would compute something else."""
import math

import numpy as np
import pytest

import levelsetpy_amd as L
from levelsetpy_amd import trace_ham as TH


def grid3(n=(12, 10, 14), low_mem=False):
    return L.createGrid(np.array([[-2., -2., -np.pi]]).T, np.array([[2., 2., np.pi * (1 - 2 / n[2])]]).T, np.array(n, dtype=np.int64).reshape(-1, 1), 2,
                        low_mem=low_mem)


class DubinsAbs(object):
    """The reference's style: grid.xs read inside the callbacks (DynamicalSystems/dubins_absolute.py:150-170)."""

    def __init__(self, grid, v, w):
        self.grid, self.v, self.w = grid, v, w

    def hamiltonian(self, t, data, p, sd=None):
        x3 = self.grid.xs[2]
        return self.v * (p[0] * np.cos(x3) + p[1] * np.sin(x3)) + self.w * np.abs(p[2])

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        x3 = self.grid.xs[2]
        if dim == 0:
            return np.abs(self.v * np.cos(x3))
        if dim == 1:
            return np.abs(self.v * np.sin(x3))
        return self.w


class Nonlinear(object):
    """Every kind of node the tracer writes: comparisons, where, maximum / minimum, powers, sign, clip, booleans as numbers, the costate range."""

    def __init__(self, grid, c):
        self.grid, self.c, self.k = grid, c, 3

    def hamiltonian(self, t, data, p, sd=None):
        x = self.grid.xs
        q = np.sqrt(p[0] ** 2 + p[1] ** 2 + 1e-12) + 0.5 * p[2] ** self.k
        gate = (x[0] > 0) * 1.0 + (x[1] <= 0.25)
        sw = np.where((p[0] > 0) & ~(p[1] > 0.5), np.maximum(p[0], 0.1 * x[1]), np.minimum(p[1], -p[2]))
        return -q + self.c * x[0] * p[1] * gate + sw * np.sign(p[2]) + np.clip(p[0], -0.3, 0.7) / (2 + np.cos(x[2])) + np.tanh(p[1]) * np.exp(-x[1] ** 2)

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        a = np.maximum(np.abs(dmin[dim]), np.abs(dmax[dim]))
        if dim == 1:
            return a + np.abs(self.c * self.grid.xs[0])
        if dim == 2:
            return a + 1.5 * np.maximum(np.abs(dmin[2]), np.abs(dmax[2])) ** 2
        return a + 0.5


def real_eval(sysobj, g, p, dmin, dmax):
    H = sysobj.hamiltonian(0., None, p, None)
    al = [np.broadcast_to(np.asarray(sysobj.dissipation(0., None, dmin, dmax, None, d), dtype=np.float64), g.shape) for d in range(g.dim)]
    return H, al


@pytest.mark.parametrize("low_mem", [False, True])
@pytest.mark.parametrize("cls,args", [(DubinsAbs, (1.3, 0.7)), (Nonlinear, (0.4,))])
def test_traced_graph_computes_what_the_callbacks_compute(cls, args, low_mem):
    g = grid3(low_mem=low_mem)
    obj = cls(g, *args)
    xs_before = [np.array(v, copy=True) for v in g.xs]
    tr = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    assert all(np.array_equal(a, b) for a, b in zip(xs_before, g.xs)) and all(isinstance(v, np.ndarray) for v in g.xs)   # the grid is as it was
    rng = np.random.default_rng(3)
    p = [rng.standard_normal(g.shape) for _ in range(3)]
    dmin, dmax = [-1.5, -0.75, -2.0], [1.25, 2.5, 0.5]
    H, al = tr.evaluate(np.meshgrid(*[np.asarray(v).ravel() for v in g.vs], indexing="ij"), p, dmin, dmax)
    Hr, alr = real_eval(obj, g, p, dmin, dmax)
    assert np.allclose(H, Hr, rtol=1e-14, atol=1e-14), float(np.abs(H - Hr).max())
    for d in range(3):
        assert np.allclose(np.broadcast_to(al[d], g.shape), alr[d], rtol=1e-14, atol=1e-14), d
    assert tr.uses_range == (cls is Nonlinear)
    # hoisted column values depend on the in-plane coordinates only
    if tr.column_source:
        assert "x[0]" not in tr.column_source and "p[" not in tr.column_source and tr.ncol >= 1


def test_parameters_become_par_and_a_changed_speed_keeps_the_source():
    g = grid3()
    obj = DubinsAbs(g, 1.3, 0.7)
    a = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    obj.v = 2.5
    b = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    assert a.params == [1.3, 0.7] and b.params == [2.5, 0.7]
    assert a.source == b.source and a.column_source == b.column_source
    assert "1.3" not in a.source and "par[0]" in (a.source + (a.column_source or ""))
    # a ninth distinct float is a literal (HamTables::par has eight slots): exact, as a hexadecimal floating literal
    class Many(DubinsAbs):
        def hamiltonian(self, t, data, p, sd=None):
            x = self.grid.xs
            return 0.1 * p[0] + 0.2 * p[1] + 0.3 * p[2] + 0.4 * x[0] + 0.6 * x[1] + 0.9 * x[2] + 1.1 * p[0] * p[1] + 1.3 * p[1] * p[2] + 0.7 * p[0] * p[2]
    m = Many(g, 1., 1.)
    c = TH.trace_callbacks(g, m.hamiltonian, m.dissipation, None)
    assert len(c.params) == 8 and float(0.7).hex() in c.source


def test_fingerprint_sees_in_place_changes():
    g = grid3()
    obj = DubinsAbs(g, 1.3, 0.7)
    sd = L.Bundle(dict(grid=g, hamFunc=obj.hamiltonian, partialFunc=obj.dissipation))
    f0 = TH.fingerprint(sd)
    assert TH.fingerprint(sd) == f0
    obj.w = 0.9
    assert TH.fingerprint(sd) != f0
    # a table captured by a closure is part of the state as well
    tab = np.linspace(0., 1., 12).reshape(-1, 1, 1)
    sd2 = L.Bundle(dict(grid=g, hamFunc=lambda t, d, p, s: p[0] * tab, partialFunc=lambda t, d, lo, hi, s, dim: 1.0))
    f1 = TH.fingerprint(sd2)
    tab[3] = 5.0
    assert TH.fingerprint(sd2) != f1
    # an object that keeps a module or a class around (`self.np = np`) is not walked into
    obj.np, obj.cls = np, DubinsAbs
    assert len(TH.fingerprint(sd)) < 50


class _Bad(object):
    def __init__(self, grid, kind):
        self.grid, self.kind = grid, kind
        self.c3 = np.cos(np.asarray(grid.xs[2])) * np.asarray(grid.xs[1])          # computed BEFORE the call, from TWO coordinates

    def hamiltonian(self, t, data, p, sd=None):
        k = self.kind
        if k == "branch":
            return p[0] if p[0].max() > 0 else -p[0]
        if k == "bool":
            if p[0] > 0:
                return p[0]
            return p[1]
        if k == "asarray":
            return p[0].numpy() * 2
        if k == "stored":
            return p[0] * self.c3
        if k == "data":
            return p[0] + data
        if k == "time":
            return p[0] * t
        if k == "mask":
            out = np.zeros_like(p[0])
            out[p[0] > 0] = 1.0
            return out
        if k == "reduce":
            return p[0] * np.sum(p[1])
        if k == "raises":
            return p[0] + undefined_name      # noqa: F821
        return p[0]

    def dissipation(self, t, data, dmin, dmax, sd, dim):
        if self.kind == "alpha_p":
            return np.abs(self._p)
        return 1.0


@pytest.mark.parametrize("kind", ["branch", "bool", "asarray", "stored", "data", "time", "mask", "reduce", "raises"])
def test_what_cannot_be_traced_is_refused(kind):
    g = grid3()
    obj = _Bad(g, kind)
    with pytest.raises(TH.TraceError):
        TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    assert isinstance(g.xs[0], np.ndarray)          # restored after a failed trace as well


def test_torch_callbacks_and_stored_coordinate_tensors():
    torch = pytest.importorskip("torch")
    g = grid3()

    class T(object):
        def __init__(self, grid):
            self.grid = grid
            self.x0 = torch.as_tensor(np.asarray(grid.vs[0]).reshape(-1, 1, 1))       # a real tensor that IS a coordinate: recognised by value

        def hamiltonian(self, t, data, p, sd=None):
            x2 = torch.as_tensor(np.asarray(self.grid.xs[2]), device=p[0].device)       # the usual way from the grid to the callback's array type
            return torch.cos(x2) * p[0] + torch.where(p[1] > 0, p[1].abs(), torch.sin(x2) * p[1]) + self.x0 * p[2].clamp(min=-1., max=1.) \
                + torch.maximum(p[0], p[1]) * 0.25 - (p[2] ** 2).sqrt()

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            x2 = torch.as_tensor(np.asarray(self.grid.xs[2]))
            return torch.cos(x2).abs() + 1.0 if dim == 0 else (torch.abs(self.x0 * torch.sin(x2)) if dim == 2 else 2.0)
    obj = T(g)
    tr = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    assert np.asarray is np.asarray.__class__ or not hasattr(np.asarray, "__wrapped__")            # the constructors are the originals again
    assert not hasattr(torch.as_tensor, "__wrapped__")
    rng = np.random.default_rng(5)
    p = [rng.standard_normal(g.shape) for _ in range(3)]
    H, al = tr.evaluate(g.xs, p)
    Hr = obj.hamiltonian(0., None, [torch.as_tensor(q) for q in p]).numpy()
    assert np.allclose(H, Hr, rtol=1e-14, atol=1e-14)
    assert np.allclose(np.broadcast_to(al[0], g.shape), np.broadcast_to(obj.dissipation(0., None, None, None, None, 0).numpy(), g.shape))


def test_stored_per_axis_arrays_become_tables():
    """Arrays computed from ONE coordinate before the call (cos of the heading kept in __init__, a per-plane gain) are written into the source as
    tables and looked up by the node's coordinate; in-plane ones are hoisted into the column expression."""
    g = grid3()

    class Stored(object):
        def __init__(self, grid):
            self.grid = grid
            self.c3 = np.cos(np.asarray(grid.xs[2]))                                         # broadcastable (1, 1, n2)
            self.s3 = np.broadcast_to(np.sin(np.asarray(grid.xs[2])), grid.shape).copy()     # the grid's full shape
            self.gain = np.linspace(1., 2., int(grid.shape[0])).reshape(-1, 1, 1)            # along the marching axis

        def hamiltonian(self, t, data, p, sd=None):
            return p[0] * self.c3 + (p[1] * self.s3) * self.gain + 0.5 * abs(p[2])

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            return [abs(self.c3) + 0 * data, (abs(self.s3) + 0 * data) * self.gain, 0.5][dim]
    obj = Stored(g)
    tr = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    assert len(tr.tables) >= 3 and "hjtab" in tr.source and tr.column_source and "hjtab" in tr.column_source
    rng = np.random.default_rng(8)
    p = [rng.standard_normal(g.shape) for _ in range(3)]
    H, al = tr.evaluate(g.xs, p)
    assert np.array_equal(H, obj.hamiltonian(0., np.zeros(g.shape), p))
    for d in range(3):
        assert np.array_equal(np.broadcast_to(al[d], g.shape), np.broadcast_to(obj.dissipation(0., np.zeros(g.shape), None, None, None, d), g.shape))
    L.register_native_hamiltonian("traced_table_check", 3, tr.source, nparams=len(tr.params), column_src=tr.column_source, ncol=tr.ncol).check("ENO3")
    # a changed table is a changed expression (its values are part of the text), and the fingerprint of the schemeData sees it
    sd = L.Bundle(dict(grid=g, hamFunc=obj.hamiltonian, partialFunc=obj.dissipation))
    f0 = TH.fingerprint(sd)
    obj.gain[3] = 7.0
    assert TH.fingerprint(sd) != f0
    assert TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None).source != tr.source


def test_generated_source_compiles():
    """hipRTC, no device: the text the tracer writes is a valid expression block for every scheme's kernel (fp64 and fp32 instantiate T)."""
    g = grid3()
    obj = Nonlinear(g, 0.4)
    tr = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    reg = L.register_native_hamiltonian("traced_compile_check", 3, tr.source, nparams=len(tr.params), column_src=tr.column_source, ncol=tr.ncol,
                                        uses_range=tr.uses_range)
    reg.check("ENO2")


def test_a_callback_whose_expression_keeps_changing_is_left_alone(monkeypatch):
    """A float beyond the eight parameter slots is a literal: a callback that changes it per call would compile a kernel per call.  After
    MAX_EXPRESSIONS_PER_CALLBACK different texts the pair stays on the split path."""
    g = grid3()

    class Drifting(DubinsAbs):
        gain = 0.0

        def hamiltonian(self, t, data, p, sd=None):
            x = self.grid.xs
            return 0.1 * p[0] + 0.2 * p[1] + 0.3 * p[2] + 0.4 * x[0] + 0.6 * x[1] + 0.9 * x[2] + 1.1 * p[0] * p[1] + 1.3 * p[1] * p[2] \
                + self.gain * p[0] * p[2]      # the ninth float
    obj = Drifting(g, 1., 1.)
    sd = L.Bundle(dict(grid=g, hamFunc=obj.hamiltonian, partialFunc=obj.dissipation))
    monkeypatch.setattr(TH, "MAX_EXPRESSIONS_PER_CALLBACK", 3)
    got = []
    for k in range(5):
        obj.gain = 0.5 + k
        got.append(TH.traced_native(sd) is not None)
    assert got == [True, True, True, False, False]
    obj.gain = 0.5                       # an expression seen before is still served
    assert TH.traced_native(sd) is not None


def test_what_a_callback_caches_during_the_trace_does_not_survive_it():
    """Lazily cached arrays (`self._x0 = ...` on first use, a dict of per-device copies) are symbolic while tracing: the objects' attributes are put
    back afterwards, so the real calls that follow compute their caches from real data."""
    g = grid3()

    class Lazy(object):
        def __init__(self, grid):
            self.grid, self._x0, self.per_device = grid, None, {}

        def x0(self, like):
            if self._x0 is None:
                self._x0 = np.asarray(self.grid.vs[0]).ravel().reshape(-1, 1, 1)
            self.per_device.setdefault("cpu", self._x0)
            self.touched = True
            return self._x0

        def hamiltonian(self, t, data, p, sd=None):
            return 0.7 * self.x0(p[0]) * p[1] + 0.5 * (p[0] ** 2 + p[1] ** 2 + p[2] ** 2)

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            return np.maximum(abs(dmin[dim]), abs(dmax[dim])) + (abs(0.7 * self.x0(data)) if dim == 1 else 0.0)
    obj = Lazy(g)
    tr = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    assert obj._x0 is None and obj.per_device == {} and not hasattr(obj, "touched")
    rng = np.random.default_rng(2)
    p = [rng.standard_normal(g.shape) for _ in range(3)]
    H = obj.hamiltonian(0., None, p)                      # a real call afterwards works on real arrays
    assert isinstance(obj._x0, np.ndarray) and np.allclose(tr.evaluate(g.xs, p, [-1.] * 3, [1.] * 3)[0], H, rtol=1e-14, atol=1e-14)
    # and a failed trace restores as well
    class LazyBad(Lazy):
        def hamiltonian(self, t, data, p, sd=None):
            self.x0(p[0])
            return p[0] if float(p[0].max()) > 0 else p[1]
    bad = LazyBad(g)
    with pytest.raises(TH.TraceError):
        TH.trace_callbacks(g, bad.hamiltonian, bad.dissipation, None)
    assert bad._x0 is None and bad.per_device == {}


class ScalarIdioms(object):
    """partialFunc written for the NUMBERS artificialDissipationGLF hands it (artificial_diss_glf.py:80-88): float(), the builtin max, math.*, a
    conditional expression.  None of these accepts a symbolic argument; the tracer rewrites them in the source for a second attempt."""
    gain = 0.25

    def __init__(self, grid, c):
        self.grid, self.c = grid, c

    def hamiltonian(self, t, data, p, sd=None):
        x0 = np.asarray(self.grid.xs[0])
        return 0.5 * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + self.c * x0 * p[1] + math.cos(0.3) * p[2]

    def dissipation(self, t, data, derivMin, derivMax, sd, dim):
        a = max(abs(float(derivMin[dim])), abs(float(derivMax[dim])))
        a = a + self.gain if dim == 2 else a
        lo = min(float(derivMin[dim]), 0.0)
        a = a + 0.125 * math.fabs(lo)
        if dim != 1:
            return a
        return a + np.abs(self.c * np.asarray(self.grid.xs[0]))


def test_scalar_idioms_are_rewritten_for_a_second_attempt(monkeypatch):
    g = grid3()
    obj = ScalarIdioms(g, 0.7)
    tr = TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)
    assert tr.uses_range
    rng = np.random.default_rng(4)
    p = [rng.standard_normal(g.shape) for _ in range(3)]
    for lo, hi in (([-1.5, -0.5, -2.0], [1.0, 2.5, 0.5]), ([0.25, -3.0, -0.1], [0.5, -1.0, 4.0])):     # both orders of |min| and |max|, min above zero
        H, al = tr.evaluate(g.xs, p, lo, hi)
        assert np.allclose(H, obj.hamiltonian(0., None, p), rtol=1e-14, atol=1e-14)
        for d in range(3):
            ref = np.broadcast_to(np.asarray(obj.dissipation(0., None, lo, hi, None, d), dtype=np.float64), g.shape)
            assert np.allclose(np.broadcast_to(al[d], g.shape), ref, rtol=1e-14, atol=1e-14), d
    # the callbacks themselves are untouched
    assert obj.dissipation(0., None, [-1., -1., -1.], [2., 2., 2.], None, 0) == 2.0 + 0.125
    monkeypatch.setenv("HJ_TRACE_REWRITE", "0")
    with pytest.raises(TH.TraceError):
        TH.trace_callbacks(g, obj.hamiltonian, obj.dissipation, None)


def test_rewrite_keeps_closures_and_refuses_what_it_cannot_fix():
    g = grid3()
    k = 0.4

    def ham(t, data, p, sd):
        return k * p[0] + (p[1] if k > 0 else -p[1]) + 0 * p[2]                      # a free variable, a conditional expression on a number

    def alpha(t, data, lo, hi, sd, dim):
        return max(abs(float(lo[dim])), abs(float(hi[dim]))) + k

    tr = TH.trace_callbacks(g, ham, alpha, None)
    assert tr.uses_range and k in tr.params

    def branchy(t, data, p, sd):
        if float(p[0].max()) > 0:                                                     # an if STATEMENT on array values stays untraceable
            return p[0]
        return -p[0]
    with pytest.raises(TH.TraceError):
        TH.trace_callbacks(g, branchy, alpha, None)


def test_explain_plan_says_which_path_and_why():
    g = grid3()
    mk = lambda o: L.Bundle(dict(grid=g, hamFunc=o.hamiltonian, partialFunc=o.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO2))  # noqa: E731
    e = L.explain_plan(mk(DubinsAbs(g, 1.3, 0.7)))
    assert e["path"] == "traced" and "alpha[2]" in e["source"] and e["params"] == [1.3, 0.7] and e["verified"] is False
    e = L.explain_plan(mk(_Bad(g, "reduce")))
    assert e["path"] == "split" and "reductions" in e["reason"]
    e = L.explain_plan(mk(L.DubinsVehicleRel(g, 1, 1)))
    assert e["path"] == "built-in"
    sd = mk(DubinsAbs(g, 1., 1.))
    sd.dissFunc = lambda *a: None
    assert L.explain_plan(sd)["path"] == "split" and "dissFunc" in L.explain_plan(sd)["reason"]


def _inplace(g):
    def f(t, d, p, sd):
        h = p[0] * 1.0
        h += p[1]
        h *= 2.0
        h -= g.xs[0]
        h /= 4.0
        return h
    return f


IDIOMS = {
    "ufunc calls": lambda g: lambda t, d, p, sd: np.add(np.multiply(p[0], g.xs[1]), np.subtract(p[1], 2.0)),
    "power, float exponent": lambda g: lambda t, d, p, sd: np.power(np.abs(p[0]) + 1.0, 1.5),
    "square, hypot": lambda g: lambda t, d, p, sd: np.square(p[0]) + np.hypot(p[1], p[2]),
    "in-place operators": _inplace,
    "full_like / ones": lambda g: lambda t, d, p, sd: np.full_like(p[0], 0.3) * p[1] + np.ones(g.shape) * 0.5 * p[0],
    "booleans as numbers": lambda g: lambda t, d, p, sd: (p[0] >= 0) * 2.0 * p[0] + (p[0] < 0) * (-1.0) * p[0],
    "where with numbers": lambda g: lambda t, d, p, sd: np.where(p[0] > 0, 1.0, -1.0) * p[1],
    "builtin sum / abs": lambda g: lambda t, d, p, sd: np.sqrt(sum(q ** 2 for q in p) + 1e-9) + abs(p[0]),
    "norm of a stack": lambda g: lambda t, d, p, sd: np.linalg.norm(np.stack(p), axis=0) + np.stack([p[0], p[1]]).sum(axis=0) + np.max(np.stack(p) ** 2, axis=0),
    "arctan2, minimum, maximum": lambda g: lambda t, d, p, sd: np.arctan2(p[0], p[1] + 3.0) + np.minimum(p[0], 0.0) + np.maximum(0.0, p[1]),
    "NumPy scalars": lambda g: lambda t, d, p, sd: np.float64(0.3) * p[0] + np.float32(0.5) * p[1] + np.array(0.25) * p[2],
    "reshape, vs": lambda g: lambda t, d, p, sd: (p[0].reshape(g.shape) * g.xs[0]).reshape(-1).reshape(g.shape) + p[1] * np.asarray(g.vs[0]).reshape(-1, 1, 1),
    "log1p, expm1": lambda g: lambda t, d, p, sd: np.log1p(np.exp(p[0])) + np.expm1(-np.abs(p[1])),
    "mod, floor division": lambda g: lambda t, d, p, sd: np.mod(g.xs[2] + 2.0, 0.75) * p[0] + (p[1] // 0.5) + (p[2] % 0.3),
    "select, heaviside, sinc": lambda g: lambda t, d, p, sd: np.select([p[0] > 0.5, p[0] < -0.5], [p[1], p[2]], 0.0) + np.heaviside(p[0], 0.5) * p[1] + np.sinc(p[2]),
    "a small linear map": lambda g: lambda t, d, p, sd: (np.array([[1., 2.], [3., 4.]]) @ np.array([1., 1.]))[0] * p[0],
    "(ham, schemeData) returned": lambda g: lambda t, d, p, sd: (p[0] * 1.5, sd),
}


@pytest.mark.parametrize("name", sorted(IDIOMS))
def test_array_idioms_trace_and_compute_the_same(name):
    g = grid3()
    f = IDIOMS[name](g)
    tr = TH.trace_callbacks(g, f, lambda t, d, lo, hi, sd, dim: 1.0, None)
    rng = np.random.default_rng(1)
    p = [rng.standard_normal(g.shape) for _ in range(3)]
    H, _ = tr.evaluate(g.xs, p)
    ref = f(0., None, p, None)
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert np.allclose(np.broadcast_to(H, g.shape), ref, rtol=1e-13, atol=1e-13), float(np.abs(H - ref).max())
