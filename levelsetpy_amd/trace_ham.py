"""A Python hamFunc / partialFunc pair TRACED into the device expression of user_ham.register_native_hamiltonian.

The reference's termLaxFriedrichs takes any callables (ExplicitIntegration/Term/term_lax_friedrich.py:111,
Dissipation/artificial_diss_glf.py:98); this package fuses the systems it knows and the ones a user has written once more as a
C expression -- everything else ran the split path at 1/25 of the fused rate (missing 4).  This module closes most of
that gap without asking the user for anything: the callbacks are called ONCE with symbolic stand-ins for their array arguments

    hamFunc(t, data, derivs, schemeData)                          derivs[d]  -> p[d]
    partialFunc(t, data, derivMin, derivMax, schemeData, dim)     derivMin[d], derivMax[d] -> dmin[d], dmax[d]
    schemeData.grid.xs[d] / .vs[d] (read by the callbacks)        -> x[d]

and every NumPy ufunc / torch function / operator they apply is recorded instead of computed.  The recorded expression graph is
written out as the `H = ...; alpha[d] = ...;` block register_native_hamiltonian compiles with hipRTC (Python floats become par[k]
so that changing a speed does not recompile; values that depend on the in-plane coordinates only are hoisted into col[k]).  What
cannot be recorded -- data-dependent `if` statements (`if p.max() > 0:`), reductions, in-place masks, a Hamiltonian that reads `t` or
`data` -- raises TraceError and the schemeData keeps the split path, as before.  (Scalar idioms -- float(derivMin[d]), the builtin max / min,
math.cos, `a if c else b` -- are rewritten in the callbacks' source for a second attempt: see _rewritten.)  Real arrays among the operands are fine if they are a
grid coordinate, constant, or vary along ONE grid axis (a cos(xs[2]) computed in __init__): those become tables in the source text,
looked up by the node's coordinate.

The traced kernel is not trusted blindly: the first time a plan built from it meets data (term.native_plan(sd, y)), the term is
evaluated by BOTH paths on that data and the fused result must agree with the callbacks' to rounding, else the trace is dropped
with a warning.  A cached plan re-traces (0.2 ms of Python) when it is looked up again: parameters changed in place are
picked up like system.native() does for the built-in systems, a changed expression is a new registration.  (Every lookup first
compares a fingerprint of the scalar state the callbacks can see -- attributes of the objects they are bound to, of the schemeData,
closure cells, module globals they name -- and re-traces only when it has changed; HJ_TRACE_RECHECK=1 re-traces always.)

    HJ_TRACE=0 switches the tracer off; HJ_TRACE_VERBOSE=1 says why a schemeData was not traced.
"""
import os
import struct
import threading
import warnings
from types import ModuleType as _ModuleType

import numpy as np

__all__ = ["TraceError", "trace_callbacks", "traced_native", "Traced", "fingerprint"]


class TraceError(Exception):
    """The callbacks did something that has no per-node device expression."""


_TLS = threading.local()


def _tracer():
    t = getattr(_TLS, "tracer", None)
    if t is None:
        raise TraceError("a symbolic array was used outside the trace that made it")
    return t


# ---------------------------------------------------------------------------------------------- the expression graph
_NUM_UNARY = {"neg": "(-{0})", "abs": "fabs({0})", "cos": "cos({0})", "sin": "sin({0})", "tan": "tan({0})", "exp": "exp({0})", "log": "log({0})",
              "sqrt": "sqrt({0})", "tanh": "tanh({0})", "sinh": "sinh({0})", "cosh": "cosh({0})", "asin": "asin({0})", "acos": "acos({0})",
              "atan": "atan({0})", "floor": "floor({0})", "ceil": "ceil({0})", "square": "({0} * {0})", "recip": "(T(1) / {0})",
              "log1p": "log1p({0})", "expm1": "expm1({0})", "log2": "log2({0})", "log10": "log10({0})",
              "sign": "(T)(({0} > T(0)) - ({0} < T(0)))", "cast": "({0} ? T(1) : T(0))"}
_NUM_BINARY = {"add": "({0} + {1})", "sub": "({0} - {1})", "mul": "({0} * {1})", "div": "({0} / {1})", "pow": "pow({0}, {1})",
               "atan2": "atan2({0}, {1})", "hypot": "hypot({0}, {1})", "fmod": "fmod({0}, {1})",
               # NumPy / torch maximum and minimum hand a NaN on, fmax / fmin do not
               "max": "((({0} >= {1}) || ({0} != {0})) ? {0} : {1})", "min": "((({0} <= {1}) || ({0} != {0})) ? {0} : {1})"}
_CMP = {"gt": ">", "ge": ">=", "lt": "<", "le": "<=", "eq": "==", "ne": "!="}
_BOOL_BINARY = {"and": "({0} && {1})", "or": "({0} || {1})", "xor": "({0} != {1})"}
_EXPENSIVE = {"log1p", "expm1", "log2", "log10", "cos", "sin", "tan", "exp", "log", "sqrt", "tanh", "sinh", "cosh", "asin", "acos", "atan", "pow", "atan2", "hypot", "div", "recip", "fmod"}

_NP_EVAL = {"neg": np.negative, "abs": np.abs, "cos": np.cos, "sin": np.sin, "tan": np.tan, "exp": np.exp, "log": np.log, "sqrt": np.sqrt,
            "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh, "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan, "floor": np.floor,
            "ceil": np.ceil, "square": np.square, "recip": np.reciprocal, "sign": np.sign, "log1p": np.log1p, "expm1": np.expm1, "log2": np.log2,
            "log10": np.log10, "add": np.add, "sub": np.subtract, "mul": np.multiply,
            "div": np.divide, "pow": np.power, "atan2": np.arctan2, "hypot": np.hypot, "fmod": np.fmod, "max": np.maximum, "min": np.minimum,
            "gt": np.greater, "ge": np.greater_equal, "lt": np.less, "le": np.less_equal, "eq": np.equal, "ne": np.not_equal,
            "and": np.logical_and, "or": np.logical_or, "xor": np.logical_xor, "not": np.logical_not}


def _noop(name):
    def f(self, *a, **k):
        return self
    f.__name__ = name
    return f


def _refuse(name, why):
    def f(self, *a, **k):
        raise TraceError("%s on a symbolic array: %s" % (name, why))
    f.__name__ = name
    return f


class Sym(object):
    """A node of the expression graph standing in for an array of the grid's shape.  kind: 'num' or 'bool'."""
    __array_priority__ = 10000.0
    __slots__ = ("op", "args", "kind", "uid", "value", "__weakref__")

    def __init__(self, op, args, kind, uid, value=None):
        self.op, self.args, self.kind, self.uid, self.value = op, args, kind, uid, value

    # ---- what array code asks an array about itself
    @property
    def shape(self):
        return _tracer().shape

    @property
    def ndim(self):
        return len(_tracer().shape)

    @property
    def size(self):
        return int(np.prod(_tracer().shape))

    @property
    def dtype(self):
        return np.dtype(np.bool_ if self.kind == "bool" else np.float64)

    @property
    def device(self):
        return "cpu"

    @property
    def T(self):
        raise TraceError("transposing a symbolic array")

    def dim(self):
        return len(_tracer().shape)

    def numel(self):
        return self.size

    def __len__(self):
        return int(_tracer().shape[0])

    def __repr__(self):
        return "<Sym %s #%d>" % (self.op, self.uid)

    # ---- operators
    def __add__(self, o): return _bin("add", self, o)
    def __radd__(self, o): return _bin("add", o, self)
    def __sub__(self, o): return _bin("sub", self, o)
    def __rsub__(self, o): return _bin("sub", o, self)
    def __mul__(self, o): return _bin("mul", self, o)
    def __rmul__(self, o): return _bin("mul", o, self)
    def __truediv__(self, o): return _bin("div", self, o)
    def __rtruediv__(self, o): return _bin("div", o, self)
    def __pow__(self, o): return _pow(self, o)
    def __rpow__(self, o): return _pow(o, self)
    def __floordiv__(self, o): return _apply("floor_divide", (self, o))
    def __rfloordiv__(self, o): return _apply("floor_divide", (o, self))
    def __mod__(self, o): return _apply("remainder", (self, o))
    def __rmod__(self, o): return _apply("remainder", (o, self))
    def __neg__(self): return _un("neg", self)
    def __pos__(self): return self
    def __abs__(self): return _un("abs", self)
    def __gt__(self, o): return _cmp("gt", self, o)
    def __ge__(self, o): return _cmp("ge", self, o)
    def __lt__(self, o): return _cmp("lt", self, o)
    def __le__(self, o): return _cmp("le", self, o)
    def __eq__(self, o): return _cmp("eq", self, o)
    def __ne__(self, o): return _cmp("ne", self, o)
    def __and__(self, o): return _boolop("and", self, o)
    def __rand__(self, o): return _boolop("and", o, self)
    def __or__(self, o): return _boolop("or", self, o)
    def __ror__(self, o): return _boolop("or", o, self)
    def __xor__(self, o): return _boolop("xor", self, o)
    def __rxor__(self, o): return _boolop("xor", o, self)
    def __invert__(self): return _not(self)
    __hash__ = object.__hash__

    __bool__ = _refuse("bool()", "Python control flow cannot depend on array values in a fused kernel (use where())")
    __float__ = _refuse("float()", "a per-node value has no single number")
    __int__ = _refuse("int()", "a per-node value has no single number")
    __index__ = _refuse("index", "a per-node value has no single number")
    __iter__ = _refuse("iteration", "not supported")
    __setitem__ = _refuse("item assignment", "in-place masks have no per-node expression (use where())")
    __iadd__ = __add__
    __isub__ = __sub__
    __imul__ = __mul__
    __itruediv__ = __truediv__

    def __getitem__(self, key):
        # x[...], x[:], x[:, :, :]: the whole array.  Anything that selects is refused.
        ks = key if isinstance(key, tuple) else (key,)
        for k in ks:
            if k is Ellipsis or k is None or (isinstance(k, slice) and k == slice(None)):
                continue
            raise TraceError("indexing a symbolic array with %r" % (key,))
        return self

    def __array__(self, *a, **k):
        raise TraceError("a symbolic array was converted to a NumPy array (np.asarray / np.array): the values are not known while tracing")

    # ---- methods of ndarray / Tensor that keep the value
    for _n in ("reshape", "view", "squeeze", "unsqueeze", "flatten", "ravel", "expand", "expand_as", "broadcast_to", "to", "double", "float", "clone", "copy",
               "contiguous", "detach", "cpu", "cuda", "astype", "type", "type_as", "requires_grad_", "view_as", "reshape_as"):
        locals()[_n] = _noop(_n)
    del _n
    for _n in ("max", "min", "sum", "mean", "prod", "any", "all", "item", "tolist", "numpy", "argmax", "argmin", "norm", "amax", "amin", "std", "var",
               "cumsum", "nonzero", "fill_", "zero_", "copy_", "masked_fill_", "masked_fill", "index_put_"):
        locals()[_n] = _refuse(_n + "()", "reductions and in-place writes have no per-node expression")
    del _n

    def abs(self): return _un("abs", self)
    def cos(self): return _un("cos", self)
    def sin(self): return _un("sin", self)
    def tan(self): return _un("tan", self)
    def exp(self): return _un("exp", self)
    def log(self): return _un("log", self)
    def sqrt(self): return _un("sqrt", self)
    def tanh(self): return _un("tanh", self)
    def square(self): return _un("square", self)
    def sign(self): return _un("sign", self)
    def neg(self): return _un("neg", self)
    def reciprocal(self): return _un("recip", self)
    def pow(self, o): return _pow(self, o)
    def add(self, o): return _bin("add", self, o)
    def sub(self, o): return _bin("sub", self, o)
    def mul(self, o): return _bin("mul", self, o)
    def div(self, o): return _bin("div", self, o)
    def maximum(self, o): return _bin("max", self, o)
    def minimum(self, o): return _bin("min", self, o)
    def logical_not(self): return _not(self)
    def logical_and(self, o): return _boolop("and", self, o)
    def logical_or(self, o): return _boolop("or", self, o)

    def clamp(self, min=None, max=None):
        r = self
        if min is not None:
            r = _bin("max", r, min)
        if max is not None:
            r = _bin("min", r, max)
        return r
    clip = clamp

    def clamp_min(self, v): return _bin("max", self, v)
    def clamp_max(self, v): return _bin("min", self, v)

    # ---- NumPy dispatch
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None or kwargs.get("where", True) is not True:
            raise TraceError("np.%s.%s / out= / where= on a symbolic array" % (ufunc.__name__, method))
        return _apply(ufunc.__name__, inputs)

    def __array_function__(self, func, types, args, kwargs):
        return _apply_function(func.__name__, args, kwargs)

    # ---- torch dispatch (torch.cos(sym), tensor * sym, torch.where(...), ...)
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        return _apply_function(name.strip("_") if name.startswith("__") else name, args, kwargs or {})


class CoordSym(Sym):
    """grid.xs[d] / grid.vs[d] during a trace."""
    __slots__ = ("axis",)


# ufunc / function names -> graph ops
_UNARY_NAMES = {"negative": "neg", "neg": "neg", "absolute": "abs", "abs": "abs", "fabs": "abs", "cos": "cos", "sin": "sin", "tan": "tan", "exp": "exp",
                "log": "log", "sqrt": "sqrt", "tanh": "tanh", "sinh": "sinh", "cosh": "cosh", "arcsin": "asin", "asin": "asin", "arccos": "acos",
                "acos": "acos", "arctan": "atan", "atan": "atan", "floor": "floor", "ceil": "ceil", "square": "square", "reciprocal": "recip",
                "sign": "sign", "sgn": "sign", "log1p": "log1p", "expm1": "expm1", "log2": "log2", "log10": "log10"}
_BINARY_NAMES = {"add": "add", "subtract": "sub", "sub": "sub", "multiply": "mul", "mul": "mul", "divide": "div", "true_divide": "div", "div": "div",
                 "truediv": "div", "maximum": "max", "fmax": "max", "minimum": "min", "fmin": "min", "arctan2": "atan2", "atan2": "atan2",
                 "hypot": "hypot", "fmod": "fmod", "radd": "add", "rmul": "mul"}
_CMP_NAMES = {"greater": "gt", "gt": "gt", "greater_equal": "ge", "ge": "ge", "less": "lt", "lt": "lt", "less_equal": "le", "le": "le",
              "equal": "eq", "eq": "eq", "not_equal": "ne", "ne": "ne"}
_BOOL_NAMES = {"logical_and": "and", "bitwise_and": "and", "and": "and", "logical_or": "or", "bitwise_or": "or", "or": "or", "logical_xor": "xor",
               "bitwise_xor": "xor", "xor": "xor"}
_IDENTITY_NAMES = {"positive", "pos", "reshape", "squeeze", "unsqueeze", "ravel", "flatten", "broadcast_to", "ascontiguousarray", "asanyarray", "copy",
                   "clone", "contiguous", "detach", "expand", "expand_as", "view", "atleast_1d", "atleast_2d", "atleast_3d", "real", "to", "double",
                   "float", "type_as", "view_as", "reshape_as"}


def _apply(name, inputs):
    if name in _UNARY_NAMES:
        return _un(_UNARY_NAMES[name], inputs[0])
    if name in ("power", "pow", "float_power"):
        return _pow(inputs[0], inputs[1])
    if name in ("rsub",):
        return _bin("sub", inputs[1], inputs[0])
    if name in ("rtruediv", "rdiv"):
        return _bin("div", inputs[1], inputs[0])
    if name in ("rpow",):
        return _pow(inputs[1], inputs[0])
    if name in _BINARY_NAMES:
        return _bin(_BINARY_NAMES[name], inputs[0], inputs[1])
    if name in _CMP_NAMES:
        return _cmp(_CMP_NAMES[name], inputs[0], inputs[1])
    if name in _BOOL_NAMES:
        return _boolop(_BOOL_NAMES[name], inputs[0], inputs[1])
    if name in ("logical_not", "invert", "bitwise_not"):
        return _not(inputs[0])
    if name in ("floor_divide", "floordiv"):
        return _un("floor", _bin("div", inputs[0], inputs[1]))
    if name in ("rfloordiv",):
        return _un("floor", _bin("div", inputs[1], inputs[0]))
    if name in ("remainder", "mod"):                      # NumPy / Python: the sign of the divisor -- a - floor(a / b) b
        return _bin("sub", inputs[0], _bin("mul", _un("floor", _bin("div", inputs[0], inputs[1])), inputs[1]))
    if name in ("rmod",):
        return _apply("remainder", (inputs[1], inputs[0]))
    if name == "heaviside":
        x = _num(inputs[0])
        return _where(_cmp("gt", x, 0), 1, _where(_cmp("lt", x, 0), 0, inputs[1]))
    raise TraceError("np.%s / torch.%s has no device expression here" % (name, name))


def _apply_function(name, args, kwargs):
    if kwargs.get("out") is not None:
        raise TraceError("%s(out=...) on a symbolic array" % name)
    if name == "where":
        if len(args) != 3:
            raise TraceError("where(cond) without values selects nodes: no per-node expression")
        return _where(args[0], args[1], args[2])
    if name in ("clip", "clamp"):
        lo = args[1] if len(args) > 1 else kwargs.get("min", kwargs.get("a_min"))
        hi = args[2] if len(args) > 2 else kwargs.get("max", kwargs.get("a_max"))
        r = _lift(args[0])
        if lo is not None:
            r = _bin("max", r, lo)
        if hi is not None:
            r = _bin("min", r, hi)
        return r
    if name in ("clamp_min",):
        return _bin("max", args[0], args[1])
    if name in ("clamp_max",):
        return _bin("min", args[0], args[1])
    if name in ("zeros_like", "ones_like", "full_like"):
        return _tracer().const(0 if name == "zeros_like" else 1) if name != "full_like" else _lift(args[1])
    if name in _IDENTITY_NAMES:
        return _lift(args[0])
    if name in ("max", "min", "amax", "amin") and len(args) == 2 and not isinstance(args[1], (int, tuple)):
        return _bin(name[-3:], args[0], args[1])          # torch.max(a, b): the binary form
    if name in ("stack", "vstack") and len(args) >= 1 and isinstance(args[0], (list, tuple)):
        if kwargs.get("axis", kwargs.get("dim", 0)) not in (0, None) or (len(args) > 1 and args[1] != 0):
            raise TraceError("stack along an axis other than 0")
        return SymStack([_num(a) for a in args[0]])
    if name == "select" and len(args) >= 2:
        r = _lift(args[2] if len(args) > 2 else kwargs.get("default", 0))
        for c, v in reversed(list(zip(args[0], args[1]))):
            r = _where(c, v, r)
        return r
    if name == "sinc":
        x = _bin("mul", args[0], math_pi())
        return _where(_cmp("eq", x, 0), 1, _bin("div", _un("sin", x), x))
    if name in ("sum", "mean", "max", "min", "amax", "amin", "prod", "any", "all", "norm", "argmax", "argmin"):
        raise TraceError("%s() of a symbolic array: reductions have no per-node expression" % name)
    return _apply(name, args)


def math_pi():
    return 3.141592653589793


class SymStack(object):
    """np.stack([a, b, c]) of symbolic arrays: a short list along a NEW leading axis -- the one kind of reduction that has a per-node expression
    (|p| = np.linalg.norm(np.stack(p), axis=0); np.stack(...).sum(0) / .max(0) / .min(0)).  Element-wise operations act on every member."""
    __array_priority__ = 20000.0

    def __init__(self, members):
        self.members = list(members)

    def __len__(self):
        return len(self.members)

    def __getitem__(self, k):
        if isinstance(k, (int, np.integer)):
            return self.members[int(k)]
        if isinstance(k, slice):
            return SymStack(self.members[k])
        raise TraceError("indexing a stack of symbolic arrays with %r" % (k,))

    def __iter__(self):
        return iter(self.members)

    @property
    def shape(self):
        return (len(self.members),) + tuple(_tracer().shape)

    def _map(self, f):
        return SymStack([f(m) for m in self.members])

    def _zip(self, o, f):
        if isinstance(o, SymStack):
            if len(o.members) != len(self.members):
                raise TraceError("stacks of different lengths")
            return SymStack([f(a, b) for a, b in zip(self.members, o.members)])
        return SymStack([f(a, o) for a in self.members])

    def __add__(self, o): return self._zip(o, lambda a, b: _bin("add", a, b))
    __radd__ = __add__
    def __sub__(self, o): return self._zip(o, lambda a, b: _bin("sub", a, b))
    def __rsub__(self, o): return self._zip(o, lambda a, b: _bin("sub", b, a))
    def __mul__(self, o): return self._zip(o, lambda a, b: _bin("mul", a, b))
    __rmul__ = __mul__
    def __truediv__(self, o): return self._zip(o, lambda a, b: _bin("div", a, b))
    def __rtruediv__(self, o): return self._zip(o, lambda a, b: _bin("div", b, a))
    def __pow__(self, o): return self._map(lambda a: _pow(a, o))
    def __neg__(self): return self._map(lambda a: _un("neg", a))
    def __abs__(self): return self._map(lambda a: _un("abs", a))

    def _axis0(self, axis, what):
        if axis not in (0,):
            raise TraceError("%s of a stack along axis %r: only the stacking axis (0) has a per-node expression" % (what, axis))

    def _fold(self, op):
        r = self.members[0]
        for m in self.members[1:]:
            r = _bin(op, r, m)
        return r

    def sum(self, axis=None, dim=None, **kw):
        self._axis0(axis if dim is None else dim, "sum")
        return self._fold("add")

    def prod(self, axis=None, dim=None, **kw):
        self._axis0(axis if dim is None else dim, "prod")
        return self._fold("mul")

    def max(self, axis=None, dim=None, **kw):
        self._axis0(axis if dim is None else dim, "max")
        return self._fold("max")

    def min(self, axis=None, dim=None, **kw):
        self._axis0(axis if dim is None else dim, "min")
        return self._fold("min")

    def mean(self, axis=None, dim=None, **kw):
        self._axis0(axis if dim is None else dim, "mean")
        return _bin("div", self._fold("add"), len(self.members))

    amax, amin = max, min

    def abs(self): return abs(self)
    def square(self): return self._map(lambda a: _un("square", a))
    def pow(self, o): return self ** o

    def norm(self, p=2, dim=None, axis=None, **kw):
        return self.__array_function__(np.linalg.norm, (), (self,), {"axis": dim if dim is not None else axis, "ord": p})

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if kwargs.get("out") is not None:
            raise TraceError("out= on a stack of symbolic arrays")
        if method == "reduce":
            self._axis0(kwargs.get("axis", 0), ufunc.__name__ + ".reduce")
            op = {"add": "add", "multiply": "mul", "maximum": "max", "minimum": "min", "fmax": "max", "fmin": "min"}.get(ufunc.__name__)
            if op is None:
                raise TraceError("np.%s.reduce of a stack" % ufunc.__name__)
            return inputs[0]._fold(op)
        if method != "__call__":
            raise TraceError("np.%s.%s of a stack of symbolic arrays" % (ufunc.__name__, method))
        n = len(self.members)
        cols = [[(a.members[k] if isinstance(a, SymStack) else a) for a in inputs] for k in range(n)]
        return SymStack([_apply(ufunc.__name__, c) for c in cols])

    def __array_function__(self, func, types, args, kwargs):
        name = func.__name__
        if name in ("sum", "prod", "max", "min", "amax", "amin", "mean"):
            return getattr(self, name)(kwargs.get("axis", args[1] if len(args) > 1 else None))
        if name == "norm":
            order = kwargs.get("ord", args[1] if len(args) > 1 else None)
            self._axis0(kwargs.get("axis", args[2] if len(args) > 2 else None), "norm")
            if order in (None, 2):
                return _un("sqrt", self._map(lambda a: _un("square", a))._fold("add"))
            if order == 1:
                return self._map(lambda a: _un("abs", a))._fold("add")
            if order == np.inf:
                return self._map(lambda a: _un("abs", a))._fold("max")
            raise TraceError("np.linalg.norm of order %r" % (order,))
        raise TraceError("np.%s of a stack of symbolic arrays" % name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))
        st = next(a for a in args if isinstance(a, SymStack))
        if name in ("sum", "prod", "amax", "amin", "mean"):
            return getattr(st, name)(kwargs.get("dim", args[1] if len(args) > 1 else None))
        if name in ("max", "min") and (len(args) > 1 or "dim" in kwargs):
            raise TraceError("torch.%s(stack, dim) returns indices as well: use amax / amin" % name)
        if name in ("norm", "vector_norm", "linalg_vector_norm"):
            return st.__array_function__(np.linalg.norm, (), (st,), {"axis": kwargs.get("dim", args[2] if len(args) > 2 else None),
                                                                        "ord": kwargs.get("p", kwargs.get("ord", args[1] if len(args) > 1 else None))})
        raise TraceError("torch.%s of a stack of symbolic arrays" % name)


# ---------------------------------------------------------------------------------------------- graph construction
def _lift(v):
    return _tracer().lift(v)


def _num(v):
    s = _lift(v)
    return _tracer().node("cast", (s,), "num") if s.kind == "bool" else s


def _bool(v):
    s = _lift(v)
    if s.kind == "bool":
        return s
    return _tracer().node("ne", (s, _tracer().const(0)), "bool")


def _un(op, a):
    return _tracer().node(op, (_num(a),), "num")


def _is_const(s, v):
    return s.op == "const" and s.value == v


def _bin(op, a, b):
    a, b = _num(a), _num(b)
    # `alpha + 0 * data`, `np.zeros_like(data) + c`, `np.ones_like(p) * c`: the idioms that give a result the grid's shape
    if op == "mul":
        if _is_const(a, 0.0) or _is_const(b, 0.0):
            return _tracer().const(0)
        if _is_const(a, 1.0):
            return b
        if _is_const(b, 1.0):
            return a
    elif op == "add":
        if _is_const(a, 0.0):
            return b
        if _is_const(b, 0.0):
            return a
    elif op == "sub" and _is_const(b, 0.0):
        return a
    elif op == "div" and _is_const(b, 1.0):
        return a
    return _tracer().node(op, (a, b), "num")


def _cmp(op, a, b):
    return _tracer().node(op, (_num(a), _num(b)), "bool")


def _boolop(op, a, b):
    return _tracer().node(op, (_bool(a), _bool(b)), "bool")


def _not(a):
    return _tracer().node("not", (_bool(a),), "bool")


def _where(c, a, b):
    return _tracer().node("where", (_bool(c), _num(a), _num(b)), "num")


def _pow(a, b):
    # NumPy squares for ** 2 and takes the root for ** 0.5; the device expression does the same so that the two agree to rounding
    if isinstance(b, (int, np.integer)) and not isinstance(b, (bool, np.bool_)):
        if int(b) == 2:
            return _un("square", a)
        if int(b) == 1:
            return _num(a)
        if int(b) == 0:
            return _tracer().const(1)
        if int(b) == -1:
            return _un("recip", a)
        if int(b) == 3:
            return _bin("mul", _un("square", a), a)
    if isinstance(b, (float, np.floating)) and float(b) == 0.5:
        return _un("sqrt", a)
    if isinstance(b, (float, np.floating)) and float(b) == 2.0:
        return _un("square", a)
    return _bin("pow", a, b)


def _is_tensor(v):
    return not isinstance(v, type) and type(v).__module__.split(".")[0] == "torch" and hasattr(v, "data_ptr") and hasattr(v, "numel")


class _Tracer(object):
    MAX_PAR = 8          # hj_ham_register2: a run-time Hamiltonian takes 0..8 parameters (HamTables::par)
    MAX_COL = 8
    MAX_TAB = 8          # per-axis tables written into the source (their values are part of the expression's text)
    MAX_TAB_LEN = 4096

    def __init__(self, grid):
        self.grid = grid
        self.dim = int(grid.dim)
        self.shape = tuple(int(v) for v in np.asarray(grid.shape).ravel()) if hasattr(grid, "shape") else tuple(int(v) for v in np.asarray(grid.N).ravel())
        self.nodes = {}
        self.order = []
        self.params = []          # values of par[k], in order of first use
        self.param_node = {}
        self.real_vs = [np.asarray(v, dtype=np.float64).ravel() for v in grid.vs]
        self.x = [self._leaf(CoordSym, "x", d) for d in range(self.dim)]
        for d, s in enumerate(self.x):
            s.axis = d
        self.p = [self._leaf(Sym, "p", d) for d in range(self.dim)]
        self.dmin = [self._leaf(Sym, "dmin", d) for d in range(self.dim)]
        self.dmax = [self._leaf(Sym, "dmax", d) for d in range(self.dim)]
        self.data = self._leaf(Sym, "data", 0)
        self.t = self._leaf(Sym, "t", 0)
        self.coord_hits = {}
        self.tables = []          # (axis, values): real arrays among the operands that vary along one axis
        self.table_index = {}

    def _leaf(self, cls, op, d):
        s = cls(op, (), "num", len(self.order), d)
        self.nodes[(op, d)] = s
        self.order.append(s)
        return s

    def node(self, op, args, kind):
        key = (op,) + tuple(a.uid for a in args)
        s = self.nodes.get(key)
        if s is None:
            s = Sym(op, args, kind, len(self.order))
            self.nodes[key] = s
            self.order.append(s)
        return s

    def const(self, v):
        """A literal written into the source (Python ints and booleans: structure, not parameters)."""
        key = ("const", repr(float(v)))
        s = self.nodes.get(key)
        if s is None:
            s = Sym("const", (), "num", len(self.order), float(v))
            self.nodes[key] = s
            self.order.append(s)
        return s

    def param(self, v):
        """A Python float: par[k] for the first MAX_PAR distinct values (changing it does not recompile), a literal after that."""
        v = float(v)
        key = struct.pack("<d", v)
        s = self.param_node.get(key)
        if s is None:
            if len(self.params) < self.MAX_PAR:
                s = Sym("par", (), "num", len(self.order), len(self.params))
                self.params.append(v)
                self.nodes[("par", len(self.params) - 1)] = s
                self.order.append(s)
            else:
                s = Sym("const", (), "num", len(self.order), v)
                self.nodes[("lit", key)] = s
                self.order.append(s)
            self.param_node[key] = s
        return s

    def lift(self, v):
        if isinstance(v, Sym):
            return v
        if isinstance(v, (bool, np.bool_, int, np.integer)):
            return self.const(int(v))
        if isinstance(v, (float, np.floating)):
            return self.param(v)
        if isinstance(v, np.ndarray):
            if v.size == 1:
                return self.param(float(v.reshape(-1)[0])) if v.dtype.kind == "f" else self.const(int(v.reshape(-1)[0]))
            return self._coordinate(v, np.asarray(v))
        if _is_tensor(v):
            if v.numel() == 1:
                return self.param(float(v.reshape(-1)[0].item())) if v.dtype.is_floating_point else self.const(int(v.reshape(-1)[0].item()))
            return self._coordinate(v, None)
        if isinstance(v, SymStack):
            raise TraceError("a stack of symbolic arrays where ONE array was expected (reduce it along axis 0 first: .sum(0), .max(0), norm(axis=0))")
        if isinstance(v, (list, tuple)):
            raise TraceError("a %s where an array was expected" % type(v).__name__)
        try:
            return self.param(float(v))
        except TraceError:
            raise
        except Exception:
            raise TraceError("an operand of type %s cannot be traced" % type(v).__name__)

    def _coordinate(self, v, host):
        """A real array among the operands: fine if it IS a grid coordinate (grid.xs[d] read before the trace, a cached device copy)."""
        hit = self.coord_hits.get(id(v))
        if hit is not None and hit[0] is v:
            return hit[1][1] if isinstance(hit[1], tuple) else self.x[hit[1]]
        shape = tuple(int(s) for s in v.shape)
        full = shape == self.shape
        cand = []
        for d in range(self.dim):
            one = tuple(self.shape[k] if k == d else 1 for k in range(self.dim))
            if full or shape == one or (len(shape) == 1 and shape[0] == self.shape[d]) or (shape == (self.shape[d], 1)):
                cand.append(d)
        for d in cand:
            ref = self.real_vs[d]
            bshape = tuple(self.shape[k] if k == d else 1 for k in range(self.dim))
            if host is not None:
                a = host.reshape(bshape) if host.size == ref.size else host
                # (compared in the ARRAY's floating type -- fp32 copies of the coordinates match -- but never in an integer or boolean one)
                ct = a.dtype if a.dtype.kind == "f" else np.float64
                ok = a.shape[d] == ref.size and bool(np.all(a.astype(ct, copy=False) == ref.reshape(bshape).astype(ct, copy=False)))
            else:
                import torch
                ct = v.dtype if v.dtype.is_floating_point else torch.float64
                r = torch.as_tensor(ref, device=v.device, dtype=ct).reshape(bshape)
                a = (v.reshape(bshape) if v.numel() == ref.size else v).to(ct)
                ok = a.shape[d] == ref.size and bool((a == r).all().item())
            if ok:
                self.coord_hits[id(v)] = (v, d)
                return self.x[d]
        # an array of one value (torch.ones(shape) * c, a filled alpha): a number
        if host is not None:
            lo, hi = host.min(), host.max()
        else:
            lo, hi = v.min().item(), v.max().item()
        if lo == hi:
            return self.param(float(lo)) if float(lo) != int(lo) or abs(lo) > 16 else self.const(int(lo))
        # an array that varies along ONE grid axis (a cos(xs[2]) stored before the call, a per-heading gain): a table over that axis, looked
        # up by the node's coordinate
        for d in cand:
            bshape = tuple(self.shape[k] if k == d else 1 for k in range(self.dim))
            first = tuple(slice(None) if k == d else slice(0, 1) for k in range(self.dim))
            if host is not None:
                a = host.reshape(bshape) if host.size == self.shape[d] else host
                line = a[first]
                ok = bool(np.all((a == line) | ((a != a) & (line != line))))
                vals = np.asarray(line, dtype=np.float64).ravel()
            else:
                a = v.reshape(bshape) if v.numel() == self.shape[d] else v
                line = a[first]
                ok = bool(((a == line) | ((a != a) & (line != line))).all().item())
                vals = line.detach().double().cpu().numpy().ravel()
            if ok and vals.size == self.shape[d] and vals.size <= self.MAX_TAB_LEN and bool(np.all(np.isfinite(vals))):
                key = (d, vals.tobytes())
                k = self.table_index.get(key)
                if k is None:
                    if len(self.tables) >= self.MAX_TAB:
                        break
                    k = len(self.tables)
                    self.tables.append((d, vals))
                    self.table_index[key] = k
                s = self.node("tab%d" % k, (self.x[d],), "num")
                self.coord_hits[id(v)] = (v, ("tab", s))
                return s
        raise TraceError("an array of shape %s that is neither a grid coordinate, nor constant, nor a function of ONE grid axis was combined with the "
                         "symbolic arguments (values computed from several coordinates BEFORE the call cannot be traced: compute them inside "
                         "the callback)" % (shape,))


# ---------------------------------------------------------------------------------------------- the result of a trace
def _literal(v):
    if v == int(v) and abs(v) < 1e15:
        return "T(%d)" % int(v)
    if v != v:
        return "T(NAN)"
    if v in (float("inf"), float("-inf")):
        return "T(%sINFINITY)" % ("-" if v < 0 else "")
    return "T(%s)" % float(v).hex()             # exact


class Traced(object):
    """source, column_source, ncol, params, uses_range of a traced callback pair; evaluate() runs the same graph in NumPy."""

    def __init__(self, tr, H, alpha):
        self.dim = tr.dim
        self.params = list(tr.params)
        self.tables = list(tr.tables)
        vs = tr.real_vs
        # index of a node on axis d from its coordinate: the grid is uniform (grids.py), x = x0 + i dx
        self._tab_geo = [(float(vs[d][0]), (len(vs[d]) - 1) / float(vs[d][-1] - vs[d][0]) if len(vs[d]) > 1 else 0.0, len(vs[d])) for d, _ in tr.tables]
        self._H, self._alpha, self._order = H, alpha, tr.order
        deps = {}
        for s in tr.order:                   # construction order is topological
            if s.op in ("x", "p", "dmin", "dmax", "data", "t"):
                deps[s.uid] = frozenset([(s.op, s.value)])
            else:
                d = frozenset()
                for a in s.args:
                    d = d | deps[a.uid]
                deps[s.uid] = d
        live = set()
        stack = [H] + list(alpha)
        while stack:
            s = stack.pop()
            if s.uid in live:
                continue
            live.add(s.uid)
            stack.extend(s.args)
        used = frozenset().union(*[deps[s.uid] for s in [H] + list(alpha)])
        for what, msg in (("data", "the value function itself (data)"), ("t", "the time t")):
            if any(k[0] == what for k in used):
                raise TraceError("the callbacks read %s: the fused kernels take H(x, p) and alpha(x, range) only" % msg)
        if any(k[0] in ("dmin", "dmax") for k in deps[H.uid]):
            raise TraceError("hamFunc reads the costate range")
        if any(k[0] == "p" for a in alpha for k in deps[a.uid]):
            raise TraceError("partialFunc reads the costate itself (only its range, derivMin / derivMax, reaches a fused alpha)")
        self.uses_range = any(k[0] in ("dmin", "dmax") for a in alpha for k in deps[a.uid])
        # values of the in-plane coordinates alone that cost something: once per grid column (col[k])
        def column_only(s):
            dd = deps[s.uid]
            return len(dd) > 0 and all(k[0] == "x" and k[1] >= 1 for k in dd)
        def costly(s, seen):
            if s.uid in seen:
                return False
            seen.add(s.uid)
            return s.op in _EXPENSIVE or s.op.startswith("tab") or any(costly(a, seen) for a in s.args)
        parents = {}
        for s in tr.order:
            if s.uid in live:
                for a in s.args:
                    parents.setdefault(a.uid, []).append(s)
        # candidates: the MAXIMAL column-only values (a parent that is not column-only reads them) that contain a transcendental, a division or
        # a table.  HJ_TRACE_HOIST=all adds plain arithmetic of two operations or more (alpha_0 = |a - b cos x2| + |w x1| of a Dubins car, five
        # flops per node and plane, as the built-in kernel's Cell keeps it): measured at 201^3, no difference (0.1168-0.1189 ms either way) --
        # the substep is not bound by those flops; off by default, a column value costs registers for the length of the march
        def weight(s, seen):
            if s.uid in seen or not s.args:
                return 0
            seen.add(s.uid)
            return 1 + sum(weight(a, seen) for a in s.args)
        hoist_all = os.environ.get("HJ_TRACE_HOIST", "costly") == "all"
        cand = []
        for s in tr.order:
            if s.uid in live and s.kind == "num" and s.args and column_only(s):
                ps = parents.get(s.uid, [])
                if not ps or any(not column_only(q) for q in ps):
                    c_, w_ = costly(s, set()), weight(s, set())
                    if c_ or (hoist_all and w_ >= 2):
                        cand.append((0 if c_ else 1, -w_, s.uid, s))
        cols = [c[3] for c in sorted(cand)[:_Tracer.MAX_COL]]
        cols.sort(key=lambda s: s.uid)
        self._cols = cols
        col_of = {s.uid: k for k, s in enumerate(cols)}
        # ---- source text
        names = {}

        def ref(s):
            return names[s.uid]

        def emit(targets, lines, stop_at_cols):
            def visit(s):
                if s.uid in names:
                    return
                if s.op in ("x", "p", "dmin", "dmax", "par"):
                    names[s.uid] = "%s[%d]" % (s.op, s.value)
                    return
                if s.op == "const":
                    names[s.uid] = _literal(s.value)
                    return
                if stop_at_cols and s.uid in col_of:
                    names[s.uid] = "col[%d]" % col_of[s.uid]
                    return
                for a in s.args:
                    visit(a)
                a = [ref(q) for q in s.args]
                if s.op in _NUM_UNARY:
                    e = _NUM_UNARY[s.op].format(*a)
                elif s.op in _NUM_BINARY:
                    e = _NUM_BINARY[s.op].format(*a)
                elif s.op in _CMP:
                    e = "(%s %s %s)" % (a[0], _CMP[s.op], a[1])
                elif s.op in _BOOL_BINARY:
                    e = _BOOL_BINARY[s.op].format(*a)
                elif s.op == "not":
                    e = "(!%s)" % a[0]
                elif s.op == "where":
                    e = "(%s ? %s : %s)" % (a[0], a[1], a[2])
                elif s.op.startswith("tab"):
                    k = int(s.op[3:])
                    x0, inv, n = self._tab_geo[k]
                    used_tables.add(k)
                    e = "hjtab%d[min(max((int)rint((double)(%s - %s) * %r), 0), %d)]" % (k, a[0], _literal(x0), inv, n - 1)
                else:
                    raise TraceError("no device expression for '%s'" % s.op)
                nm = "v%d" % s.uid
                lines.append("const %s %s = %s;" % ("bool" if s.kind == "bool" else "T", nm, e))
                names[s.uid] = nm
            for s in targets:
                visit(s)

        def table_decls(which):
            out = []
            for k in sorted(which):
                d, vals = self.tables[k]
                out.append("static constexpr T hjtab%d[%d] = {%s};" % (k, len(vals), ", ".join(_literal(float(v)) for v in vals)))
            return out
        col_lines = []
        used_tables = set()
        if cols:
            emit(cols, col_lines, False)
            for k, s in enumerate(cols):
                col_lines.append("col[%d] = %s;" % (k, names[s.uid]))
            col_lines = table_decls(used_tables) + col_lines
        self.column_source = "\n".join(col_lines) if cols else None
        self.ncol = len(cols)
        names = {}
        lines = []
        used_tables = set()
        emit([H] + list(alpha), lines, True)
        lines = table_decls(used_tables) + lines
        lines.append("H = %s;" % names[H.uid])
        for d, a in enumerate(alpha):
            lines.append("alpha[%d] = %s;" % (d, names[a.uid]))
        self.source = "\n".join(lines)

    def evaluate(self, x, p, dmin=None, dmax=None, params=None):
        """The traced graph in NumPy: (H, [alpha_d]) for coordinate arrays x[d], costates p[d] (broadcastable), ranges dmin / dmax."""
        par = self.params if params is None else params
        val = {}
        need = [self._H] + list(self._alpha)

        def ev(s):
            if s.uid in val:
                return val[s.uid]
            if s.op == "x":
                r = np.asarray(x[s.value])
            elif s.op == "p":
                r = np.asarray(p[s.value])
            elif s.op == "dmin":
                r = np.asarray(dmin[s.value])
            elif s.op == "dmax":
                r = np.asarray(dmax[s.value])
            elif s.op == "par":
                r = np.float64(par[s.value])
            elif s.op == "const":
                r = np.float64(s.value)
            else:
                a = [ev(q) for q in s.args]
                if s.op.startswith("tab"):
                    k = int(s.op[3:])
                    x0, inv, n = self._tab_geo[k]
                    idx = np.clip(np.rint((np.asarray(a[0], dtype=np.float64) - x0) * inv).astype(np.int64), 0, n - 1)
                    r = self.tables[k][1][idx]
                elif s.op == "cast":
                    r = np.asarray(a[0], dtype=np.float64)
                elif s.op == "where":
                    r = np.where(a[0], a[1], a[2])
                else:
                    r = _NP_EVAL[s.op](*a)
            val[s.uid] = r
            return r
        with np.errstate(all="ignore"):
            return ev(need[0]), [ev(a) for a in need[1:]]


# Array constructors the callbacks typically pass a coordinate through -- torch.as_tensor(np.asarray(grid.xs[2]), device=p[0].device) --
# let a symbolic array through unchanged for the length of a trace (on the tracing thread only; everybody else gets the original).
_PASS_LOCK = threading.RLock()
_PASS_DEPTH = [0]
_PASS_SAVED = []


def _passthrough(orig):
    def f(a, *args, **kwargs):
        if isinstance(a, Sym) and getattr(_TLS, "tracer", None) is not None:
            return a
        return orig(a, *args, **kwargs)
    f.__name__ = getattr(orig, "__name__", "f")
    f.__wrapped__ = orig
    return f


def _patch_constructors():
    with _PASS_LOCK:
        _PASS_DEPTH[0] += 1
        if _PASS_DEPTH[0] > 1:
            return
        mods = [(np, ("asarray", "array", "ascontiguousarray", "asanyarray"))]
        import sys
        torch = sys.modules.get("torch")
        if torch is not None:
            mods.append((torch, ("as_tensor", "tensor", "from_numpy", "asarray")))
        for mod, names in mods:
            for n in names:
                orig = getattr(mod, n, None)
                if orig is not None:
                    _PASS_SAVED.append((mod, n, orig))
                    setattr(mod, n, _passthrough(orig))


def _unpatch_constructors():
    with _PASS_LOCK:
        _PASS_DEPTH[0] -= 1
        if _PASS_DEPTH[0] == 0:
            while _PASS_SAVED:
                mod, n, orig = _PASS_SAVED.pop()
                setattr(mod, n, orig)


class _StateGuard(object):
    """Callbacks may keep things for later (`self._x0 = as_tensor(grid.vs[0])` on first use, a dict of per-device copies): whatever they
    store during the trace is symbolic and must not survive it.  The attributes of the objects the callbacks are bound to (and of the
    schemeData) are put back as they were, dict / list attributes with their contents."""

    def __init__(self, objs):
        self.saved = []
        seen = set()
        for o in objs:
            d = getattr(o, "__dict__", None)
            if d is None or id(o) in seen or not isinstance(d, dict):
                continue
            seen.add(id(o))
            inner = []
            for v in d.values():
                if isinstance(v, dict):
                    inner.append((v, dict(v)))
                elif isinstance(v, list):
                    inner.append((v, list(v)))
            self.saved.append((o, dict(d), inner))

    def restore(self):
        for o, d, inner in self.saved:
            try:
                o.__dict__.clear()
                o.__dict__.update(d)
            except Exception:           # (an object that guards its attributes: leave it)
                pass
            for v, copy_ in inner:
                if isinstance(v, dict):
                    v.clear()
                    v.update(copy_)
                else:
                    v[:] = copy_


def _owners(hamFunc, partialFunc, schemeData):
    out = []
    for f in (hamFunc, partialFunc):
        while hasattr(f, "func") and hasattr(f, "args"):             # functools.partial
            out.extend(a for a in f.args if hasattr(a, "__dict__") and not callable(a))
            f = f.func
        o = getattr(f, "__self__", None)
        if o is not None:
            out.append(o)
    if schemeData is not None:
        out.append(schemeData)
    return out


# ---------------------------------------------------------------------------------------------- second chance: scalar idioms
# artificialDissipationGLF hands partialFunc NUMBERS (the range over the grid), so callbacks are written for numbers:
#     a = max(abs(float(derivMin[d])), abs(float(derivMax[d])));   math.cos(...);   v if v > 0 else 0
# float(), the builtin max / min, the math module and a conditional expression all need a real number NOW and refuse a symbolic one.
# When the first trace fails, the callbacks' SOURCE is rewritten -- float(x) -> x, max / min -> the element-wise pair, math.f -> the NumPy
# ufunc, `a if c else b` -> where(c, a, b), each only where an argument is symbolic, the builtin otherwise -- and traced again.  The rewritten
# function exists for the length of the trace only; the kernel it yields is checked against the ORIGINAL callbacks like any other.
def _hj_float(x):
    return x if isinstance(x, Sym) else float(x)


def _hj_minmax(op, builtin):
    def f(*args, **kwargs):
        flat = args[0] if len(args) == 1 and isinstance(args[0], (list, tuple)) else args
        if kwargs or not any(isinstance(a, Sym) for a in flat):
            return builtin(*args, **kwargs)
        r = flat[0]
        for a in flat[1:]:
            r = _bin(op, r, a)
        return r
    return f


class _MathShim(object):
    """`math` for the rewritten source: the NumPy ufunc of the same name where the argument is symbolic."""
    _NAMES = {"fabs": "abs", "acos": "arccos", "asin": "arcsin", "atan": "arctan", "atan2": "arctan2", "pow": "power"}

    def __getattr__(self, name):
        import math
        real = getattr(math, name)
        if not callable(real):
            return real
        uf = getattr(np, self._NAMES.get(name, name), None)

        def f(*args):
            if uf is not None and any(isinstance(a, Sym) for a in args):
                return uf(*args)
            return real(*args)
        return f


def _hj_ifexp(cond, a, b):
    if isinstance(cond, Sym):
        return _where(cond, a(), b())
    return a() if cond else b()


def _rewritten(cb):
    """The callback with its scalar idioms rewritten (see above), or None if its source cannot be had or nothing in it needs rewriting."""
    import ast
    import inspect
    import textwrap
    import types
    fn = getattr(cb, "__func__", cb)
    if not isinstance(fn, types.FunctionType):
        return None
    try:
        tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    except (OSError, TypeError, SyntaxError, IndentationError):
        return None
    if len(tree.body) != 1 or not isinstance(tree.body[0], ast.FunctionDef):
        return None               # (a lambda in the middle of an expression, a decorated oddity)
    changed = [0]

    class R(ast.NodeTransformer):
        def visit_Call(self, node):
            self.generic_visit(node)
            if isinstance(node.func, ast.Name) and node.func.id in ("float", "max", "min"):
                node.func = ast.Name(id="__hj_" + node.func.id, ctx=ast.Load())
                changed[0] += 1
            return node

        def visit_Attribute(self, node):
            self.generic_visit(node)
            if isinstance(node.value, ast.Name) and node.value.id == "math" and isinstance(node.ctx, ast.Load):
                node.value = ast.Name(id="__hj_math", ctx=ast.Load())
                changed[0] += 1
            return node

        def visit_IfExp(self, node):
            self.generic_visit(node)
            changed[0] += 1
            lam = lambda body: ast.Lambda(args=ast.arguments(posonlyargs=[], args=[], vararg=None, kwonlyargs=[], kw_defaults=[], kwarg=None, defaults=[]), body=body)  # noqa: E731
            return ast.Call(func=ast.Name(id="__hj_ifexp", ctx=ast.Load()), args=[node.test, lam(node.body), lam(node.orelse)], keywords=[])
    fdef = R().visit(tree.body[0])
    if not changed[0]:
        return None
    fdef.decorator_list = []
    free = list(fn.__code__.co_freevars)
    if "__class__" in free:
        return None               # (zero-argument super())
    factory = ast.FunctionDef(name="__hj_factory", args=ast.arguments(posonlyargs=[], args=[ast.arg(arg=v) for v in free], vararg=None, kwonlyargs=[],
                                                                      kw_defaults=[], kwarg=None, defaults=[]),
                              body=[fdef, ast.Return(value=ast.Name(id=fdef.name, ctx=ast.Load()))], decorator_list=[])
    mod = ast.Module(body=[factory], type_ignores=[])
    ast.fix_missing_locations(mod)
    glb = dict(fn.__globals__)
    glb.update(__hj_float=_hj_float, __hj_max=_hj_minmax("max", max), __hj_min=_hj_minmax("min", min), __hj_math=_MathShim(), __hj_ifexp=_hj_ifexp)
    try:
        exec(compile(mod, "<levelsetpy_amd trace of %s>" % getattr(fn, "__qualname__", fn.__name__), "exec"), glb)
        cells = []
        for c in fn.__closure__ or ():
            cells.append(c.cell_contents)
        new = glb["__hj_factory"](*cells)
    except Exception:
        return None
    new.__defaults__ = fn.__defaults__
    new.__kwdefaults__ = fn.__kwdefaults__
    owner = getattr(cb, "__self__", None)
    return types.MethodType(new, owner) if owner is not None else new


def trace_callbacks(grid, hamFunc, partialFunc, schemeData=None):
    """Call hamFunc / partialFunc once with symbolic arrays; Traced, or TraceError with the reason.  A pair that fails as written is tried
    once more with its scalar idioms rewritten (float(), max / min, math.*, conditional expressions)."""
    try:
        return _trace_once(grid, hamFunc, partialFunc, schemeData)
    except TraceError as first:
        if os.environ.get("HJ_TRACE_REWRITE", "1") in ("0", "off"):
            raise
        h2, p2 = _rewritten(hamFunc), _rewritten(partialFunc)
        if h2 is None and p2 is None:
            raise
        try:
            return _trace_once(grid, h2 or hamFunc, p2 or partialFunc, schemeData, owners=(hamFunc, partialFunc))
        except TraceError:
            raise first


def _trace_once(grid, hamFunc, partialFunc, schemeData=None, owners=None):
    tr = _Tracer(grid)
    saved = {}
    prev = getattr(_TLS, "tracer", None)
    guard = _StateGuard(_owners(*(owners or (hamFunc, partialFunc)), schemeData))
    _TLS.tracer = tr
    _patch_constructors()
    try:
        # the callbacks read the coordinates from the grid object: symbolic for the length of the trace
        for name in ("xs", "vs"):
            if name in grid.__dict__:
                saved[name] = grid.__dict__[name]
                setattr(grid, name, list(tr.x))
        hidden = {k: grid.__dict__.pop(k) for k in ("_hj_xs_t",) if k in grid.__dict__}      # (dynamics._xs: device copies of xs)
        try:
            res = hamFunc(tr.t, tr.data, list(tr.p), schemeData)
            if isinstance(res, tuple):
                if len(res) != 2 or (res[1] is not schemeData and getattr(res[1], "__dict__", None) != getattr(schemeData, "__dict__", 0)):
                    raise TraceError("hamFunc returns a modified schemeData")
                res = res[0]
            H = tr.lift(res)
            if H.kind == "bool":
                H = tr.node("cast", (H,), "num")
            alpha = []
            for d in range(tr.dim):
                a = tr.lift(partialFunc(tr.t, tr.data, list(tr.dmin), list(tr.dmax), schemeData, d))
                alpha.append(tr.node("cast", (a,), "num") if a.kind == "bool" else a)
        finally:
            guard.restore()
            for name, v in saved.items():
                setattr(grid, name, v)
            for k in [k for k, v in grid.__dict__.items() if isinstance(v, Sym) or (isinstance(v, (list, tuple)) and any(isinstance(e, Sym) for e in v))]:
                del grid.__dict__[k]          # (anything symbolic a callback parked on the grid)
            grid.__dict__.update(hidden)
        return Traced(tr, H, alpha)
    except TraceError:
        raise
    except RecursionError:
        raise TraceError("the callbacks recursed without end on symbolic arguments")
    except Exception as e:                     # the callback's own failure on arguments it did not expect
        raise TraceError("the callbacks raised %s on symbolic arguments: %s" % (type(e).__name__, e))
    finally:
        _unpatch_constructors()
        _TLS.tracer = prev


# ---------------------------------------------------------------------------------------------- registrations of traced pairs
_REG_BY_SOURCE = {}          # (dim, source, column source, uses_range, nparams) -> NativeRegistration
_BAD_SOURCES = set()         # failed the check against the callbacks (term.verify_traced)
_WHY_NOT = {}                # id(hamFunc.__func__ or hamFunc) -> reason, reported once (HJ_TRACE_VERBOSE)


def enabled():
    return os.environ.get("HJ_TRACE", "1") not in ("0", "off", "no")


def _scalarish(v):
    if v is None or isinstance(v, (bool, int, float, str, np.bool_, np.integer, np.floating)):
        return True
    return False


def _fingerprint_of(obj, depth, seen, out):
    d = getattr(obj, "__dict__", None)
    if d is None or id(obj) in seen or len(seen) > 64:
        return
    seen.add(id(obj))
    for k, v in d.items():
        if _scalarish(v):
            out.append((k, v))
        elif isinstance(v, (list, tuple)) and len(v) <= 16 and all(_scalarish(e) for e in v):
            out.append((k, tuple(v)))
        elif isinstance(v, np.ndarray) and v.size <= 8192:
            out.append((k, hash(v.tobytes())))          # (stored per-axis tables: their values are part of the traced expression)
        elif _is_tensor(v) and v.numel() <= 8192:
            out.append((k, (v.data_ptr(), getattr(v, "_version", 0))))
        elif depth > 0 and hasattr(v, "__dict__") and not callable(v) and k not in ("grid",) and not isinstance(v, (type, _ModuleType)):
            _fingerprint_of(v, depth - 1, seen, out)          # (not into modules or classes an object keeps a reference to: `self.torch = torch`)


def fingerprint(sd):
    """The scalar state the callbacks can be expected to read: attributes of the objects they are bound to (two levels), of the schemeData,
    closure cells and module globals they name.  A cached traced plan is re-traced when this changes (or always: HJ_TRACE_RECHECK=1)."""
    out, seen = [], set()
    for f in (sd.hamFunc, sd.partialFunc):
        while hasattr(f, "func") and hasattr(f, "args") and hasattr(f, "keywords"):          # functools.partial
            out.append(("<partial>", tuple(a for a in f.args if _scalarish(a)), tuple(sorted((k, v) for k, v in (f.keywords or {}).items() if _scalarish(v)))))
            for a in tuple(f.args) + tuple((f.keywords or {}).values()):
                if not _scalarish(a):
                    _fingerprint_of(a, 1, seen, out)
            f = f.func
        owner = getattr(f, "__self__", None)
        if owner is not None:
            _fingerprint_of(owner, 2, seen, out)
        fn = getattr(f, "__func__", f)
        for cell in getattr(fn, "__closure__", None) or ():
            try:
                v = cell.cell_contents
            except ValueError:
                continue
            if _scalarish(v):
                out.append(("<cell>", v))
            elif isinstance(v, np.ndarray) and v.size <= 8192:
                out.append(("<cell>", hash(v.tobytes())))             # (a per-axis table the closure captured)
            elif _is_tensor(v) and v.numel() <= 8192:
                out.append(("<cell>", (v.data_ptr(), getattr(v, "_version", 0))))
            else:
                _fingerprint_of(v, 1, seen, out)
        code, glb = getattr(fn, "__code__", None), getattr(fn, "__globals__", None)
        if code is not None and glb is not None:
            for name in code.co_names:
                v = glb.get(name)
                if v is not None and _scalarish(v):
                    out.append((name, v))
                elif isinstance(v, np.ndarray) and v.size <= 8192:
                    out.append((name, hash(v.tobytes())))
    _fingerprint_of(sd, 1, seen, out)
    return tuple(out)


class _TracedSystem(object):
    """What a plan carries for a traced pair: .grid, and ._hj_native with the registration and the parameters as of NOW."""

    def __init__(self, sd, reg, traced):
        self.grid = sd.grid
        self._sd = sd
        self._hj_native = self
        self.reg = reg
        self._key = _key_of(traced)
        self._params = list(traced.params)
        self._print = fingerprint(sd)

    def params(self, _obj=None):
        # parameters changed in place come out as new par[] values; a different EXPRESSION is a different kernel -- the cached plan is then
        # dropped by term.native_plan (the ham id it compares changes).  Re-traced when the scalar state the callbacks can see has changed
        # (fingerprint) or on every lookup (HJ_TRACE_RECHECK=1: state the fingerprint does not reach)
        if os.environ.get("HJ_TRACE_RECHECK", "0") in ("0", ""):
            fp = fingerprint(self._sd)
            if fp == self._print:
                return list(self._params)
            self._print = fp
        try:
            tr = trace_callbacks(self.grid, self._sd.hamFunc, self._sd.partialFunc, self._sd)
        except TraceError:
            self.reg = _NoReg
            return []
        if _key_of(tr) != self._key:
            reg = _registration(tr, getattr(self._sd.hamFunc, "__func__", self._sd.hamFunc), getattr(self._sd.hamFunc, "__self__", None))
            self.reg = reg if reg is not None else _NoReg
            self._key = _key_of(tr)
        self._params = list(tr.params)
        return list(tr.params)


class _NoRegType(object):
    ham_id = -1
    uses_range = False
    nparams = 0
    name = "untraceable"


_NoReg = _NoRegType()


def _key_of(tr):
    return (tr.dim, tr.source, tr.column_source, tr.uses_range, len(tr.params))


# A callback pair whose EXPRESSION keeps changing (a fifth float that varies per step, a table rebuilt every call) would compile a kernel per
# call -- a second or two each, where the split path takes milliseconds.  Distinct expressions per callback are counted; beyond the limit the
# pair is left on the split path for the rest of the process.
MAX_EXPRESSIONS_PER_CALLBACK = 8
_CHURN = {}                  # (id(code object), id(bound object)) -> hashes of the expression keys seen


def _registration(tr, ident=None, owner=None):
    key = _key_of(tr)
    if key in _BAD_SOURCES:
        return None
    reg = _REG_BY_SOURCE.get(key)
    if ident is not None:
        # (keyed by the CODE object -- a lambda or a closure made anew for every call is still the same callback -- and, for a bound method, the
        #  object it is bound to: one class may serve many systems, each with its own expression)
        seen = _CHURN.setdefault((id(getattr(ident, "__code__", ident)), id(owner) if owner is not None else 0), set())
        if hash(key) not in seen:
            if len(seen) >= MAX_EXPRESSIONS_PER_CALLBACK:
                if os.environ.get("HJ_TRACE_VERBOSE"):
                    warnings.warn("levelsetpy_amd: %r has produced more than %d different expressions: left on the split path" % (ident, MAX_EXPRESSIONS_PER_CALLBACK))
                return None
            seen.add(hash(key))
    if reg is None:
        from .user_ham import NativeRegistration
        import hashlib
        # (the name is part of the generated source, hence of the on-disk kernel cache's key: derived from the text, not from a counter)
        tag = hashlib.sha1(repr(key).encode()).hexdigest()[:12]
        reg = NativeRegistration("traced_%s" % tag, tr.dim, tr.source, nparams=len(tr.params), column_src=tr.column_source, ncol=tr.ncol,
                                 uses_range=tr.uses_range)
        reg.traced_key = key
        _REG_BY_SOURCE[key] = reg
    return reg


def mark_bad(reg):
    key = getattr(reg, "traced_key", None)
    if key is not None:
        _BAD_SOURCES.add(key)
        _REG_BY_SOURCE.pop(key, None)


def traced_native(sd):
    """(system, ham_id, params) like dynamics.native_of for a schemeData whose callbacks could be traced, else None."""
    if not enabled():
        return None
    ident = getattr(sd.hamFunc, "__func__", sd.hamFunc)
    try:
        tr = trace_callbacks(sd.grid, sd.hamFunc, sd.partialFunc, sd)
        reg = _registration(tr, ident, getattr(sd.hamFunc, "__self__", None))
        if reg is None:
            return None
    except TraceError as e:
        if os.environ.get("HJ_TRACE_VERBOSE") and _WHY_NOT.get(id(ident)) != str(e):
            _WHY_NOT[id(ident)] = str(e)
            warnings.warn("levelsetpy_amd: hamFunc / partialFunc stay on the split path: %s" % e)
        return None
    except ValueError as e:          # the library refused the registration
        if os.environ.get("HJ_TRACE_VERBOSE"):
            warnings.warn("levelsetpy_amd: traced callbacks were not registered: %s" % e)
        return None
    system = _TracedSystem(sd, reg, tr)
    return system, reg.ham_id, list(tr.params)
